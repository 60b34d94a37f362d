// Launch programs (include/avid_hip.h, "Launch programs"): the host side of a forward / backward pass as one call.
//
// avid_program_run walks an array of avid_instr records and calls the library's own entry points — exactly the calls a
// binding would make one by one (avid-cma_amd/avid_hip/ops.py), so results are bit-identical — and places the
// cross-stream dependencies itself.  Host cost per record: reference resolution (a few adds) + the entry point's own
// dispatch + hipLaunchKernel; no interpreter, no allocator, no autograd node.
#include <atomic>
#include <vector>

#include "common.h"

namespace avid {

// Events for AVID_OP_WAIT.  An event may be re-recorded while an earlier wait on it is still pending: a
// hipStreamWaitEvent captures the record that is current when it is called.  One pool per device, round-robin.
// The index is atomic: the thread that runs the forward program and autograd's thread (the backward program) both draw
// from the pool; the events themselves are created on first use under a flag (the first call of a process is the
// forward's).  Contract (include/avid_hip.h, avid_stream_wait): an event is re-used after 64 draws — a wait is "consumed"
// the moment hipStreamWaitEvent returns, so re-recording never disturbs an earlier wait.
struct EventPool {
  hipEvent_t ev[64];
  std::atomic<unsigned> next{0};
  std::atomic<int> ready{0};      // 0: empty, 1: being filled, 2: filled
};
static EventPool g_wait_events[16];

static hipEvent_t wait_event() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  EventPool& p = g_wait_events[dev & 15];
  int st = p.ready.load(std::memory_order_acquire);
  if (st != 2) {
    int expect = 0;
    if (p.ready.compare_exchange_strong(expect, 1)) {
      for (auto& e : p.ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
      p.ready.store(2, std::memory_order_release);
    } else {
      while (p.ready.load(std::memory_order_acquire) != 2) {}
    }
  }
  return p.ev[p.next.fetch_add(1, std::memory_order_relaxed) & 63];
}

static inline size_t maxz(size_t a, size_t b) { return a > b ? a : b; }

}  // namespace avid

using namespace avid;

static size_t instr_ws_bytes(const avid_instr* prog, int k, int end) {
  const avid_instr& in = prog[k];
  switch (in.op) {
    case AVID_OP_CONV_FWD: return avid_conv_fwd_workspace_bytes(&in.d);
    case AVID_OP_CONV_DGRAD: return avid_conv_dgrad_workspace_bytes(&in.d);
    case AVID_OP_CONV_WGRAD: return avid_conv_wgrad_workspace_bytes(&in.d);
    case AVID_OP_WGRAD_GROUP: {
      avid_wgrad_item items[12];
      const int n = in.i[0];
      if (n < 1 || n > 12 || k + n >= end) return 0;
      for (int j = 0; j < n; ++j) {
        items[j].d = prog[k + 1 + j].d;
        items[j].x = items[j].dy = nullptr;
        items[j].dw = nullptr;
      }
      return avid_conv_wgrad_group_workspace_bytes(n, items);
    }
    case AVID_OP_BN_FWD:
    case AVID_OP_BN_BWD: return avid_bn_workspace_bytes(in.n[0], in.i[0]);
    case AVID_OP_BN_POOL_FWD:
    case AVID_OP_BN_POOL_BWD: return avid_bn_workspace_bytes((int64_t)in.i[0] * in.i[1] * in.i[2] * in.i[3], in.i[4]);
    default: return 0;
  }
}

// One wave that spins for ~`us` microseconds (s_memrealtime: 100 MHz constant clock): the probe kernel of
// avid_hip/streams.py, which looks for stream pairs that hardware queues / pipes serialise.
__global__ void probe_spin_kernel(long long ticks, int* sink) {
  const long long t0 = __builtin_readcyclecounter();
  long long t = t0;
  int acc = 0;
  while (t - t0 < ticks) {
    __builtin_amdgcn_s_sleep(8);
    t = __builtin_readcyclecounter();
    ++acc;
  }
  if (sink && ticks < 0) *sink = acc;
}

extern "C" int avid_probe_spin(int us, avid_stream_t stream) {
  AVID_REQUIRE(us > 0 && us < 100000, AVID_E_BADARG, "probe_spin: duration out of range");
  // s_memtime counts shader-clock cycles on gfx9 (~2.1-2.4 GHz under no load): an approximate duration is all the probe needs
  hipLaunchKernelGGL(probe_spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)us * 2100, nullptr);
  return check_launch("probe_spin");
}

// One wave that spins for `us` microseconds of the constant 100 MHz clock and reports how many shader-clock cycles
// went by: out[0] = s_memtime ticks (shader clock), out[1] = s_memrealtime ticks (100 MHz).
__global__ void clock_probe_kernel(long long real_ticks, long long* out) {
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  const long long c0 = __builtin_readcyclecounter();
  long long r = r0;
  while (r - r0 < real_ticks) {
    __builtin_amdgcn_s_sleep(32);
    r = __builtin_amdgcn_s_memrealtime();
  }
  const long long c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    out[0] = c1 - c0;
    out[1] = r - r0;
  }
}

extern "C" int avid_clock_probe(int us, long long* out2, avid_stream_t stream) {
  AVID_REQUIRE(us > 0 && us <= 2000000 && out2, AVID_E_BADARG, "clock_probe: bad arguments");
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)us * 100, out2);
  return check_launch("clock_probe");
}

extern "C" int avid_stream_wait(avid_stream_t waiter, avid_stream_t waited) {
  if (waiter == waited) return AVID_OK;
  hipEvent_t e = wait_event();
  hipError_t he = hipEventRecord(e, (hipStream_t)waited);
  if (he == hipSuccess) he = hipStreamWaitEvent((hipStream_t)waiter, e, 0);
  if (he != hipSuccess) {
    set_error("stream_wait: %s", hipGetErrorString(he));
    return AVID_E_HIP;
  }
  return AVID_OK;
}

extern "C" size_t avid_program_instr_bytes(void) { return sizeof(avid_instr); }

extern "C" int avid_program_workspace_bytes(const avid_instr* prog, int begin, int end, int n_streams, size_t* out_bytes) {
  AVID_REQUIRE(prog && out_bytes && begin >= 0 && end >= begin && n_streams > 0, AVID_E_BADARG, "program_workspace_bytes: bad arguments");
  for (int s = 0; s < n_streams; ++s) out_bytes[s] = 0;
  for (int k = begin; k < end; ++k) {
    const avid_instr& in = prog[k];
    if (in.op == AVID_OP_WAIT || in.op == AVID_OP_WGRAD_ITEM || in.op == AVID_OP_NOP) continue;
    AVID_REQUIRE(in.stream >= 0 && in.stream < n_streams, AVID_E_BADARG, "program record %d: stream %d of %d", k, in.stream, n_streams);
    out_bytes[in.stream] = maxz(out_bytes[in.stream], instr_ws_bytes(prog, k, end));
  }
  return AVID_OK;
}

extern "C" int avid_program_run(const avid_instr* prog, int begin, int end, void* const* slots, int n_slots,
                                const avid_stream_t* streams, const avid_stream_ws* ws, int n_streams) {
  AVID_REQUIRE(prog && slots && streams && ws && begin >= 0 && end >= begin && n_streams > 0, AVID_E_BADARG,
               "program_run: bad arguments");
  // every tensor reference of [begin, end) is resolved BEFORE the first launch: a reference to an empty or out-of-range
  // slot would otherwise become a null OPTIONAL operand (addend, statistics, pre-split weights ...) and run the record
  // with other semantics, behind records that have already been launched
  constexpr int NREF = (int)(sizeof(prog[0].t) / sizeof(prog[0].t[0]));
  for (int k = begin; k < end; ++k) {
    const avid_instr& in = prog[k];
    if (in.op == AVID_OP_NOP) continue;
    AVID_REQUIRE(in.op > AVID_OP_NOP && in.op < AVID_OP_COUNT_, AVID_E_BADARG, "program record %d: unknown kind %d", k, in.op);
    if (in.op == AVID_OP_WAIT) {
      AVID_REQUIRE(in.i[0] >= 0 && in.i[0] < n_streams && in.i[1] >= 0 && in.i[1] < n_streams, AVID_E_BADARG,
                   "program record %d: wait %d <- %d of %d streams", k, in.i[0], in.i[1], n_streams);
      continue;
    }
    if (in.op != AVID_OP_WGRAD_ITEM)
      AVID_REQUIRE(in.stream >= 0 && in.stream < n_streams, AVID_E_BADARG, "program record %d: stream %d of %d", k, in.stream, n_streams);
    for (int j = 0; j < NREF; ++j) {
      const int sl = in.t[j].slot;
      AVID_REQUIRE(sl < 0 || (sl < n_slots && slots[sl]), AVID_E_BADARG,
                   "program record %d (kind %d): tensor reference %d names slot %d — outside [0, %d) or empty; nothing was launched",
                   k, in.op, j, sl, n_slots);
    }
  }
  auto P = [&](const avid_ref& r) -> char* {
    if (r.slot < 0) return nullptr;
    return static_cast<char*>(slots[r.slot]) + r.off;
  };
#define F(r) reinterpret_cast<float*>(P(r))
  for (int k = begin; k < end; ++k) {
    const avid_instr& in = prog[k];
    const avid_ref* t = in.t;
    int rc = AVID_OK;
    if (in.op == AVID_OP_NOP || in.op == AVID_OP_WGRAD_ITEM) continue;
    if (in.op == AVID_OP_WAIT) {
      const int a = in.i[0], b = in.i[1];
      AVID_REQUIRE(a >= 0 && a < n_streams && b >= 0 && b < n_streams, AVID_E_BADARG, "program record %d: wait %d <- %d of %d streams",
                   k, a, b, n_streams);
      if (a == b || streams[a] == streams[b]) continue;
      hipEvent_t e = wait_event();
      hipError_t he = hipEventRecord(e, (hipStream_t)streams[b]);
      if (he == hipSuccess) he = hipStreamWaitEvent((hipStream_t)streams[a], e, 0);
      if (he != hipSuccess) {
        set_error("program record %d (wait): %s", k, hipGetErrorString(he));
        return AVID_E_HIP;
      }
      continue;
    }
    AVID_REQUIRE(in.stream >= 0 && in.stream < n_streams, AVID_E_BADARG, "program record %d: stream %d of %d", k, in.stream, n_streams);
    avid_stream_t s = streams[in.stream];
    void* w = ws[in.stream].ptr;
    const size_t wb = ws[in.stream].bytes;
    switch (in.op) {
      case AVID_OP_MEMSET0: {
        hipError_t he = hipMemsetAsync(P(t[0]), 0, (size_t)in.n[0], (hipStream_t)s);
        if (he != hipSuccess) {
          set_error("memset: %s", hipGetErrorString(he));
          rc = AVID_E_HIP;
        }
        break;
      }
      case AVID_OP_CONV_FWD: {
        avid_in_affine aff;       // i[1]: 0 none, 1 the input's BatchNorm, 2 ... + ReLU; t[7]: its saved [4][C] vectors, i[2] = C
        if (in.i[1]) {
          const float* s4 = F(t[7]);
          aff.scale = s4 + 2 * in.i[2]; aff.shift = s4 + 3 * in.i[2]; aff.relu = in.i[1] == 2;
        }
        rc = avid_conv_fwd_in(&in.d, F(t[0]), in.i[1] ? &aff : nullptr, F(t[1]), F(t[2]), F(t[3]), F(t[4]), in.i[0], F(t[5]), F(t[6]),
                              w, wb, s);
        break;
      }
      case AVID_OP_CONV_DGRAD: {
        avid_bn_bwd_fuse bn;
        if (in.i[4]) {
          bn.x = F(t[6]); bn.scale = F(t[7]); bn.shift = F(t[8]); bn.mean = F(t[9]); bn.invstd = F(t[10]);
          bn.relu = in.i[3];
          bn.partials = F(t[11]);
        }
        const bool strided_add = in.i[0] > 0;
        rc = avid_conv_dgrad(&in.d, F(t[0]), F(t[1]), F(t[2]), F(t[3]), F(t[4]), strided_add ? in.i : nullptr, F(t[5]),
                             in.i[4] ? &bn : nullptr, w, wb, s);
        break;
      }
      case AVID_OP_CONV_WGRAD: {
        avid_in_affine aff;       // i[0]: 0 none, 1 the input's BatchNorm, 2 ... + ReLU; t[3]: its saved [4][C] vectors, i[1] = C
        if (in.i[0]) {
          const float* s4 = F(t[3]);
          aff.scale = s4 + 2 * in.i[1]; aff.shift = s4 + 3 * in.i[1]; aff.relu = in.i[0] == 2;
        }
        rc = avid_conv_wgrad_in(&in.d, F(t[0]), in.i[0] ? &aff : nullptr, F(t[1]), F(t[2]), w, wb, s);
        break;
      }
      case AVID_OP_WGRAD_GROUP: {
        const int n = in.i[0];
        AVID_REQUIRE(n >= 1 && n <= 12 && k + n < end, AVID_E_BADARG, "program record %d: a group of %d items", k, n);
        avid_wgrad_item items[12];
        for (int j = 0; j < n; ++j) {
          const avid_instr& it = prog[k + 1 + j];
          AVID_REQUIRE(it.op == AVID_OP_WGRAD_ITEM, AVID_E_BADARG, "program record %d: item %d of the group is a record of kind %d", k, j, it.op);
          items[j].d = it.d;
          items[j].x = F(it.t[0]);
          items[j].dy = F(it.t[1]);
          items[j].dw = F(it.t[2]);
        }
        rc = avid_conv_wgrad_group(n, items, w, wb, s);
        break;
      }
      case AVID_OP_BN_FWD: {
        const int C = in.i[0];
        float* s4 = F(t[6]);
        rc = avid_bn_fwd_train(in.n[0], C, F(t[0]), F(t[1]), F(t[2]), F(t[3]), F(t[4]), in.f[0], in.f[1], in.i[1], F(t[5]), s4,
                               s4 + C, s4 + 2 * C, s4 + 3 * C, reinterpret_cast<int64_t*>(P(t[7])), F(t[8]), in.i[2], w, wb, s);
        break;
      }
      case AVID_OP_BN_BWD: {
        const int C = in.i[0];
        const float* s4 = F(t[3]);
        rc = avid_bn_bwd(in.n[0], C, F(t[0]), F(t[1]), F(t[2]), s4, s4 + C, s4 + 2 * C, s4 + 3 * C, in.i[1], F(t[4]), F(t[5]),
                         F(t[6]), F(t[7]), in.i[2], in.i[3], w, wb, s);
        break;
      }
      case AVID_OP_BN_POOL_FWD: {
        const int C = in.i[4];
        float* s4 = F(t[7]);
        rc = avid_bn_relu_maxpool_fwd(in.i[0], in.i[1], in.i[2], in.i[3], C, F(t[0]), F(t[1]), F(t[2]), F(t[3]), F(t[4]), in.f[0],
                                      in.f[1], F(t[5]), reinterpret_cast<uint8_t*>(P(t[6])), s4, s4 + C, s4 + 2 * C, s4 + 3 * C,
                                      reinterpret_cast<int64_t*>(P(t[8])), F(t[9]), in.i[5], w, wb, s);
        break;
      }
      case AVID_OP_BN_POOL_BWD: {
        const int C = in.i[4];
        const float* s4 = F(t[4]);
        rc = avid_bn_relu_maxpool_bwd(in.i[0], in.i[1], in.i[2], in.i[3], C, F(t[0]), F(t[1]), reinterpret_cast<uint8_t*>(P(t[2])),
                                      F(t[3]), s4, s4 + C, s4 + 2 * C, s4 + 3 * C, F(t[5]), F(t[6]), F(t[7]), w, wb, s);
        break;
      }
      case AVID_OP_GPOOL_FWD:
        rc = avid_global_maxpool_fwd(in.i[0], in.i[1], in.i[2], F(t[0]), F(t[1]), reinterpret_cast<int32_t*>(P(t[2])), s);
        break;
      case AVID_OP_GPOOL_BWD:
        rc = avid_global_maxpool_bwd(in.i[0], in.i[1], in.i[2], F(t[0]), reinterpret_cast<int32_t*>(P(t[1])), F(t[2]), s);
        break;
      case AVID_OP_RELU_BWD:
        rc = avid_relu_bwd(in.n[0], F(t[0]), F(t[1]), F(t[2]), s);
        break;
      case AVID_OP_COLSUM:
        rc = avid_colsum(in.n[0], in.i[0], F(t[0]), F(t[1]), s);
        break;
      case AVID_OP_WT_BATCH:
        rc = avid_weight_transpose_batched(in.i[0], reinterpret_cast<const avid_wt_desc*>(P(t[0])), in.n[0], s);
        break;
      case AVID_OP_ADAM: {
        uint64_t* step_dev = reinterpret_cast<uint64_t*>(P(t[4]));
        if (step_dev) rc = avid_counter_add(step_dev, 1, s);
        if (rc == AVID_OK)
          rc = avid_adam_flat(in.n[0], F(t[0]), F(t[1]), F(t[2]), F(t[3]), in.f[0], in.f[1], in.f[2], in.f[3], in.f[4], in.n[1],
                              reinterpret_cast<const int64_t*>(step_dev), F(t[5]), in.f[5], s);
        break;
      }
      default:
        set_error("program record %d: unknown kind %d", k, in.op);
        return AVID_E_BADARG;
    }
    if (rc != AVID_OK) {
      char msg[400];
      snprintf(msg, sizeof(msg), "%s", avid_last_error());
      set_error("program record %d (kind %d): %s", k, in.op, msg);
      return rc;
    }
  }
#undef F
  return AVID_OK;
}
