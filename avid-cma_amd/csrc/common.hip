// Error plumbing + device query for libavid_hip.so.
#include <stdarg.h>
#include <stdlib.h>

#include <map>
#include <string>
#include <vector>

#include <atomic>
#include "common.h"

namespace avid {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

struct TimingRec {
  const char* name;
  double flops, bytes;
  hipEvent_t e0, e1;
};
static bool g_timing = false;
static std::vector<TimingRec> g_recs;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

ScopedTimer::ScopedTimer(hipStream_t stream, const char* name, double flops, double bytes) : s(stream), slot(-1) {
  if (!g_timing) return;
  TimingRec r{name, flops, bytes, get_event(), get_event()};
  (void)hipEventRecord(r.e0, s);
  slot = (int)g_recs.size();
  g_recs.push_back(r);
}
ScopedTimer::~ScopedTimer() {
  if (slot >= 0) (void)hipEventRecord(g_recs[slot].e1, s);
}

// ---- CU budget of the persistent kernels (include/avid_hip.h: avid_set_cu_budget)
// -1: not configured (AVID_CU_RESERVE from the environment on first use); 0: every CU.  Read from the forward thread and from
// autograd's backward thread: atomic (relaxed; both would initialise it to the same value)
static std::atomic<int> g_cu_budget{-1};

static int physical_cus() {       // per device: a process may touch more than one
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!cus[dev]) {
    hipDeviceProp_t pr;
    int n = 0;
    if (hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
    cus[dev] = n > 0 ? n : 256;
  }
  return cus[dev];
}

int device_cus() {
  const int phys = physical_cus();
  int budget = g_cu_budget.load(std::memory_order_relaxed);
  if (budget < 0) {
    const char* e = getenv("AVID_CU_RESERVE");
    const int reserve = e ? atoi(e) : 0;
    budget = reserve > 0 && reserve < phys ? phys - reserve : 0;
    g_cu_budget.store(budget, std::memory_order_relaxed);
  }
  if (budget <= 0 || budget >= phys) return phys;
  // whole CUs per XCD (the logical workgroup numberings deal a contiguous eighth of every round to each XCD), at least one
  const int c = budget / 8 * 8;
  return c >= 8 ? c : 8;
}
}  // namespace avid

extern "C" int avid_set_cu_budget(int cus) {
  avid::g_cu_budget.store(cus > 0 ? cus : 0, std::memory_order_relaxed);
  return avid::device_cus();
}

extern "C" int avid_cu_budget(void) { return avid::device_cus(); }

extern "C" int avid_timing_enable(int on) {
  using namespace avid;
  for (auto& r : g_recs) {
    g_pool.push_back(r.e0);
    g_pool.push_back(r.e1);
  }
  g_recs.clear();
  g_timing = on != 0;
  return AVID_OK;
}

// CSV: name,launches,total_ms,total_flops,total_bytes  (one line per kernel; synchronises the events)
extern "C" int avid_timing_report(char* buf, size_t len) {
  using namespace avid;
  if (!buf || len == 0) {
    set_error("timing_report: bad buffer");
    return AVID_E_BADARG;
  }
  struct Agg { long n = 0; double ms = 0, flops = 0, bytes = 0; };
  std::map<std::string, Agg> agg;
  for (auto& r : g_recs) {
    if (hipEventSynchronize(r.e1) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    Agg& a = agg[r.name];
    a.n += 1; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
  }
  std::string out;
  char line[512];
  for (auto& kv : agg) {
    snprintf(line, sizeof(line), "%s;%ld;%.6f;%.6e;%.6e\n", kv.first.c_str(), kv.second.n, kv.second.ms,
             kv.second.flops, kv.second.bytes);
    out += line;
  }
  if (out.size() + 1 > len) {
    set_error("timing_report: buffer too small (%zu needed)", out.size() + 1);
    return AVID_E_BADARG;
  }
  memcpy(buf, out.c_str(), out.size() + 1);
  return AVID_OK;
}

extern "C" const char* avid_last_error(void) { return avid::g_err; }
extern "C" int avid_version(void) { return 130; }

extern "C" int avid_device_info(int device, int* cu_count, int* lds_bytes, char* arch, int arch_len) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    avid::set_error("hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e));
    return AVID_E_HIP;
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (arch && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return AVID_OK;
}
