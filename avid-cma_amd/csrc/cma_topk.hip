// CMA correspondence search (criterions/avid_cma.py:42-73) — SURVEY.md §8(f) rank 1 ("next").
#include "common.h"

using namespace avid;

extern "C" size_t avid_cma_topk_workspace_bytes(int64_t N, int nq, int pos_k) {
  (void)N; (void)nq; (void)pos_k;
  return 0;
}

extern "C" int avid_cma_topk(int64_t N, int D, const float* view1, const float* view2, int64_t q0, int nq, int pos_k,
                             int kind, int32_t* out, void* ws, size_t ws_bytes, avid_stream_t stream) {
  (void)N; (void)D; (void)view1; (void)view2; (void)q0; (void)nq; (void)pos_k; (void)kind; (void)out; (void)ws;
  (void)ws_bytes; (void)stream;
  set_error("cma_topk: not implemented yet");
  return AVID_E_UNSUPPORTED;
}
