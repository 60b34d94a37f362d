// Dev tool: a stand-in for a communication kernel that shares the GPU with the training step — G workgroups of
// T threads, `lds` bytes of LDS each, that stream over a buffer for `us` microseconds (one workgroup per "channel",
// long-lived, memory-bound, few registers: RCCL's footprint, not its traffic pattern).  Built by tools/interference.py.
#include <hip/hip_runtime.h>
extern "C" __global__ void interfere_kernel(float* buf, long long n, long long ticks) {
  extern __shared__ float sh[];
  const long long t0 = wall_clock64();
  float acc = 0.f;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  while (wall_clock64() - t0 < ticks) {            // 100 MHz constant clock
    acc += buf[i % n];
    i += (long long)gridDim.x * blockDim.x;
    if (threadIdx.x == 0) sh[0] = acc;
  }
  if (acc == 12345.f) buf[0] = acc + sh[0];
}
extern "C" int interfere_launch(float* buf, long long n, int G, int T, int lds, double us, void* stream) {
  hipLaunchKernelGGL(interfere_kernel, dim3(G), dim3(T), lds, (hipStream_t)stream, buf, n, (long long)(us * 100.0));
  return (int)hipGetLastError();
}
