// Error plumbing + device query for libavid_hip.so.
#include <stdarg.h>

#include "common.h"

namespace avid {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace avid

extern "C" const char* avid_last_error(void) { return avid::g_err; }
extern "C" int avid_version(void) { return 100; }

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
