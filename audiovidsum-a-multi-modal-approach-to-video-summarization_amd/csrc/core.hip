// Library-level entry points of libavsum_hip.so: version, error string, device info.
#include "avs_internal.h"
#include <string.h>

static thread_local char g_err[512] = "";

void avs_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int avs_abi_version(void) { return AVS_ABI_VERSION; }

extern "C" const char* avs_last_error(void) { return g_err; }

extern "C" int avs_device_info(int dev, int* cu_count, int* clock_khz, int64_t* hbm_bytes, char* arch,
                               int arch_len) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) {
    avs_set_error("avs_device_info: %s", hipGetErrorString(e));
    return AVS_E_HIP;
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (clock_khz) *clock_khz = prop.clockRate;
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  if (arch && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return AVS_OK;
}
