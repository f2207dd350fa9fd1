// capi.hip - error plumbing and device queries of the C-ABI (include/vdm4cdm_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"

namespace vdm {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return VDM_OK;
    set_error("%s: %s (%s)", what, hipGetErrorString(e), hipGetErrorName(e));
    return VDM_ERR_HIP;
}

}  // namespace vdm

extern "C" const char* vdm_last_error(void) { return vdm::g_err; }

extern "C" int vdm_abi_version(void) { return VDM_ABI_VERSION; }

extern "C" int vdm_device_info(int device, int* cu_count, int* lds_bytes, char* arch_name) {
    hipDeviceProp_t p;
    int e = vdm::check_hip(hipGetDeviceProperties(&p, device), "hipGetDeviceProperties");
    if (e) return e;
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)p.maxSharedMemoryPerMultiProcessor;
    if (arch_name) {
        strncpy(arch_name, p.gcnArchName, 63);
        arch_name[63] = 0;
    }
    return VDM_OK;
}
