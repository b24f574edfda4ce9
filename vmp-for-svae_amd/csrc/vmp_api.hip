// ABI plumbing of libvmp_hip.so: version, thread-local error message, launch check.
#include "vmp_common.h"

namespace vmp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

size_t lds_budget() {
    int dev = 0, v = 0;
    size_t cap = 160 * 1024;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0 && (size_t)v < cap)
        cap = (size_t)v;
    return cap;
}

int set_dyn_lds(const void* kernel, size_t bytes, const char* what) {
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("%s: %zu bytes of dynamic LDS refused (%s); the device offers %zu", what, bytes, hipGetErrorString(e), lds_budget());
        return (int)e;
    }
    return 0;
}

}  // namespace vmp

extern "C" {
int vmp_abi_version(void) { return VMP_ABI_VERSION; }
const char* vmp_last_error(void) { return vmp::g_err; }
}
