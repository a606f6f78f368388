// Error reporting + device queries for libdf3d_hip.so.
#include "common.h"

namespace df3d {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace df3d

extern "C" {

const char* df3d_last_error(void) { return df3d::g_err; }

int df3d_version(void) { return DF3D_ABI_VERSION; }

int df3d_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        df3d::set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        return DF3D_ENOGPU;
    }
    return n;
}

int df3d_device_name(int dev, char* buf, int buflen) {
    DF3D_CHECK_ARG(buf && buflen > 0, "null buffer");
    hipDeviceProp_t prop;
    DF3D_HIP(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, buflen, "%s (%s)", prop.name, prop.gcnArchName);
    return DF3D_OK;
}

}  // extern "C"
