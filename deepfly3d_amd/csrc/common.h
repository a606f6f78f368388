// Shared helpers for libdf3d_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/df3d_hip.h"

namespace df3d {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

#define DF3D_CHECK_ARG(cond, msg)                     \
    do {                                              \
        if (!(cond)) {                                \
            df3d::set_error("%s: %s", __func__, msg); \
            return DF3D_EINVAL;                       \
        }                                             \
    } while (0)

#define DF3D_HIP(call)                                                                         \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess) {                                                               \
            df3d::set_error("%s: %s failed: %s", __func__, #call, hipGetErrorString(e__));     \
            return DF3D_EHIP;                                                                  \
        }                                                                                      \
    } while (0)

#define DF3D_LAUNCH_CHECK()                                                                    \
    do {                                                                                       \
        hipError_t e__ = hipGetLastError();                                                    \
        if (e__ != hipSuccess) {                                                               \
            df3d::set_error("%s: kernel launch failed: %s", __func__, hipGetErrorString(e__)); \
            return DF3D_EHIP;                                                                  \
        }                                                                                      \
    } while (0)

constexpr int WAVE = 64;

}  // namespace df3d
