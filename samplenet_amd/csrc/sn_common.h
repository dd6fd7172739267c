// sn_common.h -- shared helpers of libsamplenet_hip.so (error reporting, small device utilities).
#pragma once
#include <hip/hip_runtime.h>

#include "samplenet_hip.h"

int sn_set_error(int code, const char *fmt, ...);

#define SN_REQUIRE(cond, msg)                                                           \
    do {                                                                                \
        (void)hipGetLastError();                                                        \
        if (!(cond)) return sn_set_error(SN_ERR_BAD_ARGUMENT, "%s: %s", __func__, msg); \
    } while (0)

// hipGetLastError() reports (and clears) the most recent error of ANY runtime call on this host
// thread, including calls made by other libraries in the process: discard stale state on entry so
// that SN_LAUNCH_CHECK only sees this entry point's own launches.
#define SN_ENTER() (void)hipGetLastError()

#define SN_LAUNCH_CHECK()                                                                        \
    do {                                                                                         \
        hipError_t e_ = hipGetLastError();                                                       \
        if (e_ != hipSuccess) return sn_set_error((int)e_, "%s: %s", __func__, hipGetErrorString(e_)); \
    } while (0)

typedef unsigned long long sn_u64;

namespace sn {

constexpr int kWave = 64;
constexpr sn_u64 kKeyInf = ~0ull;

// (distance, index) packed so that unsigned order == lexicographic (distance, index) order.
// Valid for distances >= +0 (sums of squares): their IEEE bit patterns are monotone as integers.
__device__ __forceinline__ sn_u64 make_key(float d, int idx)
{
    return ((sn_u64)__float_as_uint(d) << 32) | (unsigned)idx;
}
__device__ __forceinline__ float key_dist(sn_u64 k) { return __uint_as_float((unsigned)(k >> 32)); }
__device__ __forceinline__ int key_index(sn_u64 k) { return (int)(unsigned)k; }

__device__ __forceinline__ float readlane_f(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// element offset of channel c of point i in a (N,3) [BNC] or (3,N) [BCN] cloud
__device__ __forceinline__ int pt_off(int layout, int npts, int i, int c)
{
    return layout == SN_LAYOUT_BNC ? i * 3 + c : c * npts + i;
}

}  // namespace sn
