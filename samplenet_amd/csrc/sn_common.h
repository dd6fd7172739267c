// sn_common.h -- shared helpers of libsamplenet_hip.so (error reporting, small device utilities).
#pragma once
#include <hip/hip_runtime.h>

#include "samplenet_hip.h"
#include "samplenet_hip_internal.h"

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

// More than 64 KB of dynamic LDS must be requested per kernel with hipFuncSetAttribute -- and the attribute is PER DEVICE: a
// process that drives several GPUs needs it on each.  One SnLdsAttr per kernel (function-local static): a bit per
// device ordinal, the request's return code checked.  bytes: the largest dynamic LDS size the kernel is ever launched with.
struct SnLdsAttr {
    sn_u64 done = 0;  // (benign race: two threads may both issue the idempotent request)
};
inline int sn_lds_attr(SnLdsAttr &a, const void *fn, size_t bytes, const char *who)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const sn_u64 bit = 1ull << (dev & 63);
    if (a.done & bit) return 0;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return sn_set_error(SN_ERR_UNSUPPORTED, "%s: %zu bytes of dynamic LDS refused on device %d (%s)", who, bytes, dev, hipGetErrorString(e));
    a.done |= bit;
    return 0;
}
// the same for a kernel whose LDS size depends on the call's shape: the request is renewed whenever a call needs more than this
// device has been granted so far
struct SnLdsAttrGrow {
    size_t got[64] = {};
};
inline int sn_lds_attr_grow(SnLdsAttrGrow &a, const void *fn, size_t bytes, const char *who)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    size_t &g = a.got[dev & 63];
    if (bytes <= g) return 0;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return sn_set_error(SN_ERR_UNSUPPORTED, "%s: %zu bytes of dynamic LDS refused on device %d (%s)", who, bytes, dev, hipGetErrorString(e));
    g = bytes;
    return 0;
}

// spacing (in 32-bit words) of the words of an FC chain launch's `sync` state: word i lives at sync[i * SN_FC_SYNC_STRIDE]
// (fc_chain.hip: kFcSyncStride; geometry_ops.hip: the step tail reads the error words)
#ifndef SN_FC_SYNC_STRIDE
#define SN_FC_SYNC_STRIDE 32
#endif

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

// Touches every 64-byte line of the first BYTES bytes of the kernel-argument segment with one batch of scalar loads and waits
// for them.  The compiler fetches kernel arguments lazily, group by group, next to their first use, and every group is a
// scalar-memory round trip of its own (~0.2 us each on MI355X: tools/micro/kernarg_latency.hip) in front of the dependent
// work; with the lines in the scalar cache those later fetches are hits.  Pays where the compiler's fetches are scattered through
// a latency-bound prologue (loss backward, closing kernel of the step: -0.3 us each, same-box A/B); costs a round trip of its
// own where they already leave as one batch (the GEMM kernels and the FC chains: 0 .. +0.7 us) -- not used there.
template <int BYTES>
__device__ __forceinline__ void kernarg_warm()
{
    auto kp = __builtin_amdgcn_kernarg_segment_ptr();
    constexpr int L = (BYTES + 63) / 64;
    static_assert(L >= 1 && L <= 8, "kernarg_warm: 1..8 lines");
    int d0, d1, d2, d3, d4, d5, d6, d7;
    // (one asm block: the destination registers stay reserved until the wait inside it)
    asm volatile(
        "s_load_dword %0, %8, 0x0\n\t"
        ".if %9 > 1\n\t s_load_dword %1, %8, 0x40\n\t .endif\n\t"
        ".if %9 > 2\n\t s_load_dword %2, %8, 0x80\n\t .endif\n\t"
        ".if %9 > 3\n\t s_load_dword %3, %8, 0xc0\n\t .endif\n\t"
        ".if %9 > 4\n\t s_load_dword %4, %8, 0x100\n\t .endif\n\t"
        ".if %9 > 5\n\t s_load_dword %5, %8, 0x140\n\t .endif\n\t"
        ".if %9 > 6\n\t s_load_dword %6, %8, 0x180\n\t .endif\n\t"
        ".if %9 > 7\n\t s_load_dword %7, %8, 0x1c0\n\t .endif\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6), "=&s"(d7)
        : "s"(kp), "n"(L)
        : "memory");
}

// Args: the kernel's parameter types, in order.  IMPLICIT: bytes of the implicit arguments behind them that the kernel is known
// to read (a kernel that uses gridDim / blockDim has the three block counts and three group sizes there: 24 bytes); 0 if unsure.
template <int IMPLICIT, typename... Args>
__device__ __forceinline__ void kernarg_warm_for()
{
    // the sizes alone (no padding) add up to a LOWER bound of the explicit argument area, so the last line touched -- the one
    // that holds byte `total - 4` -- lies inside the segment whatever the layout
    constexpr int total = (int)(0 + ... + sizeof(Args)) + IMPLICIT;
    static_assert(total >= 4 && IMPLICIT >= 0 && IMPLICIT <= 24, "kernarg_warm_for: bad sizes");
    constexpr int lines = (total - 4) / 64 + 1;
    kernarg_warm<(lines < 8 ? lines : 8) * 64>();
}

// sigma = max(T^2, min_sigma) with torch.max's NaN behaviour (soft_projection.py:97-99: a NaN temperature gives a NaN sigma;
// fmaxf alone would return min_sigma and hide a diverged parameter behind finite outputs)
__device__ __forceinline__ float sn_sigma(float T, float min_sigma)
{
    const float t2 = T * T;
    return t2 != t2 ? t2 : fmaxf(t2, min_sigma);
}

// element offset of channel c of point i in a (N,3) [BNC] or (3,N) [BCN] cloud
__device__ __forceinline__ int pt_off(int layout, int npts, int i, int c)
{
    return layout == SN_LAYOUT_BNC ? i * 3 + c : c * npts + i;
}

}  // namespace sn
