// pk_fma_cotenancy.hip -- stand-alone reproducer for DESIGN.md 6c: does a two-pass packed fp32 instruction whose DESTINATION
// pair aliases a SOURCE pair give wrong low halves when a second process shares the GPU (wave save / restore between the passes)?
//
//   hipcc --offload-arch=gfx950 -O2 -o pk_fma_cotenancy tools/micro/pk_fma_cotenancy.hip
//   ./pk_fma_cotenancy alias 2000 & ./pk_fma_cotenancy alias 2000 ; wait      # two processes at once: the suspect form
//   ./pk_fma_cotenancy plain 2000 & ./pk_fma_cotenancy plain 2000 ; wait      # destination disjoint from the sources
//   ./pk_fma_cotenancy alias 2000                                             # one process alone
// Every launch runs the same exact-integer recurrence (x <- x * 1 + 1 on small integers: no rounding, any deviation is a wrong
// value, not a reordering) in every lane of a grid that fills the chip, and compares ALL lanes with the closed form.  Prints
// the number of launches with a wrong lane and which half of the pair was wrong.  Exit code 1 if any launch deviated.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

typedef float f2v __attribute__((ext_vector_type(2)));

template <bool ALIAS>
__global__ void __launch_bounds__(256) pk_kernel(float *out, int iters)
{
    f2v acc = {(float)(threadIdx.x & 7), (float)(threadIdx.x & 3) + 100.f};
    const f2v one = {1.f, 1.f};
    f2v tmp;
    for (int i = 0; i < iters; ++i) {
        if (ALIAS) {
            // the pattern the compiler's SLP-packed code produced: v_pk_fma_f32 v[2:3], v[30:31], v[2:3], ... (dst = src1)
            asm volatile("v_pk_fma_f32 %0, %1, %0, %1" : "+v"(acc) : "v"(one));
        } else {
            asm volatile("v_pk_fma_f32 %0, %2, %1, %2\n\tv_pk_fma_f32 %1, %2, %0, %2" : "=&v"(tmp), "+v"(acc) : "v"(one));
            ++i;  // (two steps per trip, each into a pair disjoint from its sources)
        }
    }
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    out[2 * t] = acc.x, out[2 * t + 1] = acc.y;
}

int main(int argc, char **argv)
{
    const bool alias = argc < 2 || !strcmp(argv[1], "alias");
    const int launches = argc > 2 ? atoi(argv[2]) : 2000, iters = 4096, blocks = 256 * 8, threads = 256;
    const size_t n = (size_t)blocks * threads;
    float *d = nullptr;
    if (hipMalloc(&d, 2 * n * sizeof(float)) != hipSuccess) return 2;
    std::vector<float> h(2 * n);
    int bad_launches = 0;
    long long bad_lo = 0, bad_hi = 0;
    for (int l = 0; l < launches; ++l) {
        if (alias) hipLaunchKernelGGL(pk_kernel<true>, dim3(blocks), dim3(threads), 0, 0, d, iters);
        else hipLaunchKernelGGL(pk_kernel<false>, dim3(blocks), dim3(threads), 0, 0, d, iters);
        if (hipMemcpy(h.data(), d, 2 * n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 2;
        long long lo = 0, hi = 0;
        for (size_t t = 0; t < n; ++t) {
            const int tx = (int)(t % threads);
            lo += h[2 * t] != (float)((tx & 7) + iters);
            hi += h[2 * t + 1] != (float)((tx & 3) + 100 + iters);
        }
        bad_launches += (lo + hi) != 0;
        bad_lo += lo, bad_hi += hi;
    }
    printf("pk_fma_cotenancy %s: %d of %d launches deviated (wrong low halves %lld, wrong high halves %lld)\n", alias ? "alias" : "plain",
           bad_launches, launches, bad_lo, bad_hi);
    return bad_launches ? 1 : 0;
}
