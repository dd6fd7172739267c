// resident_spin.hip -- a stand-in for the RESIDENT kernels of a collective library beside the sampler's step (VERDICT r5 #7c): n
// workgroups that occupy n CUs of XCD 0 (observed placement: block b -> XCD b % 8; blocks with b % 8 != 0 leave at once), each
// holding `lds_bytes` of LDS and 256 threads, spinning on a device flag until it is raised (or a wall-clock bound passes: it can
// never hang the device).  tests/test_gpu_cotenancy.py builds it with hipcc on the GPU box and runs the FC chain launches -- 8 and
// 16 co-resident workgroups of 137 KB LDS on the same XCD -- beside it.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o libresident_spin.so resident_spin.hip
#include <hip/hip_runtime.h>

__global__ void __launch_bounds__(256) resident_spin_kernel(const unsigned *flag, unsigned long long max_ticks, unsigned *arrived)
{
    extern __shared__ float lds[];
    if (blockIdx.x & 7) return;
    if (threadIdx.x == 0) {
        lds[0] = 1.f;  // (the allocation is real: the CU's LDS is taken)
        atomicAdd(arrived, 1u);
        const unsigned long long t0 = wall_clock64();  // 100 MHz
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            if (wall_clock64() - t0 > max_ticks) break;
            __builtin_amdgcn_s_sleep(32);
        }
    }
    __syncthreads();
}

// n workgroups on XCD 0, each with lds_bytes of dynamic LDS; they leave when *flag != 0 or after max_ms milliseconds.
extern "C" int resident_spin_start(int n, int lds_bytes, const unsigned *flag, unsigned *arrived, int max_ms, void *stream)
{
    if (hipFuncSetAttribute((const void *)resident_spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return 1;
    hipLaunchKernelGGL(resident_spin_kernel, dim3(8 * n), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, flag,
                       (unsigned long long)max_ms * 100000ull, arrived);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
