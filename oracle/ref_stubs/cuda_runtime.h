/* Empty stand-in so that the reference's host-only CPU functions in
 * classification/structural_losses/approxmatch.cpp (approxmatch_cpu, matchcost_cpu,
 * matchcostgrad_cpu, lines 17-125) compile with plain g++ in a container that has no CUDA.
 * Only the harness main() of that file touches these names; it is renamed away by the build
 * (-Dmain=...) and never called.  TEST INFRASTRUCTURE ONLY (see oracle/README.md). */
#pragma once
#include <cstddef>
enum cudaMemcpyKind { cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2 };
static inline int cudaMalloc(void *p, size_t) { *(void **)p = nullptr; return 0; }
template <class T> static inline int cudaMalloc(T **p, size_t) { *p = nullptr; return 0; }
static inline int cudaMemcpy(void *, const void *, size_t, cudaMemcpyKind) { return 0; }
static inline int cudaMemset(void *, int, size_t) { return 0; }
static inline int cudaDeviceSynchronize() { return 0; }
static inline int cudaFree(void *) { return 0; }
