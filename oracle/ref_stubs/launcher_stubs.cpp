/* Link-time stand-ins for the device launchers the reference host files declare but whose
 * definitions live in .cu files that cannot be built here (no CUDA).  They are never called:
 * only the reference's CPU functions are exercised through oracle/_ref.
 *   - approxmatch.cpp:126-128   approxmatchLauncher / matchcostLauncher / matchcostgradLauncher
 *   - chamfer_distance.cpp:4-24 ChamferDistanceKernelLauncher / ChamferDistanceGradKernelLauncher
 * TEST INFRASTRUCTURE ONLY. */
#include <cstdlib>
#ifdef STUB_APPROXMATCH
void approxmatchLauncher(int, int, int, const float *, const float *, float *) { abort(); }
void matchcostLauncher(int, int, int, const float *, const float *, const float *, float *) { abort(); }
void matchcostgradLauncher(int, int, int, const float *, const float *, const float *, float *) { abort(); }
#endif
#ifdef STUB_CHAMFER
int ChamferDistanceKernelLauncher(const int, const int, const float *, const int, const float *,
                                  float *, int *, float *, int *) { abort(); }
int ChamferDistanceGradKernelLauncher(const int, const int, const float *, const int, const float *,
                                      const float *, const int *, const float *, const int *,
                                      float *, float *) { abort(); }
#endif
