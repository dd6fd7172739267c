#!/bin/bash
# Debug-only library with per-workgroup phase timestamps in the MLP GEMM kernels (see tools/timeline.py).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_dbg
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Isamplenet_amd/csrc -DSN_TIMELINE -Wno-unused-function"
/opt/rocm/bin/hipcc -x hip -c samplenet_amd/csrc/pointnet_mlp.hip -o tools/_dbg/pointnet_mlp_tl.o $F
/opt/rocm/bin/hipcc -x hip -c samplenet_amd/csrc/capi_common.cpp -o tools/_dbg/capi_common_tl.o $F
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_dbg/libsamplenet_hip_tl.so tools/_dbg/pointnet_mlp_tl.o tools/_dbg/capi_common_tl.o samplenet_amd/lib/pairscan.o samplenet_amd/lib/geometry_ops.o samplenet_amd/lib/emd.o
