#!/bin/bash
# Debug-only library with per-workgroup phase timestamps (see tools/timeline.py for the MLP GEMM kernels,
# tools/pairscan_timeline.py for the pair scan and the loss backward).  LEVEL=2: every stamp of the geometric kernels first
# waits for the memory operations before it (serialised phases); default 1: stamps only.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_ab
L=${LEVEL:-1}
X=${EXTRA:-}          # extra flags for the MLP unit (e.g. EXTRA="-DSN_CBX_SKEW=1")
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Isamplenet_amd/csrc -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops"
G="$F -ffp-contract=off -DSN_PS_TIMELINE=$L -DSN_CS_TIMELINE=$L"
# the four MLP translation units as ONE (the stamp buffers of mlp_device.h then exist once, whichever kernel writes them)
printf '#include "pointnet_mlp.hip"\n#include "pointnet_mlp_backward.hip"\n#include "fc_chain.hip"\n#include "task_network.hip"\n' > tools/_ab/pointnet_mlp_tl.hip
/opt/rocm/bin/hipcc -x hip -c tools/_ab/pointnet_mlp_tl.hip -o tools/_ab/pointnet_mlp_tl.o $F -DSN_TIMELINE $X
/opt/rocm/bin/hipcc -x hip -c samplenet_amd/csrc/capi_common.cpp -o tools/_ab/capi_common_tl.o $F -DSN_TIMELINE
/opt/rocm/bin/hipcc -x hip -c samplenet_amd/csrc/pairscan.hip -o tools/_ab/pairscan_tl.o $G
/opt/rocm/bin/hipcc -x hip -c samplenet_amd/csrc/geometry_ops.hip -o tools/_ab/geometry_ops_tl.o $G
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_ab/libsamplenet_hip_tl.so tools/_ab/pointnet_mlp_tl.o tools/_ab/capi_common_tl.o tools/_ab/pairscan_tl.o tools/_ab/geometry_ops_tl.o samplenet_amd/lib/emd.o
