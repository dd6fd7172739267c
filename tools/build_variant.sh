#!/bin/bash
# tools/build_variant.sh NAME FILE.hip "-DFLAG=V ..." : a second build of libsamplenet_hip.so in which FILE.hip is compiled with extra
# flags -> tools/_ab/libsamplenet_hip_NAME.so (git-ignored, travels with gpurun).  Use: SAMPLENET_AMD_LIB=tools/_ab/libsamplenet_hip_NAME.so
# python bench.py ...   for a same-box A/B of a kernel change (box-to-box variance on the pool is ~10 %).
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; EXTRA=$3
mkdir -p tools/_ab
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Isamplenet_amd/csrc -Wall -Wno-unused-function"
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
case $SRC in
  emd.hip) F="$F -ffp-contract=off -fno-slp-vectorize";;   # hand-written packed instructions (see samplenet_amd/build.py)
  pointnet_mlp.hip|pointnet_mlp_backward.hip|fc_chain.hip|task_network.hip|capi_common.cpp) F="$F $NOPK";;
  *) F="$F $NOPK -ffp-contract=off";;
esac
/opt/rocm/bin/hipcc -x hip -c samplenet_amd/csrc/$SRC -o tools/_ab/${SRC%.*}_$NAME.o $F $EXTRA
OBJS=""
for s in capi_common pairscan geometry_ops emd pointnet_mlp pointnet_mlp_backward fc_chain task_network; do
  if [ "$s" == "${SRC%.*}" ]; then OBJS="$OBJS tools/_ab/${s}_$NAME.o"; else OBJS="$OBJS samplenet_amd/lib/$s.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_ab/libsamplenet_hip_$NAME.so $OBJS
echo tools/_ab/libsamplenet_hip_$NAME.so
