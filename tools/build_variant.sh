#!/bin/bash
# tools/build_variant.sh NAME FILE.hip[,FILE2.hip...] "-DFLAG=V ..." : a second build of libsamplenet_hip.so in which the FILEs are compiled with extra
# flags -> tools/_ab/libsamplenet_hip_NAME.so (git-ignored, travels with gpurun).  Use: SAMPLENET_AMD_LIB=tools/_ab/libsamplenet_hip_NAME.so
# python bench.py ...   for a same-box A/B of a kernel change (box-to-box variance on the pool is ~10 %).
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; EXTRA=$3
mkdir -p tools/_ab
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Isamplenet_amd/csrc -Wall -Wno-unused-function"
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
OBJS=""
for s in capi_common pairscan geometry_ops emd pointnet_mlp pointnet_mlp_backward fc_chain task_network; do
  hit=0
  for one in ${SRC//,/ }; do
    if [ "$s" == "${one%.*}" ]; then hit=1; fi
  done
  if [ $hit == 1 ]; then
    FF="$F"
    case $s in
      emd) FF="$F -ffp-contract=off -fno-slp-vectorize";;   # hand-written packed instructions (see samplenet_amd/build.py)
      pointnet_mlp|pointnet_mlp_backward|fc_chain|task_network) FF="$F $NOPK";;
      capi_common) FF="$F $NOPK";;
      *) FF="$F $NOPK -ffp-contract=off";;
    esac
    src=samplenet_amd/csrc/$s.hip; [ -f $src ] || src=samplenet_amd/csrc/$s.cpp
    /opt/rocm/bin/hipcc -x hip -c $src -o tools/_ab/${s}_$NAME.o $FF $EXTRA 2>&1 | grep -v "not a recognized feature" || true &
    OBJS="$OBJS tools/_ab/${s}_$NAME.o"
  else
    OBJS="$OBJS samplenet_amd/lib/$s.o"
  fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_ab/libsamplenet_hip_$NAME.so $OBJS
echo tools/_ab/libsamplenet_hip_$NAME.so
