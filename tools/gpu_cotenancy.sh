# The two-process co-tenancy runs of the round-end bundle alone, bounded (run as: gpurun --timeout 150 -- 'bash tools/gpu_cotenancy.sh r05').
# tools/micro/pk_fma_cotenancy must have been built in the container (hipcc --offload-arch=gfx950 -O2; see its header).
R=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
T="timeout -s KILL"
{
  echo "# tools/cotenancy_stress.py on one MI355X: two processes at once (parent + child), every pass compared with the process's first pass bit for bit"
  echo "## product build (compiler-packed fp32 off; emd.o: hand-written packed instructions, destinations disjoint from their sources)"
  $T 40 python tools/cotenancy_stress.py fwd 6000 2>&1 | grep cotenancy_stress
  $T 50 python tools/cotenancy_stress.py step 60 2>&1 | grep cotenancy_stress
  $T 40 python tools/cotenancy_stress.py emd 4000 2>&1 | grep cotenancy_stress
  if [ -x tools/micro/pk_fma_cotenancy ]; then
    echo "## tools/micro/pk_fma_cotenancy (stand-alone: v_pk_fma_f32 with destination = source pair, exact-integer recurrence, every lane checked)"
    (cd tools/micro; $T 30 ./pk_fma_cotenancy alias 3000 & $T 30 ./pk_fma_cotenancy alias 3000; wait)
    (cd tools/micro; $T 30 ./pk_fma_cotenancy plain 3000 & $T 30 ./pk_fma_cotenancy plain 3000; wait)
  fi
} > $OUT/cotenancy_stress.txt 2>&1
cat $OUT/cotenancy_stress.txt
