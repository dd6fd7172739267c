for rep in 1 2; do
for v in "" ps0; do
  if [ -n "$v" ]; then export SAMPLENET_AMD_LIB=$PWD/tools/_dbg/libsamplenet_hip_$v.so; else unset SAMPLENET_AMD_LIB; fi
  echo "== variant '${v:-packed}'"
  timeout 200 python tools/pairscan_scaling.py 32 512 8192 2>&1 | grep "B="
  timeout 200 python bench.py --steps 1500 --warmup 100 --no-probes 2>/dev/null | tail -1 | cut -c1-90
done
done
unset SAMPLENET_AMD_LIB
timeout 300 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_headline.py tests/test_gpu_samplenet.py -q -m gpu -x 2>&1 | tail -3
