timeout 600 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_headline.py tests/test_gpu_samplenet.py -q -m gpu 2>&1 | tail -4
SAMPLENET_AMD_LIB=$PWD/tools/_dbg/libsamplenet_hip_tl.so timeout 200 python tools/fc_chain_timeline.py 2>&1 | grep -v amdgpu.ids | sed -n 18,50p | cut -c1-70
for rep in 1 2 3; do
for v in "" base; do
  if [ -n "$v" ]; then export SAMPLENET_AMD_LIB=$PWD/tools/_dbg/libsamplenet_hip_$v.so; else unset SAMPLENET_AMD_LIB; fi
  echo "variant '${v:-new}' $(timeout 200 python bench.py --steps 2000 --warmup 100 --no-probes 2>/dev/null | tail -1 | cut -c1-70)"
done
done
