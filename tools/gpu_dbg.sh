timeout 200 python tools/eager_profile.py 2>&1 | grep -v amdgpu.ids
timeout 300 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_samplenet.py tests/test_gpu_headline.py -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15
