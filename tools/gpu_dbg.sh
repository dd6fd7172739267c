SAMPLENET_AMD_LIB=$PWD/tools/_dbg/libsamplenet_hip_tl.so timeout 200 python tools/fc_chain_timeline.py 2>&1 | grep -v amdgpu.ids | cut -c1-75
