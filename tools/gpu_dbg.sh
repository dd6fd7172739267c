mkdir -p gpurun_out/r03
SAMPLENET_AMD_LIB=$PWD/tools/_dbg/libsamplenet_hip_tl.so timeout 200 python tools/fc_chain_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03/fc_chain_timeline.txt; tail -3 gpurun_out/r03/fc_chain_timeline.txt
