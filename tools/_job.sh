cd $GRAFT_REPO_ROOT
echo "## product build (hand-written packed instructions in emd.o, destination disjoint from the sources), two processes"
timeout 600 python tools/cotenancy_stress.py emd 1500
echo "## round-3 sources of emd.hip with the COMPILER's packed code (plain -O3), two processes"
SAMPLENET_AMD_LIB=$PWD/tools/_ab/libsamplenet_hip_emdcpk.so timeout 600 python tools/cotenancy_stress.py emd 1500
