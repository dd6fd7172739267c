cd $GRAFT_REPO_ROOT
timeout -s KILL 300 python -m pytest tests/test_gpu_rccl.py -x -q --timeout 300 2>&1 | grep -E "passed|failed|Error|assert|^E" | head -20
