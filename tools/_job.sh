cd $GRAFT_REPO_ROOT
timeout -s KILL 400 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_headline.py -x -q --timeout 300 2>&1 | tail -3
