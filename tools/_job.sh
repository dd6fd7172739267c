cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_surface.py tests/test_gpu_samplenet.py tests/test_gpu_headline.py tests/test_gpu_two_ranks.py -x -q --timeout 900 2>&1 | tail -15
