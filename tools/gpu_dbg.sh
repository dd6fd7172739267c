timeout 900 python -m pytest tests/test_gpu_mlp.py -q -m gpu -k twin 2>&1 | tail -12
