timeout -s KILL 600 python -m pytest tests/test_gpu_emd.py tests/test_gpu_cotenancy.py -m gpu -q -k "emd" 2>&1 | tail -3
timeout -s KILL 200 python tools/cotenancy_stress.py emd 3000 2>&1 | grep cotenancy_stress
