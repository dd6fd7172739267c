timeout 600 python -m pytest tests/test_gpu_mlp.py -q -m gpu -k "fc_chain_backward_equals" 2>&1 | grep -A25 "^    def test_fc_chain_backward\|Error" | tail -60
