python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -k "two_pairs" 2>&1 | tail -15
