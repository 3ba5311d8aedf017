python -m pytest tests/test_gpu_gmm.py -x -q 2>&1 | tail -15
