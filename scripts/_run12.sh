python -m pytest tests/test_gpu_train.py tests/test_gpu_dp_train.py tests/test_single_level.py -x -q 2>&1 | tail -4
python scripts/_probe_train_host.py
