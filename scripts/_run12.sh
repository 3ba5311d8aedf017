python scripts/_probe_train_host.py
python -m pytest tests/test_gpu_train.py tests/test_gpu_dp_train.py tests/test_gpu_hiergru.py tests/test_gpu_run_loop.py tests/test_gpu_gru_persist.py tests/test_gpu_shims.py -x -q 2>&1 | tail -5
