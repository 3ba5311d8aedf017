python -m pytest tests/test_gpu_train.py tests/test_gpu_dp_train.py tests/test_gpu_hiergru.py -x -q 2>&1 | tail -12
python scripts/_probe_train_host.py
HIERTCN_GRU_BPTT_CHAIN=1 python scripts/_probe_train_host.py
