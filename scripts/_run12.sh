python -m pytest tests/test_gpu_hiergru.py tests/test_gpu_gmm.py -x -q 2>&1 | tail -4
python scripts/_probe_train_host.py
HIERTCN_LGRU_REFRESH_UNFUSED=1 python scripts/_probe_train_host.py | tail -1
