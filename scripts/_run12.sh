python scripts/_probe_train_host.py
HIERTCN_GEMM_NO_BATCH=1 python scripts/_probe_train_host.py
LOW=gru D=128 python scripts/train_bench.py 32 30 ragged | tail -1 | cut -c1-80
HIERTCN_GEMM_NO_BATCH=1 LOW=gru D=128 python scripts/train_bench.py 32 30 ragged | tail -1 | cut -c1-80
LOW=gru D=128 python scripts/train_bench.py 1024 3 | tail -1 | cut -c1-80
HIERTCN_GEMM_NO_BATCH=1 LOW=gru D=128 python scripts/train_bench.py 1024 3 | tail -1 | cut -c1-80
python -m pytest tests/test_gpu_train.py tests/test_gpu_dp_train.py tests/test_gpu_hiergru.py -x -q 2>&1 | tail -4
