LOW=gru D=128 python scripts/cfg2_bench.py 32 300 2>/dev/null| head -1
HIERTCN_LGRU_PIPE=0 LOW=gru D=128 python scripts/cfg2_bench.py 32 300 2>/dev/null| head -1
LOW=gru D=64 python scripts/cfg2_bench.py 32 300 2>/dev/null| head -1
HIERTCN_LGRU_PIPE=0 LOW=gru D=64 python scripts/cfg2_bench.py 32 300 2>/dev/null| head -1
LOW=gru D=128 python scripts/cfg2_bench.py 128 300 2>/dev/null| head -1
HIERTCN_LGRU_PIPE=0 LOW=gru D=128 python scripts/cfg2_bench.py 128 300 2>/dev/null| head -1
python scripts/_probe_train_host.py | tail -1
HIERTCN_LGRU_PIPE=0 python scripts/_probe_train_host.py | tail -1
python -m pytest tests/test_gpu_hiergru.py tests/test_gpu_run_loop.py tests/test_gpu_train.py tests/test_gpu_gmm.py -x -q 2>&1 | tail -5
