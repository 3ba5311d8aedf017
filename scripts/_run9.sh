python -m pytest tests/test_gpu_train.py tests/test_gpu_dp_train.py -x -q 2>&1 | tail -8
N=20778 D=64 python scripts/train_bench.py 1024 5
HIERTCN_SGRAD64_OFF=1 N=20778 D=64 python scripts/train_bench.py 1024 5
N=300000 D=64 python scripts/train_bench.py 1024 3
HIERTCN_SGRAD64_OFF=1 N=300000 D=64 python scripts/train_bench.py 1024 3
