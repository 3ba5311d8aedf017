python -m pytest tests/test_gpu_hiergru.py tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -3
LOW=gru EVAL=1 timeout 300 python3 scripts/train_bench.py 1024 3 2>&1 | tail -2 | cut -c1-150
LOW=gru EVAL=1 HIERTCN_LGRU_TILE_MAX=32768 HIERTCN_LGRU_DEFER_MAX_GIB=1 timeout 300 python3 scripts/train_bench.py 1024 3 2>&1 | tail -2 | cut -c1-150 | sed "s/^/old limits: /"
LOW=gru EVAL=1 timeout 300 python3 scripts/train_bench.py 4096 2 2>&1 | tail -2 | cut -c1-150
LOW=gru EVAL=1 HIERTCN_LGRU_TILE_MAX=32768 HIERTCN_LGRU_DEFER_MAX_GIB=1 timeout 300 python3 scripts/train_bench.py 4096 2 2>&1 | tail -2 | cut -c1-150 | sed "s/^/old limits: /"
