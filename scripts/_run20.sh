python -m pytest tests/test_gpu_run_loop.py tests/test_data_loader.py tests/test_run_loop.py -x -q 2>&1 | tail -3
TRAIN=600 python scripts/run_loop_bench.py 60000 32 0 2>&1 | grep "evaluate_hier\|training loop"
HIERTCN_PY_PACKER=1 python scripts/run_loop_bench.py 60000 32 0 2>&1 | grep evaluate_hier
python scripts/run_loop_bench.py 60000 128 0 2>&1 | grep evaluate_hier
python scripts/run_loop_bench.py 60000 512 0 2>&1 | grep evaluate_hier
LOW=gru D=128 python scripts/run_loop_bench.py 60000 32 0 2>&1 | grep evaluate_hier
python scripts/loader_bench.py 20000 32 4096 2>&1 | grep direct
