python -m pytest tests/test_gpu_run_loop.py tests/test_run_loop.py tests/test_gpu_dp_train.py -x -q 2>&1 | tail -12
python scripts/run_loop_bench.py 60000 32 0 2>&1 | grep evaluate_hier
HIERTCN_EVAL_SERIAL=1 python scripts/run_loop_bench.py 60000 32 0 2>&1 | grep evaluate_hier
python scripts/run_loop_bench.py 60000 128 0 2>&1 | grep evaluate_hier
HIERTCN_EVAL_SERIAL=1 python scripts/run_loop_bench.py 60000 128 0 2>&1 | grep evaluate_hier
