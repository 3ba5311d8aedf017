D=128 python scripts/cfg2_bench.py 32 300 2>/dev/null| head -1
HIERTCN_SCORE_SMALL_NB2=1 D=128 python scripts/cfg2_bench.py 32 300 2>/dev/null| head -1
LOW=gru D=128 python scripts/cfg2_bench.py 32 300 2>/dev/null| head -1
HIERTCN_SCORE_SMALL_NB2=1 LOW=gru D=128 python scripts/cfg2_bench.py 32 300 2>/dev/null| head -1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_xing_shape.py tests/test_gpu_run_loop.py tests/test_gpu_hiergru.py tests/test_single_level.py -x -q 2>&1 | tail -3
