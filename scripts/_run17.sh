DTYPE=split16 python scripts/cfg2_bench.py 32 200 2>/dev/null| head -1
HIERTCN_SCORE_DYN_OFF=1 DTYPE=split16 python scripts/cfg2_bench.py 32 200 2>/dev/null| head -1
DTYPE=split16 LOW=gru D=128 python scripts/cfg2_bench.py 32 200 2>/dev/null| head -1
DTYPE=fp32 python scripts/cfg2_bench.py 32 200 2>/dev/null| head -1
python -m pytest tests/test_gpu_split16.py tests/test_gpu_exact_mode.py tests/test_gpu_run_loop.py -x -q 2>&1 | tail -4
