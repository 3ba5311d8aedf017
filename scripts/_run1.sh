set -x
python -m pytest tests/test_gpu_split16.py -x -q 2>&1 | tail -30
python -m pytest tests/test_gpu_exact_mode.py -x -q 2>&1 | tail -30
python scripts/split16_bench.py 4096 2>&1 | tail -10
cp gpurun_out/exact_mode_errors.json gpurun_out/r4_exact_mode_errors.json 2>/dev/null
