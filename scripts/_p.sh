python -m pytest tests/test_gpu_parity.py tests/test_gpu_xing_shape.py tests/test_gpu_fuzz.py tests/test_gpu_run_loop.py tests/test_gpu_gru_persist.py -x -q -m gpu 2>&1 | tail -4
for B in 128 256 512 600 1024 2048; do
    timeout 120 python3 scripts/cfg2_bench.py $B 100 2>&1 | grep "ms per batch" | head -1
    HIERTCN_SCORE_DYN_OFF=1 timeout 120 python3 scripts/cfg2_bench.py $B 100 2>&1 | grep "ms per batch" | head -1 | sed "s/^/dyn off: /"
done
