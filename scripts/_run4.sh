python -m pytest tests/test_gpu_hiergru.py -x -q 2>&1 | tail -3
for D in 256 128 64; do
echo "d=$D tile"; D=$D LOW=gru python scripts/cfg2_bench.py 32 2>&1 | tail -2 | head -1
done
