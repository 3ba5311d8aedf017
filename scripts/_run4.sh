python -m pytest tests/test_gpu_hiergru.py -x -q 2>&1 | tail -15
for D in 128 64; do
echo "d=$D tile"; D=$D LOW=gru python scripts/cfg2_bench.py 32 2>&1 | tail -2 | head -1
echo "d=$D persistent"; HIERTCN_LGRU_TILE=0 D=$D LOW=gru python scripts/cfg2_bench.py 32 2>&1 | tail -2 | head -1
done
