for mx in 2048 100000; do
for D in 128 256; do
echo "== max $mx d=$D"; HIERTCN_LGRU_TILE_MAX=$mx D=$D LOW=gru EVAL=1 python3 scripts/train_bench.py 1024 3 2>&1 | grep -v amdgpu | tail -2 | cut -c1-90
done; done
