for v in "" 0 2 02; do
echo "== HIERTCN_SGRAD_ONEWAVE='$v'"
env HIERTCN_SGRAD_ONEWAVE="$v" python scripts/train_bench.py 1024 3 2>&1 | grep -v amdgpu | tail -2
done
HIERTCN_SGRAD_ONEWAVE=02 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -3
