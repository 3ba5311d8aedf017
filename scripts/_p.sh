python -m pytest tests/test_gpu_train.py tests/test_gpu_fuzz.py tests/test_gpu_dp_train.py -x -q -m gpu 2>&1 | tail -3
export N=20778 S=10 L=20 D=256
for B in 128 256 512 1024 2048; do
  timeout 120 python3 scripts/train_bench.py $B 20 ragged 2>&1 | tail -1 | cut -c1-75
  HIERTCN_SGRAD_DYN_MAX_BLOCKS=127 timeout 120 python3 scripts/train_bench.py $B 20 ragged 2>&1 | tail -1 | cut -c1-75 | sed "s/^/  blocks<128: /"
done
