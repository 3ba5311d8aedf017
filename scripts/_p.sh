export N=20778 S=10 L=20 D=256
for B in 4096 8192; do
  for mx in 64 1000000; do
    HIERTCN_SKINNY_MAX_ROWS=$mx timeout 120 python3 scripts/train_bench.py $B 10 ragged 2>&1 | tail -1 | cut -c1-75 | sed "s/^/max $mx: /"
  done
done
for mx in 64 1000000; do
HIERTCN_SKINNY_MAX_ROWS=$mx LOW=gru D=128 timeout 120 python3 scripts/train_bench.py 256 20 ragged 2>&1 | tail -1 | cut -c1-75 | sed "s/^/gru d128 256 lanes max $mx: /"
done
