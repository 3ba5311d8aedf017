python -m pytest tests/test_gpu_hiergru.py -x -q 2>&1 | tail -3
for cfg in "D=128 LOW=gru" "D=256 LOW=gru"; do env N=20778 $cfg python3 scripts/train_bench.py 32 30 2>&1 | tail -1 | cut -c1-60 | sed "s/^/$cfg: /"; done
