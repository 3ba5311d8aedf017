for lanes in 32 64 128 256 512 2048; do python scripts/cfg2_bench.py $lanes 100 2>/dev/null | head -1; done
