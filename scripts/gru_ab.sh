# A/B of the across-session GRU scan: chain of GEMM launches (HIERTCN_GRU_CHAIN=1) vs the persistent launch, reference batching
for lanes in 32 64 128 256; do
  echo "== lanes $lanes"
  echo -n "chain:      "; HIERTCN_GRU_CHAIN=1 python scripts/cfg2_bench.py $lanes 300 2>&1 | tail -1
  echo -n "persistent: "; HIERTCN_GRU_PERSIST_MAX_LANES=1024 python scripts/cfg2_bench.py $lanes 300 2>&1 | tail -1
done
