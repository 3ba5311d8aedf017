# A/B of the across-session GRU scan: chain of GEMM launches (HIERTCN_GRU_CHAIN=1) vs the persistent launch, reference batching.
# The persistent launch is built for up to 96 lanes (hiertcn_tuning_t.gru_persist_max_lanes: its LDS image and residency check); wider batches
# take the chain whatever is asked for, so the A/B stops there.
for lanes in 32 64 96; do
  echo "== lanes $lanes"
  echo -n "chain:      "; HIERTCN_GRU_CHAIN=1 python scripts/cfg2_bench.py $lanes 300 2>&1 | tail -1
  echo -n "persistent: "; HIERTCN_GRU_PERSIST_MAX_LANES=96 python scripts/cfg2_bench.py $lanes 300 2>&1 | tail -1
done
