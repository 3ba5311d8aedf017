# The four runs of the scaling curve (BASELINE.json: sessions/sec at 1, 2, 4 and 8 MI355X of ONE node), exactly as the driver
# launches them: one rank per GPU over RCCL (backend "nccl"), weak scaling (bench.py's default users per step per GPU: 16,384).  Each prints ONE JSON line whose
# `multi_gpu` object carries backend / rccl_ranks / distinct_devices / per-rank launch times.
# usage (on a node with >= 8 GPUs): bash scripts/scale.sh [steps] [warmup] > scale.jsonl
STEPS=${1:-6}; WARMUP=${2:-2}; PORT=${PORT:-29611}
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$(dirname "$0")/.."
python bench.py --gpus 1 --steps $STEPS --warmup $WARMUP
for N in 2 4 8; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $N --steps $STEPS --warmup $WARMUP
done
