python -m pytest tests -x -q -m gpu 2>&1 | tail -3
echo "== scripts/lane_sweep.sh (HierTCN d=256)"; bash scripts/lane_sweep.sh
echo "== scripts/lane_sweep.sh LOW=gru (HierGRU d=256)"; bash scripts/lane_sweep.sh LOW=gru
