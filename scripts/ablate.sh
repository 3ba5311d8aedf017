# ablation of the scoring kernel: HIERTCN_SCORE_VARIANT=<mode><stagger>; prints launch_ms per variant
for v in 01 00 11 10 21 20; do
  HIERTCN_LIB=$PWD/hiertcn_amd/libhiertcn_gfx950_ablation.so HIERTCN_SCORE_VARIANT=$v python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant $v', 'launch_ms', d['roofline']['launch_ms'], 'TF', d['roofline']['achieved'], 'step_ms', d['ms_per_step'])"
done
