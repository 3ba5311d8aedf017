# Every number quoted in DESIGN.md / README.md, one command each (run on an MI355X box from the repo root; ~10 minutes).
# "Previous forms" lines run superseded kernels: those exist in the ablation build only (hiertcn_amd/build.py: libhiertcn_gfx950_ablation.so), loaded
# through HIERTCN_LIB; the other HIERTCN_* variables below are mapped onto hiertcn_tuning_t when the Engine is created (hiertcn_amd/_lib.py).
set -u
ABL="HIERTCN_LIB=$PWD/hiertcn_amd/libhiertcn_gfx950_ablation.so"
python3 -c "import __graft_entry__ as g; g.build(); g.smoke()" | tail -1
echo "== headline (bench.py defaults: 16,384 users per evaluation step, 4096 per training step, cfg-3 shape)"; python3 bench.py 2>/dev/null | tail -1 | python3 -c "
import sys, json
b = json.loads(sys.stdin.read()); r = b['roofline']
t = b.get('train') or {}
print(f\"{b['value']:.0f} sessions/s, {b['ms_per_step']:.1f} ms/step; scoring {r['achieved']:.0f} TFLOP/s = {r['frac']:.3f} of peak, {r['frac_of_library_gemm']:.2f} x library GEMM; cpu {b['cpu_baseline']['value']:.1f} sessions/s on {b['cpu_baseline']['cores']} cores; train leg {t.get('ms_per_step')} ms/step = {t.get('sessions_per_s')} sessions/s\")"
echo "== ragged sessions (lengths ~ U{1..30})"; python3 bench.py --ragged --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys, json; b = json.loads(sys.stdin.read()); print(f\"{b['value']:.0f} sessions/s\")"
echo "== reference batching (32 lanes x 10 x <=20, XING catalogue)"; python3 scripts/cfg2_bench.py 2>&1 | tail -2
echo "== reference batching, HierGRU (the reference's default low level): d = 256, 128 (the reference's default width), 64; then the launch chain it replaces"
for d in 256 128 64; do D=$d LOW=gru python3 scripts/cfg2_bench.py 2>&1 | tail -2 | head -1 | sed "s/^/d=$d (128, 64: lgru_pipe; 256: lgru_wide): /"; done
for d in 256 128 64; do D=$d LOW=gru HIERTCN_LGRU_PIPE=0 HIERTCN_LGRU_WIDE=0 python3 scripts/cfg2_bench.py 2>&1 | tail -2 | head -1 | sed "s/^/d=$d session tiles: /"; done
echo "== reference batching before this round's small-batch forms (k_tcn_fused four sessions per workgroup, host-sized catalogue split, two-block scoring shape)"
env $ABL HIERTCN_TCN_GS2_MAX=0 HIERTCN_SCORE_DYN_OFF=1 python3 scripts/cfg2_bench.py 2>&1 | tail -2 | head -1
echo "== reference batching in the parity-grade modes (split16, fp32)"; for t in split16 fp32; do DTYPE=$t python3 scripts/cfg2_bench.py 32 100 2>&1 | tail -2 | head -1 | sed "s/^/$t: /"; done
for d in 256 128 64; do D=$d LOW=gru HIERTCN_LGRU_TILE=0 HIERTCN_LGRU_CHAIN=1 python3 scripts/cfg2_bench.py 2>&1 | tail -2 | head -1 | sed "s/^/d=$d launch chain: /"; done
echo "== split16 (the fast parity-grade mode): scoring kernel alone (614,400 x 300,000; loss + rank, top-10), then the evaluation step"; python3 scripts/split16_bench.py 4096 2>&1 | grep -v amdgpu.ids | tail -3
echo "== split16 at d = 512 (csrc/score_x3w.hip): 153,600 rows x 1,000,000 items, loss + rank then top-10; 307,200 x 100,000"
python3 scripts/split16_kernel_bench.py 512 1000000 153600 2>&1 | grep -v amdgpu.ids; python3 scripts/split16_kernel_bench.py 512 1000000 153600 10 2>&1 | grep -v amdgpu.ids; python3 scripts/split16_kernel_bench.py 512 100000 307200 2>&1 | grep -v amdgpu.ids
echo "== across-session GRU scan: chain of GEMM launches vs the persistent launch"; bash scripts/gru_ab.sh 2>&1 | grep -v "^$"
echo "== scoring kernel alone, d = 256 / 128 / 64 (2,457,600 rows x 300,000 items)"
for d in 256 128 64; do D=$d python3 scripts/score_diag.py 2457600 300000 2>&1 | tail -1 | sed "s/^/d=$d: /"; done
echo "== the clock the chip holds: same kernel on random / zero / relu operands (rocm-smi sclk, socket power)"; python3 scripts/power_probe.py 2457600 300000 4 2>&1 | grep -v amdgpu.ids | tail -4
echo "== the scoring main loop without its staging, both MFMA shapes x both epilogues (scripts/shape_bench.hip)"
[ -x scripts/build/shape_bench ] || { mkdir -p scripts/build; hipcc --offload-arch=gfx950 -O3 -fno-honor-nans -Wno-unused-result scripts/shape_bench.hip -o scripts/build/shape_bench 2>/dev/null; }
scripts/build/shape_bench 2.0 2>&1 | grep -v "epi 0 " | grep "random"
echo "== d = 512, N = 1,000,000 (bf16, then the IEEE-half build)"
D=512 python3 scripts/score_diag.py 614400 1000000 2>&1 | tail -1
DTYPE=fp16 D=512 python3 scripts/score_diag.py 614400 1000000 2>&1 | tail -1
echo "== ranked top-k flavour"; python3 scripts/topk_bench.py 2>&1 | tail -3
echo "== ranked top-k flavour at d = 512 (307,200 rows x 300,000 items)"; D=512 python3 scripts/topk_bench.py 2>&1 | tail -3
echo "== training step (1024 users), dense / ragged / previous scoring backward (G chunks + the plain GEMMs of gemm.hip) / HierGRU"
python3 scripts/train_bench.py 1024 3 2>&1 | tail -1
python3 scripts/train_bench.py 1024 3 ragged 2>&1 | tail -1
env $ABL HIERTCN_TRAIN_LEGACY=1 python3 scripts/train_bench.py 1024 3 2>&1 | tail -1
LOW=gru EVAL=1 python3 scripts/train_bench.py 1024 3 2>&1 | tail -2
echo "== training step at the reference's OWN batch (32 lanes x 10 slots x 20 steps, XING catalogue): HierTCN d = 256, HierGRU d = 128; then with the launch chains / unsplit small-batch kernels of round 3"
for cfg in "D=256" "D=128 LOW=gru" "D=256 LOW=gru"; do env N=20778 S=10 L=20 $cfg python3 scripts/train_bench.py 32 30 ragged 2>&1 | tail -1 | cut -c1-60 | sed "s/^/$cfg: /"; done
for cfg in "D=256" "D=128 LOW=gru"; do env $ABL N=20778 S=10 L=20 HIERTCN_SGRAD_DYN_OFF=1 HIERTCN_SGRAD0_SLABS_OFF=1 HIERTCN_GRU_TRAIN_CHAIN=1 HIERTCN_GRU_BPTT_CHAIN=1 HIERTCN_GRU_BPTT_UNFUSED=1 HIERTCN_GEMM_NO_BATCH=1 HIERTCN_LGRU_PIPE=0 HIERTCN_LGRU_WIDE=0 HIERTCN_LGRU_WIDE_BWD=0 HIERTCN_LGRU_REFRESH_UNFUSED=1 $cfg python3 scripts/train_bench.py 32 30 ragged 2>&1 | tail -1 | cut -c1-60 | sed "s/^/$cfg, previous forms: /"; done
echo "== the run loops end to end (evaluate_hier over a 60,000-user synthetic XING file, then 600 training steps), pipelined then blocking"; TRAIN=600 python3 scripts/run_loop_bench.py 60000 32 0 2>&1 | grep "evaluate_hier\|training loop"; HIERTCN_EVAL_SERIAL=1 python3 scripts/run_loop_bench.py 60000 32 0 2>&1 | grep "evaluate_hier"
echo "== soak of the reference-batch paths (in-launch hand-offs, device-sized grids)"; python3 scripts/soak_refbatch.py 500 50 2>&1 | grep -v amdgpu.ids | tail -5
echo "== training step at 32 lanes x 20 x 30 (cfg-3's slots and steps), XING catalogue: HierTCN d = 128 / 256, HierGRU d = 128 / 256"
for cfg in "D=128" "D=256" "D=128 LOW=gru" "D=256 LOW=gru"; do env N=20778 $cfg python3 scripts/train_bench.py 32 30 2>&1 | tail -1 | cut -c1-60 | sed "s/^/$cfg: /"; done
for cfg in "D=128 LOW=gru" "D=256 LOW=gru"; do env N=20778 HIERTCN_LGRU_TILE=0 $cfg python3 scripts/train_bench.py 32 30 2>&1 | tail -1 | cut -c1-60 | sed "s/^/$cfg, launch chains: /"; done
echo "== data-parallel code path, 2 ranks sharing the GPU over gloo (smoke: RCCL needs one GPU per rank)"
HIERTCN_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --users-per-step 512 --train-users-per-step 256 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys, json; b = json.loads(sys.stdin.read()); t = b['train']; print(b['n_gpus'], 'ranks:', round(b['value']), 'sessions/s eval;', 'train', t.get('ms_per_step'), 'ms/step', t.get('collectives'))"
echo "== single-level model_tcn: six blocks, T = 500, 256 sequences (LDS-tiled layer kernel, then HIERTCN_TCN_LAYER_V1=1) and the evaluation step"
python3 scripts/tcn_layer_bench.py single eval 2>&1 | grep -v amdgpu.ids
env $ABL HIERTCN_TCN_LAYER_V1=1 python3 scripts/tcn_layer_bench.py single eval 2>&1 | grep -v amdgpu.ids | sed "s/^/V1: /"
echo "== HierGRU low level: split-operand cells vs the fp32 cells"
LOW=gru EVAL=1 python3 scripts/train_bench.py 1024 3 2>&1 | tail -2
LOW=gru EVAL=1 HIERTCN_LGRU_FP32=1 python3 scripts/train_bench.py 1024 3 2>&1 | tail -2 | sed "s/^/fp32 cells: /"
echo "== mixture head (614,400 rows x 20,778 outputs x 5 mixtures): Gram form, then the pass over the mean tables"; python3 scripts/gmm_bench.py 2>&1 | tail -1; HIERTCN_GMM_TABLE_PASS=1 python3 scripts/gmm_bench.py 2>&1 | tail -1
echo "== exact fp32 evaluation mode: throughput, and errors at a trained model's logit scale (tests/test_gpu_exact_mode.py)"
python3 scripts/exact_mode_bench.py 256 2>&1 | grep -v amdgpu.ids
python3 -m pytest tests/test_gpu_exact_mode.py -q -m gpu -s -k xing_shape 2>&1 | grep -E "mode,|storage,|passed|failed"   # (fp32 and split16 modes)
echo "== tile-skip census (profiles/r3_score_ablation.md)"; python3 scripts/skip_tile_census.py 20778 16 2>&1 | grep -v amdgpu.ids
echo "== parity against the exact oracle on unrounded fp32 weights"; python3 tests/tools/parity_fp32_probe.py 2>&1 | grep shape | cut -c1-400
echo "== ids starting in host memory"; python3 scripts/h2d_bench.py 2>&1 | tail -3
echo "== repeated full-size launches agree (ranks, target logits bitwise; ce to 5e-5)"; python3 scripts/soak_score.py 2>&1 | grep mismatching
echo "== evaluation batch and training step against the number of lanes (DESIGN 4.9, profiles/r4_lane_sweep.txt)"; bash scripts/lane_sweep.sh 2>&1 | grep -v amdgpu.ids
echo "== fp32 TRAINING mode at the reference's batch (32 x 10 x 20, XING catalogue): HierTCN d = 256, HierGRU d = 128"
python3 scripts/fp32_train_bench.py 30 2>&1 | tail -1; LOW=gru D=128 python3 scripts/fp32_train_bench.py 30 2>&1 | tail -1
echo "== single-level models at the reference's defaults (model_gru: hidden 200 -> 256; model_tcn: six blocks), 32 users x T = 200"
python3 scripts/single_level_bench.py 2>&1 | grep single-level
echo "== mixture models' TRAINING step (csrc/gmm_train.hip): 32 lanes x 200 steps, XING catalogue, d = 256, 5 mixtures: model_tcn_gmm, model_gru_gmm, model_tcn_gmm with a learned scale"
python3 scripts/gmm_train_bench.py tcn 20 2>&1 | tail -1; python3 scripts/gmm_train_bench.py gru 5 2>&1 | tail -1; python3 scripts/gmm_train_bench.py tcn 5 learned 2>&1 | tail -1
echo "== soak of the mixture training step (repeatable forward, two handles on one trajectory, evaluation after sync)"; python3 scripts/soak_gmm_train.py tcn 20 50 2>&1 | grep -v amdgpu.ids | tail -4
echo "== ... and of model_hier's step with a mixture low level (the state carried from step to step)"; for l in tcn_gmm gru_gmm; do python3 scripts/soak_hier_gmm_train.py $l 20 50 2>&1 | grep -v amdgpu.ids | tail -4 | head -3; done
