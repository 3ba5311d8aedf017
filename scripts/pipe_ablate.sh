# A/B of the pipelined scoring kernel (C = 256): HIERTCN_PIPE_VARIANT = 0 v3, 1 count only, 2 none, 3 softmax sum only, 4 no reference check,
# 5 two-tile kernel (production until v6), 6 deeper ring, 7-9 packed fp32, a/b fewer LDS reads (timing only), c/d four tiles, e two tiles on the
# padded image, f four tiles + seamless boundary (= production), h = f without the stage barrier (timing only); anything else = production
# needs the ablation build (python -m hiertcn_amd.build -> libhiertcn_gfx950_ablation.so), loaded through HIERTCN_LIB:
for v in ${VARIANTS:-0 5 0 5}; do
  HIERTCN_LIB=$PWD/hiertcn_amd/libhiertcn_gfx950_ablation.so HIERTCN_PIPE_VARIANT=$v python3 scripts/score_diag.py ${ROWS:-1228800} 300000 2>&1 | tail -1 | sed "s/^/pipe variant $v: /"
done
