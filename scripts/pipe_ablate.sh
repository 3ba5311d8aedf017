# A/B of the pipelined scoring kernel: HIERTCN_PIPE_VARIANT = 0 v3, 1 count only, 2 none, 3 softmax sum only, 4 no reference check, 5 pipelined epilogue
for v in ${VARIANTS:-0 5 0 5}; do
  HIERTCN_PIPE_VARIANT=$v python3 scripts/score_diag.py ${ROWS:-1228800} 300000 2>&1 | tail -1 | sed "s/^/pipe variant $v: /"
done
