bash scripts/pmc_topk.sh > gpurun_out/r4_topk_pmc.txt 2>&1
tail -8 gpurun_out/r4_topk_pmc.txt | cut -c1-400
bash scripts/profile_train.sh r4 > gpurun_out/r4_train_pmc.log 2>&1
tail -50 gpurun_out/r4_train_pmc.log | cut -c1-200
