# Round profile of `bench.py` (default config): kernel-trace stats, then PMC passes in their own runs
# (MI355X_MICROARCH.md 'rocprofv3 PMC slots': FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage (on the GPU box): bash scripts/profile_round.sh <tag> <git head>      (the head is passed in: the box has no .git)
# -> gpurun_out/prof_<tag>_<head>/{summary.json, score_traffic.json, trace/, pmc_*/}; copy summary.json, the kernel_stats csv and score_traffic.json into profiles/
TAG=${1:-r1}
HEAD=${2:-unknown}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT  # every rocprofv3 call below sits under `timeout`: a failed counter request once hung for the whole gpurun limit
OUT=gpurun_out/prof_${TAG}_$HEAD   # one directory per (tag, head): gpurun MERGES what a call wrote into the local gpurun_out/, and rocprofv3 names its files by pid -- a second run into the same directory would be summed into the first when summarise_profile.py is re-run locally
mkdir -p $OUT
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-clock-probe --exact-steps 2 --exact-warmup 1"
BP="$B --no-train --no-gmm --no-reference-batch"   # the counter passes serialise every kernel: only the legs whose kernels are read
# (the trace pass skips the reference-batch / run-loop legs: their 32-lane launches of the SAME kernel names would be averaged into the bench-shape launches the roofline is about)
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --no-reference-batch > $OUT/bench_trace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -- $BP > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- $BP > $OUT/pmc_write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_sq -- $BP > $OUT/pmc_sq.log 2>&1
python3 bench.py --steps 4 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench_trace.log | cut -c1-300
python3 scripts/summarise_profile.py $OUT $TAG $HEAD
