# rocprofv3 summary of the layer-at-a-time low level at the single-level model_tcn shape: kernel stats, then PMC passes
# in their own runs.  usage (on the GPU box): bash scripts/profile_tcn_layer.sh <tag>
TAG=${1:-r3}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_tcn_layer_$TAG
mkdir -p $OUT
B="python3 scripts/tcn_layer_bench.py single"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1
python3 scripts/tcn_layer_bench.py single eval > $OUT/bench.txt 2>&1
HIERTCN_LIB=$PWD/hiertcn_amd/libhiertcn_gfx950_ablation.so HIERTCN_TCN_LAYER_V1=1 python3 scripts/tcn_layer_bench.py single eval > $OUT/bench_v1.txt 2>&1
python3 scripts/summarise_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/bench.txt $OUT/bench_v1.txt $OUT/summary.txt
