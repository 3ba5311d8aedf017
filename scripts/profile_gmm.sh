# rocprofv3 kernel-trace summary of the mixture loss, Gram form (default) and table pass.  usage (GPU box): bash scripts/profile_gmm.sh <tag>
TAG=${1:-r3}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_gmm_$TAG
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gram -- python3 scripts/gmm_bench.py > $OUT/gram.log 2>&1
export HIERTCN_GMM_TABLE_PASS=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/table -- python3 scripts/gmm_bench.py > $OUT/table.log 2>&1
for d in gram table; do echo "== $d"; tail -1 $OUT/$d.log | cut -c1-200; f=$(find $OUT/$d -name '*kernel_stats.csv' | head -1); head -8 "$f" | cut -c1-200; cp "$f" $OUT/${d}_kernel_stats.csv; done
