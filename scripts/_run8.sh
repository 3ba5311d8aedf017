bash scripts/profile_hiergru.sh 2>&1 | tail -70
python -m pytest tests -m gpu -x -q 2>&1 | tail -8
python bench.py > gpurun_out/bench_r4_b.json 2> gpurun_out/bench_r4_b.err; tail -3 gpurun_out/bench_r4_b.err
