set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python bench.py > gpurun_out/bench_r4_a.json 2> gpurun_out/bench_r4_a.err; tail -3 gpurun_out/bench_r4_a.err
