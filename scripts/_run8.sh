python -m pytest tests -m gpu -x -q 2>&1 | tail -8
python bench.py > gpurun_out/bench_r4_c.json 2> gpurun_out/bench_r4_c.err; tail -3 gpurun_out/bench_r4_c.err
