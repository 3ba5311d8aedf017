python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -3
python -m pytest tests/test_gpu_split16.py tests/test_gpu_run_loop.py tests/test_gpu_exact_mode.py tests/test_gpu_hiergru.py tests/test_gpu_dp_train.py -x -q 2>&1 | tail -5
