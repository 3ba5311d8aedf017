set -x
python -m pytest tests/test_gpu_parity.py -x -q -k "topk or ragged or forward_eval" 2>&1 | tail -5
python scripts/topk_bench.py
HIERTCN_TOPK_V3=1 python scripts/topk_bench.py
HIERTCN_LIB=$PWD/hiertcn_amd/libhiertcn_gfx950_vb.so python scripts/topk_bench.py
