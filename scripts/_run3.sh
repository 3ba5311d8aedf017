HIERTCN_TOPK_V3=1 python scripts/topk_bench.py
HIERTCN_TOPK_V3=1 HIERTCN_LIB=$PWD/hiertcn_amd/libhiertcn_gfx950_vb.so python scripts/topk_bench.py
