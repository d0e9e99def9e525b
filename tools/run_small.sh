export PYTHONUNBUFFERED=1
for sp in 0 6 10 20 40; do echo "== split $sp"; timeout -k 10 120 python tools/gemm_bench.py --tiles 3,8,2,7,11,5,9,13,14,16 --iters 20 --split $sp --only "1280 @8" 2>&1 | grep -v amdgpu; done
for sp in 0 3 6 12; do echo "== split $sp"; timeout -k 10 120 python tools/gemm_bench.py --tiles 3,8,2,7,11,5,9,13,14,16 --iters 20 --split $sp --only "1280->1280 @16" 2>&1 | grep -v amdgpu; done
