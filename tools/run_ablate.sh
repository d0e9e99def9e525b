# SDOD_GEMM_DEBUG ablation of the v2 GEMM (1: no MFMA, 2: no DMA after the prologue, 4: no LDS fragment reads)
export PYTHONUNBUFFERED=1
for SP in 1 4 12 30; do for D in 0 5 7; do echo "== split $SP SDOD_GEMM_DEBUG=$D"; for S in "conv 1280->1280 @8" "small M512"; do SDOD_GEMM_DEBUG=$D timeout -k 10 120 python tools/gemm_bench.py --split $SP --tiles 3,8,14,17,20 --iters 30 --only "$S" 2>&1 | grep -v amdgpu | tail -1; done; done; done
