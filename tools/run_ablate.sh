# SDOD_GEMM_DEBUG ablation of the v2 GEMM (1: no MFMA, 2: no LDS fragment reads, 4: no epilogue)
export PYTHONUNBUFFERED=1
for D in 0 4 5 7; do echo "== SDOD_GEMM_DEBUG=$D"; for S in "conv 320->320" "ff1 320" "tiny"; do SDOD_GEMM_DEBUG=$D timeout -k 10 120 python tools/gemm_bench.py --tiles 14,9,10,8 --iters 30 --only "$S" 2>&1 | grep -v amdgpu | tail -1; done; done
