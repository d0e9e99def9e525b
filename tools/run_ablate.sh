# Compile-time ablation of the v2 GEMM main loop (GPU box).  Build the variants first (in the container):
#   cd stable-diffusion-on-device_amd && for n in 1 4 5 7; do make $(pwd)/lib/libsdod_abl$n.so; done
# mask bits: 1 no MFMA, 2 no DMA after the prologue, 4 no LDS fragment reads
export PYTHONUNBUFFERED=1
for L in libsdod.so libsdod_abl1.so libsdod_abl4.so libsdod_abl5.so libsdod_abl7.so; do echo "== $L"; for S in "conv 320->320" "conv 640->640 @32" "ff1 320" "vae conv 512"; do timeout -k 10 120 python tools/gemm_bench.py --lib $L --tiles 6,13,14,10,21,8 --iters 20 --only "$S" 2>&1 | grep -v amdgpu | tail -1; done; done
