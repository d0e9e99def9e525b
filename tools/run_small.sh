export PYTHONUNBUFFERED=1
echo "== cold autotune" && timeout -k 10 500 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>gpurun_out/bench_cold.err | tail -1 | tee gpurun_out/bench_cold.json &&
echo "== hot autotune" && SDOD_AUTOTUNE=hot timeout -k 10 500 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>gpurun_out/bench_hot.err | tail -1 | tee gpurun_out/bench_hot.json &&
timeout -k 10 300 python tools/unet_profile.py unet --top 60 > gpurun_out/unet_prof6.txt 2>&1
