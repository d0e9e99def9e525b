export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -x -q 2>&1 | tail -4 &&
timeout -k 10 500 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>gpurun_out/bench_c.err | tail -1 > gpurun_out/bench_c.json && cut -c1-300 gpurun_out/bench_c.json
