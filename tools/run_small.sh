export PYTHONUNBUFFERED=1
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -x -q 2>&1 | tail -3 &&
timeout -k 10 600 python -m pytest tests/test_pipeline_gpu.py -x -q -k "deterministic" 2>&1 | tail -5 &&
for F in "--per-step-launches" ""; do timeout -k 10 400 python bench.py --no-cpu-baseline --steps 3 --warmup 1 $F 2>gpurun_out/bench_e.err | tail -1 > gpurun_out/bench_e.json; python -c "
import json; d=json.load(open('gpurun_out/bench_e.json')); print('$F', d['value'], 'img/s', d['ms_per_step'], 'ms/image', d['unet_step_ms'], 'ms/step', 'setup', d['setup_s']); f=d['roofline']['families']; print('splitk_reduce', f.get('splitk_reduce')); print('traffic', d['roofline'].get('traffic'), d['roofline'].get('mfma_util_pmc'))"; done
