export PYTHONUNBUFFERED=1
for HW in 1024 256 1024 256; do SDOD_GN_SMALL_HW=$HW timeout -k 10 400 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/bench_gn$HW.json; python -c "
import json; d=json.load(open('gpurun_out/bench_gn$HW.json')); print('gn_small up to hw $HW:', d['value'], 'img/s', d['unet_step_ms'], 'ms/step')"; done
