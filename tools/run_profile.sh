# Round profile set (GPU box).  Order matters: the tune cache is written by an UN-profiled run (timings under a profiler
# would poison the picks), the kernel trace replays the hipGraph, the PMC passes run the launch list eagerly (separate
# passes per counter group, never combined with trace domains).
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
export SDOD_TUNE_CACHE=$PWD/gpurun_out/tune_cache.txt
rm -f $SDOD_TUNE_CACHE
timeout -k 10 400 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/prof_warm.json 2> gpurun_out/prof_warm.err && echo "tune cache lines: $(wc -l < $SDOD_TUNE_CACHE)" &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_d -o r01d -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/bench_under_rocprof.json 2> gpurun_out/prof_d.err && echo "trace done" &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python3 bench.py --no-cpu-baseline --no-hip-graph --steps 1 --warmup 0 > gpurun_out/pmc_f.json 2> gpurun_out/pmc_f.err && echo "fetch pass done" &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py --no-cpu-baseline --no-hip-graph --steps 1 --warmup 0 > gpurun_out/pmc_w.json 2> gpurun_out/pmc_w.err && echo "write pass done" &&
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_m -o m -- python3 bench.py --no-cpu-baseline --no-hip-graph --steps 1 --warmup 0 > gpurun_out/pmc_m.json 2> gpurun_out/pmc_m.err && echo "mfma pass done"
unset SDOD_TUNE_CACHE
timeout -k 10 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err && cut -c1-200 gpurun_out/bench_final.json
