export PYTHONUNBUFFERED=1 TMPDIR=/tmp
export SDOD_TUNE_CACHE=$PWD/gpurun_out/tune_cache.txt
echo "tune cache lines: $(wc -l < $SDOD_TUNE_CACHE)" &&
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py --no-cpu-baseline --no-hip-graph --steps 1 --warmup 0 > gpurun_out/pmc_w.json 2> gpurun_out/pmc_w.err && echo "write pass done" &&
timeout -k 10 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err && cut -c1-200 gpurun_out/bench_final.json
