# Round profile set (GPU box).  Every process builds its launch lists from the shipped tune table (tune/gfx950.tune), so the
# un-profiled bench line, the kernel trace and the three PMC passes describe ONE launch list.  The kernel trace replays the
# trajectory graph exactly as the bench does; the PMC passes run the launch list eagerly (counters are per dispatch) and are
# separate passes, never combined with trace domains.  usage: bash tools/run_profile.sh [tag]   (tag default r03)
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
TAG="${1:-r03}"
O=gpurun_out/$TAG
rm -rf $O && mkdir -p $O
timeout -k 10 600 python3 bench.py --steps 5 --warmup 2 > $O/bench_line.json 2> $O/bench_line.err && echo "bench done" &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_line_under_rocprof.json 2> $O/trace.err && echo "trace done" &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- python3 bench.py --no-cpu-baseline --no-hip-graph --steps 1 --warmup 0 > $O/pmc_f.json 2> $O/pmc_f.err && echo "fetch pass done" &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- python3 bench.py --no-cpu-baseline --no-hip-graph --steps 1 --warmup 0 > $O/pmc_w.json 2> $O/pmc_w.err && echo "write pass done" &&
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_m -o m -- python3 bench.py --no-cpu-baseline --no-hip-graph --steps 1 --warmup 0 > $O/pmc_m.json 2> $O/pmc_m.err && echo "mfma pass done" &&
python3 tools/pmc_summary.py $(find $O/trace -name '*kernel_stats.csv' | head -1) $(find $O/pmc_f -name '*counter_collection.csv' | head -1) $(find $O/pmc_w -name '*counter_collection.csv' | head -1) $(find $O/pmc_m -name '*counter_collection.csv' | head -1) 16 > $O/pmc_summary.json && echo "summary done" &&
cp $(find $O/trace -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv
ls -la $O | head -30
