# L2 hit rate / fabric traffic of single GEMM shapes (GPU box): separate rocprofv3 --pmc passes over tools/gemm_bench.py
# usage: bash tools/run_l2_probe.sh "<shape substring>" "<tiles>"
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
SHAPE="${1:-conv 320->320}"; TILES="${2:-13,23,31,28}"
rm -rf gpurun_out/l2probe_h gpurun_out/l2probe_f
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d gpurun_out/l2probe_h -o h -- python3 tools/gemm_bench.py --only "$SHAPE" --tiles $TILES --iters 4 > gpurun_out/l2probe_h.txt 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/l2probe_f -o f -- python3 tools/gemm_bench.py --only "$SHAPE" --tiles $TILES --iters 4 > gpurun_out/l2probe_f.txt 2>&1 &&
python3 - <<'PY'
import collections, csv, glob
for tag in ('h', 'f'):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(f'gpurun_out/l2probe_{tag}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(path)):
            agg[r['Kernel_Name'][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, c in agg.items():
        if 'gemm' not in k:
            continue
        m = {n: sum(v) / len(v) for n, v in c.items()}
        extra = ''
        if 'TCC_HIT_sum' in m:
            extra = f" hit rate {m['TCC_HIT_sum'] / max(1.0, m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}"
        if 'FETCH_SIZE' in m:
            extra = f" fabric fetch {2 * 1024 * m['FETCH_SIZE'] / 1e6:.1f} MB/launch (x2 gfx950 correction)"
        print(k, {n: round(v) for n, v in m.items()}, extra, 'launches', len(next(iter(c.values()))))
PY
