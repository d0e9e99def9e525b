# SQ counters of the attention kernel (GPU box): where do the wave cycles go?  two --pmc passes (8 SQ slots each)
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
rm -rf gpurun_out/attn_pmc1 gpurun_out/attn_pmc2
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC --output-format csv -d gpurun_out/attn_pmc1 -o a -- python3 tools/attn_bench.py --reps 2 > gpurun_out/attn_pmc1.txt 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/attn_pmc2 -o b -- python3 tools/attn_bench.py --reps 2 > gpurun_out/attn_pmc2.txt 2>&1
python3 - <<'PY'
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob('gpurun_out/attn_pmc*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if 'attn_kernel' in r['Kernel_Name']:
            key = (r['Kernel_Name'][:60], r.get('Grid_Size', ''), r.get('LDS_Block_Size', ''))
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in sorted(agg.items()):
    print(k, {n: round(sum(v) / len(v)) for n, v in sorted(c.items())})
PY
