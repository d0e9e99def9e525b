#!/usr/bin/env python3
"""Per-kernel summary of one round's rocprofv3 passes (tools/run_profile.sh): average duration from the kernel-trace stats,
HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes, MFMA utilisation from the SQ pass.
gfx950 corrections (MI355X_MICROARCH.md): FETCH_SIZE and WRITE_SIZE count 1024-byte units (of 64-byte requests summed),
FETCH_SIZE reports half of the bytes of wide coalesced streams (x2); GRBM_GUI_ACTIVE is summed over the 8 XCDs;
SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, so  MFMA utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024).
usage: python tools/pmc_summary.py <kernel_stats.csv> <fetch_counter_collection.csv> <write_...csv> <sq_...csv> [top_n]"""
import collections
import csv
import json
import sys


def per_kernel(path, wanted):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] in wanted:
            agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    return agg


def main():
    stats, fpath, wpath, mpath = sys.argv[1:5]
    top = int(sys.argv[5]) if len(sys.argv) > 5 else 12
    rows = list(csv.DictReader(open(stats)))
    total = sum(float(r['TotalDurationNs']) for r in rows)
    f = per_kernel(fpath, {'FETCH_SIZE'}); w = per_kernel(wpath, {'WRITE_SIZE'})
    m = per_kernel(mpath, {'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'GRBM_GUI_ACTIVE'})
    mean = lambda xs: sum(xs) / len(xs) if xs else None
    out = []
    for r in rows[:top]:
        k = r['Name']
        fetch = mean(f[k]['FETCH_SIZE']); write = mean(w[k]['WRITE_SIZE'])
        busy = mean(m[k]['SQ_VALU_MFMA_BUSY_CYCLES']); gui = mean(m[k]['GRBM_GUI_ACTIVE'])
        avg_us = float(r['AverageNs']) / 1e3
        e = {'kernel': k, 'calls': int(r['Calls']), 'avg_us': round(avg_us, 2), 'share_of_gpu_time': round(float(r['TotalDurationNs']) / total, 4),
             'hbm_fetch_bytes_per_launch': round(2 * 1024 * fetch) if fetch is not None else None,
             'hbm_write_bytes_per_launch': round(1024 * write) if write is not None else None,
             'mfma_util': round(busy / (gui / 8 * 1024), 4) if busy is not None and gui else None}
        if e['hbm_fetch_bytes_per_launch'] is not None and e['hbm_write_bytes_per_launch'] is not None:
            e['hbm_gbs_at_avg_duration'] = round((e['hbm_fetch_bytes_per_launch'] + e['hbm_write_bytes_per_launch']) / (avg_us * 1e-6) / 1e9, 1)
        out.append(e)
    print(json.dumps({'source': 'tools/run_profile.sh', 'kernels': out}, indent=1))


if __name__ == '__main__':
    main()
