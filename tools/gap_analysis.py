#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV (the bench's trajectory-graph replays): how much of a
UNet step is launch gaps rather than kernels, and after / before which kernels the gaps are longest.
usage: python tools/gap_analysis.py <kernel_trace.csv> [max_gap_us=20]   (gaps above max_gap are host-side pauses: skipped)"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r'(gemm_glds_kernel|conv_halo_kernel|gemm_apanel_kernel|attn_kernel|splitk_reduce_\w*kernel|gn_\w+|\w+_kernel)', name)
    return m.group(1) if m else name[:40]


def main():
    path = sys.argv[1]
    max_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
    rows.sort()
    after, before = defaultdict(list), defaultdict(list)
    gaps, busy = [], 0
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        g = (s1 - e0) / 1e3
        busy += e0 - s0
        if g > max_gap:
            continue
        gaps.append(g)
        after[n0].append(g)
        before[n1].append(g)
    gaps.sort()
    n = len(gaps)
    print(f'{len(rows)} dispatches, {n} gaps <= {max_gap} us: mean {sum(gaps) / n:.2f} us, median {gaps[n // 2]:.2f}, p90 {gaps[int(n * 0.9)]:.2f}, '
          f'sum {sum(gaps) / 1e3:.2f} ms against {busy / 1e6:.2f} ms of kernel time ({100 * sum(gaps) * 1e3 / busy:.1f} %)')
    for title, d in (('gap AFTER kernel', after), ('gap BEFORE kernel', before)):
        print(f'\n{title:34s} {"n":>6s} {"mean us":>8s} {"median":>8s} {"total ms":>9s}')
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:12]:
            v = sorted(v)
            print(f'{k:34s} {len(v):6d} {sum(v) / len(v):8.2f} {v[len(v) // 2]:8.2f} {sum(v) / 1e3:9.3f}')


if __name__ == '__main__':
    main()
