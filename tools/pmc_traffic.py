#!/usr/bin/env python3
"""Per-launch HBM traffic of one kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected separately, as
MI355X_MICROARCH.md prescribes: they do not fit one pass).  gfx950 corrections from that guide: both counters are in
KiB-like units of 1024 B, and FETCH_SIZE reports exactly half of the bytes of a wide coalesced stream -> doubled.
usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel-substring> [last_n]"""
import csv
import json
import sys


def per_dispatch(path, counter, sub):
    vals = []
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter and sub in r['Kernel_Name']:
            vals.append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    vals.sort()
    return [v for _, v in vals]


def main():
    fpath, wpath, sub = sys.argv[1], sys.argv[2], sys.argv[3]
    last = int(sys.argv[4]) if len(sys.argv) > 4 else 400
    f = per_dispatch(fpath, 'FETCH_SIZE', sub)[-last:]
    w = per_dispatch(wpath, 'WRITE_SIZE', sub)[-last:]
    fetch = 2.0 * 1024.0 * sum(f) / len(f)
    write = 1024.0 * sum(w) / len(w)
    print(json.dumps({'kernel_substring': sub, 'dispatches_averaged': [len(f), len(w)], 'fetch_bytes_per_launch': fetch,
                      'write_bytes_per_launch': write, 'hbm_bytes_per_launch': fetch + write,
                      'corrections': 'FETCH_SIZE*1024*2 (gfx950 reports half of wide coalesced reads), WRITE_SIZE*1024'}))


if __name__ == '__main__':
    main()
