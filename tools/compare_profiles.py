#!/usr/bin/env python3
"""Side by side of two tools/unet_profile.py outputs (developer tool): python tools/compare_profiles.py A.txt B.txt [min_us]
Rows are (label, shape); prints launches, us each in A and B, the difference per evaluation, and the sums per kernel family."""
import re
import sys


def parse(path):
    rows = {}
    for line in open(path):
        m = re.match(r'^(\S+)\s+(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$', line.rstrip())
        if m:
            rows[(m.group(1), m.group(2).strip())] = (int(m.group(3)), float(m.group(4)))
    return rows


def family(label, shape):
    if label.startswith('attn'):
        return 'attention'
    if label.startswith('gn'):
        return 'groupnorm'
    if label.startswith('splitk'):
        return 'splitk_reduce'
    if 'rows' in shape:
        return 'linear'
    if 'conv' in shape:
        return 'conv'
    return label


def main():
    a, b = parse(sys.argv[1]), parse(sys.argv[2])
    min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    fam = {}
    print(f'{"label":16s} {"shape":44s} {"n":>3s} {"A us":>8s} {"B us":>8s} {"delta/eval":>10s}')
    for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0))[0] * a.get(k, (0, 0))[1])):
        na, ua = a.get(k, (0, 0.0)); nb, ub = b.get(k, (0, 0.0))
        d = nb * ub - na * ua
        f = fam.setdefault(family(*k), [0.0, 0.0, 0, 0])
        f[0] += na * ua; f[1] += nb * ub; f[2] += na; f[3] += nb
        if abs(d) >= min_us:
            print(f'{k[0]:16s} {k[1]:44s} {max(na, nb):3d} {ua:8.2f} {ub:8.2f} {d:+10.1f}')
    print()
    ta = tb = 0.0
    for k, (x, y, na, nb) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        print(f'{k:16s} launches {na:4d} -> {nb:4d}   {x:9.1f} -> {y:9.1f} us   {y - x:+8.1f}')
        ta += x; tb += y
    print(f'{"total":16s} {ta:9.1f} -> {tb:9.1f} us   {tb - ta:+8.1f}')


if __name__ == '__main__':
    main()
