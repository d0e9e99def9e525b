#!/usr/bin/env python3
"""Fused attention on the UNet's shapes (packed qkv / cached kv layouts, as the launch list calls it), timed as a launch
list replays it: `reps` launches in one device graph.  usage: python tools/attn_bench.py [--reps 10]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stable-diffusion-on-device_amd')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch  # noqa: E402

# (B, heads, Lq, Lk, d, count per UNet evaluation)
SHAPES = [(2, 8, 4096, 4096, 40, 5), (2, 8, 1024, 1024, 80, 5), (2, 8, 256, 256, 160, 5), (2, 8, 64, 64, 160, 1),
          (2, 8, 4096, 77, 40, 5), (2, 8, 1024, 77, 80, 5), (2, 8, 256, 77, 160, 5), (2, 8, 64, 77, 160, 1),
          (4, 8, 4096, 4096, 40, 0), (2, 5, 9216, 9216, 64, 0)]


def main():
    global ops, graph_time
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--lib', default='', help='developer build in lib/ to load instead of libsdod.so')
    args = ap.parse_args()
    if args.lib:
        os.environ['SDOD_LIBSDOD'] = args.lib
    from sdod.amd import ops
    from gn_bench import graph_time
    d = torch.device('cuda:0')
    tot = 0.0
    P = ctypes.c_void_p
    print(f'{"shape":34s} {"us":>8s} {"TF/s":>8s} {"x count":>8s}')
    for b, h, lq, lk, dh, cnt in SHAPES:
        c = h * dh
        out = torch.empty(b, lq, c, dtype=torch.float16, device=d)
        if lq == lk:
            qkv = torch.randn(b, lq, 3 * c).half().to(d)
            q, k, v, ldq, ldk = qkv.data_ptr(), qkv.data_ptr() + 2 * c, qkv.data_ptr() + 4 * c, 3 * c, 3 * c
        else:
            qt = torch.randn(b, lq, c).half().to(d)
            kv = torch.randn(b, lk, 2 * c).half().to(d)
            q, k, v, ldq, ldk = qt.data_ptr(), kv.data_ptr(), kv.data_ptr() + 2 * c, c, 2 * c
        us = graph_time(lambda: ops.attention_strided(P(q), P(k), P(v), P(out.data_ptr()), b, h, lq, lk, dh, ldq, ldk, ldk, c, dh ** -0.5), args.reps)
        fl = 4.0 * b * h * lq * lk * dh
        tot += us * cnt
        print(f'B{b} h{h} lq{lq:5d} lk{lk:5d} d{dh:4d}        {us:8.2f} {fl / us / 1e6:8.1f} {us * cnt:8.1f}', flush=True)
    print(f'sum over one UNet evaluation: {tot / 1e3:.3f} ms')


if __name__ == '__main__':
    main()
