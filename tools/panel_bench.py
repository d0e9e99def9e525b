#!/usr/bin/env python3
"""The transformer's short-K / wide-N Linears: ring-kernel tiles against the A-panel kernel (developer tool, GPU box only).
Times to_q|to_k|to_v (LayerNorm folded) and the GEGLU projection (LayerNorm folded) of the three attention levels, cold
weights / warm activations (what a graph replay meets), with the tile the shipped table holds, a few ring tiles, and the
A-panel tile of that K.  usage: python tools/panel_bench.py [--iters 12] [--hot]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stable-diffusion-on-device_amd'))
import torch  # noqa: E402
from sdod.amd import ops  # noqa: E402

# (name, M, C, N, geglu, ln, ring tiles to compare, panel tile)
CASES = [
    ('qkv  @64 M8192 N960  K320', 8192, 320, 960, False, True, [14, 23, 31], 53),
    ('ff1  @64 M8192 N2560 K320', 8192, 320, 2560, True, True, [14, 23, 36], 53),
    ('qkv  @32 M2048 N1920 K640', 2048, 640, 1920, False, True, [24, 14, 30], 54),
    ('ff1  @32 M2048 N5120 K640', 2048, 640, 5120, True, True, [14, 23, 20], 54),
    ('qkv  @16 M512 N3840 K1280', 512, 1280, 3840, False, True, [34, 30, 28], 55),
    ('ff1  @16 M512 N10240 K1280', 512, 1280, 10240, True, True, [20, 30, 34], 55),
    ('q2   @64 M8192 N320 K320', 8192, 320, 320, False, True, [31, 28], 53),
    ('out  @64 M8192 N320 K320 +res', 8192, 320, 320, False, False, [31, 28], 53),
    ('qkv  @96 M18432 N960 K320', 18432, 320, 960, False, True, [14, 31], 53),
    ('ff1  @96 M18432 N2560 K320', 18432, 320, 2560, True, True, [14, 23], 53),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=12)
    ap.add_argument('--hot', action='store_true', help='back-to-back launches (weights cache-resident) instead of the cold-weights timing')
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    d = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    scratch = None if a.hot else torch.zeros(512 << 20, dtype=torch.uint8, device=d)
    print(f'{"case":32s} {"GFLOP":>7s}  ' + 'tile: us (TF/s) ...', flush=True)
    for name, m, c, n, geglu, ln, ring, panel in CASES:
        if a.only and a.only not in name:
            continue
        x = (torch.randn(m, c, generator=g) + 0.5).half().to(d)
        w = (torch.randn(n, c, generator=g) * c ** -0.5).half().to(d)
        bias = torch.randn(n, generator=g).to(d)
        kw = {}
        if ln:
            w, sv, tv = ops.ln_fold(w, (1 + 0.1 * torch.randn(c, generator=g)).to(d), (0.1 * torch.randn(c, generator=g)).to(d), bias)
            bias = tv; kw['ln_s'] = sv
        if geglu:
            kw['geglu'] = True
        if '+res' in name:
            kw['residual'] = torch.randn(m, n, generator=g).half().to(d)
        fl = 2.0 * m * n * c
        row = f'{name:32s} {fl / 1e9:7.2f}  '
        for t in ring + [panel]:
            try:
                us = ops.gemm(x, w, bias, tile=t, time_iters=a.iters, cold_scratch=scratch, **kw) * 1e3
                row += f't{t}: {us:6.1f} ({fl / us / 1e6:5.0f})  '
            except Exception as ex:
                row += f't{t}: -- ({str(ex)[:40]})  '
        print(row, flush=True)


if __name__ == '__main__':
    main()
