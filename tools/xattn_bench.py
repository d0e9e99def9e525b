#!/usr/bin/env python3
"""The two GEMMs of the folded cross-attention on the UNet's shapes, per tile, with and without the softmax epilogue / LayerNorm
fold (developer tool, GPU box): python tools/xattn_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stable-diffusion-on-device_amd')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch  # noqa: E402
from sdod.amd import ops  # noqa: E402
from gn_bench import graph_time  # noqa: E402

d = torch.device('cuda:0')
B = 2
for hw, c in ((64, 320), (32, 640), (16, 1280), (8, 1280)):
    rows = hw * hw
    x = torch.randn(B * rows, c).half().to(d)
    w1 = (torch.randn(B, 640, c) / c ** 0.5).half().to(d); s1 = torch.randn(B, 640).to(d); t1 = torch.randn(B, 640).to(d)
    w2 = (torch.randn(B, c, 640) / 25).half().to(d); bo = torch.randn(c).to(d)
    p = torch.empty(B * rows, 640, dtype=torch.float16, device=d); out = torch.empty(B * rows, c, dtype=torch.float16, device=d)
    line = f'{hw}x{hw} C{c}: scores'
    for tile in (31, 48, 56, 57, 58, 59, 60):
        for sm, ln in ((80, True),):
            try:
                f = lambda: ops.gemm(x, w1, t1, ln_s=s1 if ln else None, rows_per_img=rows, softmax_cols=sm, tile=tile, out=p)
                f()
                line += f' | t{tile}{"s" if sm else ""}{"l" if ln else ""} {graph_time(f, 10):.1f}'
            except Exception as e:  # noqa: BLE001
                line += f' | t{tile} n/a'
    print(line)
    line = f'{hw}x{hw} C{c}: out   '
    for tile in (0, 27, 28, 31, 46, 32, 8, 13):
        f = lambda: ops.gemm(p, w2, bo, residual=x, rows_per_img=rows, tile=tile, out=out)
        f()
        line += f' | t{tile} {graph_time(f, 10):.1f}'
    print(line)
