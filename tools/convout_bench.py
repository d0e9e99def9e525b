import sys, os
sys.path.insert(0, 'stable-diffusion-on-device_amd'); sys.path.insert(0, 'tools')
import torch
from sdod.amd import ops
from gn_bench import graph_time
d = torch.device('cuda:0')
for (n, hw, c, co) in ((2, 64, 320, 4), (1, 512, 128, 3)):
    x = torch.randn(n, hw, hw, c).half().to(d); w = (torch.randn(co, 9 * c) / 50).half().to(d); b = torch.randn(co).to(d)
    ref = ops.gemm(x, w, b, conv=dict(stride=1))
    line = f'conv3 {n}x{hw}x{hw}x{c} -> {co}: default {graph_time(lambda: ops.gemm(x, w, b, conv=dict(stride=1)), 5):.1f}'
    for tile in (45, 44, 42, 41, 40, 38, 8, 28, 32):
        for sk in (1, 2, 3, 5):
            try:
                f = lambda: ops.gemm(x, w, b, conv=dict(stride=1), tile=tile, split_k=sk)
                o = f()
                err = float((o.float() - ref.float()).abs().max())
                line += f' | t{tile}x{sk} {graph_time(f, 5):.1f}' + ('' if err < 0.05 else f' ERR {err:.2f}')
            except Exception as e:
                line += f' | t{tile}x{sk} n/a'
    print(line)
