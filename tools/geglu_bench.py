import sys
sys.path.insert(0, 'stable-diffusion-on-device_amd'); sys.path.insert(0, 'tools')
import torch
from sdod.amd import ops
from gn_bench import graph_time
d = torch.device('cuda:0')
for m, c in ((8192, 320), (2048, 640), (512, 1280)):
    x = torch.randn(m, c).half().to(d); w = (torch.randn(8 * c, c) / c ** 0.5).half().to(d); b = torch.randn(8 * c).to(d)
    s = torch.randn(8 * c).to(d)
    line = f'geglu+ln M{m} N{8*c} K{c}:'
    for tile in (14, 25, 53, 54, 61, 62, 23, 20):
        try:
            f = lambda: ops.gemm(x, w, b, ln_s=s, geglu=True, tile=tile)
            f(); line += f' t{tile} {graph_time(f, 10):.1f}'
        except Exception as e:
            line += f' t{tile} n/a'
    print(line)
