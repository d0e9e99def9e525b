#!/usr/bin/env python3
"""LayerNorm kernel timing on the UNet's token shapes (developer tool, GPU box only): back-to-back launches, HIP events."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stable-diffusion-on-device_amd'))
import torch  # noqa: E402
from sdod.amd import ops  # noqa: E402

d = torch.device('cuda:0')
for m, c in ((8192, 320), (2048, 640), (512, 1280), (18432, 320), (4608, 640), (1152, 1280), (154, 768), (154, 1024)):
    x = torch.randn(m, c, device=d).half(); w = torch.ones(c, device=d); b = torch.zeros(c, device=d); y = torch.empty_like(x)
    for _ in range(5):
        ops.layer_norm(x, w, b, 1e-5, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.layer_norm(x, w, b, 1e-5, out=y)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f'layer_norm {m:6d} x {c:4d}: {us:7.2f} us  {4.0 * m * c / us / 1e3:7.1f} GB/s')
