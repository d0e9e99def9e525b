#!/usr/bin/env python3
"""Per-launch profile of one graph (developer tool, GPU box): python tools/unet_profile.py [unet|vae|text] [--hw 64] [--top 60]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stable-diffusion-on-device_amd'))
import torch  # noqa: E402
from sdod.amd import engine as E, weights as Wt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('kind', nargs='?', default='unet')
ap.add_argument('--hw', type=int, default=64)
ap.add_argument('--batch', type=int, default=2)
ap.add_argument('--top', type=int, default=70)
ap.add_argument('--model', default='sd14', choices=['sd14', 'sd21'], help='sd21: SD v2.1-768 UNet shapes (config 5), use --hw 96')
ap.add_argument('--quant', action='store_true', help='keep the conv / linear weights affine uint8 in HBM (config 5: sdod_model_config.weight_quant)')
ap.add_argument('--quant-auto', action='store_true', help='weight_quant = 2: uint8 codes streamed only by the blocks with <= 128 rows, the rest dequantised at load')
a = ap.parse_args()
a.quant = a.quant or a.quant_auto
cfg = (E.sd21_config if a.model == 'sd21' else E.sd14_config)(a.hw, a.hw)
if a.quant:
    cfg.weight_quant = 2 if a.quant_auto else 1
g = {'unet': E.UNet, 'vae': E.VaeDecoder, 'text': E.TextEncoder}[a.kind](cfg, a.batch if a.kind != 'vae' else 1)
sd = Wt.synthetic_state_dict(g.param_table(), seed=1, dtype=torch.float32 if a.quant else torch.float16)
g.load_state_dict(Wt.quantize_state_dict(sd) if a.quant else sd)
g.finalize()
g.execute()
import time  # noqa: E402
g.execute(True); g.execute(True); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.execute(True, True) if a.kind == 'unet' else g.execute(True)
torch.cuda.synchronize()
print(f'{a.kind} {a.model} hw{a.hw} batch {g.batch}: {(time.perf_counter() - t0) * 100:.3f} ms per hipGraph replay, {g.stats()["flops"] / 1e12:.3f} TFLOP')
ms = g.profile(iters=5)
tab = g.op_table(); det = g.op_details()
rows = sorted(zip(ms, tab, det), key=lambda r: -r[0])
tot = sum(ms)
print(f'{a.kind}: {len(ms)} launches, {tot:.3f} ms eager (event-timed)')
agg = {}
for t, (lab, fl, by), d in zip(ms, tab, det):
    k = (lab, d)
    e = agg.setdefault(k, [0.0, 0, fl, by]); e[0] += t; e[1] += 1
print(f'{"label":18s} {"shape":44s} {"n":>3s} {"us each":>8s} {"ms tot":>7s} {"TF/s":>7s} {"GB/s":>7s}')
for (lab, d), (t, n, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:a.top]:
    us = 1e3 * t / n
    print(f'{lab:18s} {d:44s} {n:3d} {us:8.1f} {t:7.3f} {fl / us / 1e6 if fl else 0:7.1f} {by / us / 1e3 if by else 0:7.1f}')
