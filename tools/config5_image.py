#!/usr/bin/env python3
"""BASELINE config 5 end to end on one MI355X (developer tool, GPU box only): SD v2.1-768 shapes -- OpenCLIP-H text tower,
96x96 latent, 20-step v-prediction PLMS around the batch-2 (cond + uncond) UNet with guidance, VAE decode 96 -> 768, uint8
image -- with the UNet's conv / linear weights kept affine uint8 in HBM (the reference's `quantize=8` path, todlc.py:105-108)
or, with --fp16, the fp16 image of the same weights.  Synthetic seeded weights; one trajectory-graph replay per image.

usage: python tools/config5_image.py [--fp16] [--images 3]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'stable-diffusion-on-device_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--fp16', action='store_true', help='dequantise once at load (fp16 weights in HBM) instead of streaming uint8 codes')
ap.add_argument('--images', type=int, default=3)
ap.add_argument('--hw', type=int, default=96, help='latent size (96 = 768 px)')
args = ap.parse_args()

from sdod.amd import engine as E, weights as Wt  # noqa: E402
from sdod.amd.pipeline import Txt2Img, device_latent  # noqa: E402

t0 = time.time()
cfg = E.sd21_config(args.hw, args.hw)
tables = {'unet': E.UNet(cfg, 2).param_table(), 'temb': E.Temb(cfg, 1).param_table(),
          'vae': E.VaeDecoder(cfg, 1).param_table(), 'text': E.TextEncoder(cfg, 1).param_table()}
sds = {k: Wt.synthetic_state_dict(t, seed=2100 + i) for i, (k, t) in enumerate(tables.items())}
for k in ('unet', 'temb'):
    sds[k] = Wt.quantize_state_dict(sds[k])     # an int8-weight checkpoint: QuantU8 tensors
pipe = Txt2Img(state_dicts=sds, images_per_gpu=1, latent_hw=args.hw, model='sd21', weight_quant=not args.fp16)
ids = np.zeros((77,), dtype=np.int64); ids[0] = 49406; ids[1:6] = (320, 1125, 539, 320, 2368); ids[6:] = 49407
ids_u = np.full((77,), 49407, dtype=np.int64); ids_u[0] = 49406


def one_image(k):
    x_T = device_latent(42, k, shape=(4, args.hw, args.hw), device='cuda:0')   # Philox keyed by (seed, image index)
    ctx2 = pipe.encode_tokens(ids_u, ids)
    return pipe.generate_graphed(ctx2, x_T, steps=20, guidance=7.5, sampler='plms')


img = one_image(0)                                # captures the trajectory graph (set-up)
torch.cuda.synchronize()
print(f'set-up {time.time() - t0:.1f} s; image {tuple(img.shape)} {img.dtype}; UNet {pipe.unet.stats()}', flush=True)
one_image(1); torch.cuda.synchronize()
t1 = time.perf_counter()
for k in range(args.images):
    img = one_image(2 + k)
torch.cuda.synchronize()
dt = (time.perf_counter() - t1) / args.images
assert img.dtype == torch.uint8 and torch.isfinite(img.float()).all()
print(f'config 5, {"fp16" if args.fp16 else "uint8"} UNet weights in HBM, {args.hw * 8} px, 20-step PLMS: {dt * 1e3:.1f} ms per image, {1.0 / dt:.2f} images/s '
      f'({dt * 1e3 / 21:.2f} ms per UNet evaluation incl. VAE / text share)')
