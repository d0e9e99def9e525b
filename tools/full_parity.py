#!/usr/bin/env python3
"""Full-size end-to-end parity (GPU box, ~3 min of CPU oracle time): SD v1.4 shapes, 64x64 latent, 20-step PLMS, guidance
7.5, same synthetic weights and x_T on both sides; GPU fp16 pipeline (whole-trajectory graph) vs the fp32 CPU oracle.
Prints the final-latent rel-L2 and the uint8 image statistics.  usage: python tools/full_parity.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'stable-diffusion-on-device_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle import pipeline_oracle as PO, sd_torch as S  # noqa: E402  (developer tool: the oracle is the checker here)
from sdod.amd import engine as E, weights as Wt  # noqa: E402
from sdod.amd.pipeline import Txt2Img, initial_latent  # noqa: E402

torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
cfg = E.sd14_config(64, 64)
tables = {'unet': E.UNet(cfg, 2).param_table(), 'temb': E.Temb(cfg, 1).param_table(), 'vae': E.VaeDecoder(cfg, 1).param_table(),
          'text': E.TextEncoder(cfg, 1).param_table()}
sds = {k: Wt.synthetic_state_dict(t, seed=1234 + i) for i, (k, t) in enumerate(tables.items())}
pipe = Txt2Img(state_dicts=sds, latent_hw=64)
ids_u = np.full(77, 49407, np.int64); ids_u[0] = 49406
ids_c = ids_u.copy(); ids_c[1:9] = [320, 1125, 539, 550, 18376, 6765, 320, 4558]
ctx2 = pipe.encode_tokens(ids_u, ids_c)
x_T = initial_latent(42, 0)
img = pipe.generate_graphed(ctx2, x_T, steps=20, guidance=7.5, sampler='plms').cpu().numpy()
z = pipe.sample_plms(ctx2, x_T, steps=20, guidance=7.5).cpu()
print('gpu done', flush=True)
with torch.device('meta'):
    unet, vae = S.UNetModel(), S.AutoencoderKLDecode()
unet.load_state_dict({**sds['unet'], **sds['temb']}, assign=True); vae.load_state_dict(sds['vae'], assign=True)
c = ctx2.float().cpu()
t0 = time.time()
z_ref = PO.plms_sample(unet.eval(), c[0:1], c[1:2], x_T, steps=20, scale=7.5)
print(f'cpu oracle sampler: {time.time() - t0:.0f} s', flush=True)
rel = float((z.double() - z_ref.double()).norm() / z_ref.double().norm())
img_ref = PO.decode_u8(vae.eval(), z_ref, mode=1)
d = np.abs(img.astype(np.int32) - img_ref.astype(np.int32))
print(f'final latent rel-L2 {rel:.3e}; uint8 512x512 image: max diff {int(d.max())}, mean {float(d.mean()):.4f}, '
      f'within 1 LSB {float((d <= 1).mean()):.5f}, within 2 LSB {float((d <= 2).mean()):.5f}')
