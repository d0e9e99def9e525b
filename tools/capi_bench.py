#!/usr/bin/env python3
"""Times the reference's C API end to end (libsdod_setup / libsdod_generate_image, DPM-Solver++ 20 steps, 512x512) on the
GPU box with synthetic weight containers written to a temporary models_dir.  usage: python tools/capi_bench.py"""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stable-diffusion-on-device_amd'))
import torch  # noqa: E402
from sdod.amd import engine as E, weights as Wt  # noqa: E402
from sdod.amd.host import LibSdod  # noqa: E402

torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
d = tempfile.mkdtemp(prefix='sdod_models_')
try:
    cfg = E.sd14_config(64, 64)
    tables = {'unet': E.UNet(cfg, 2).param_table(), 'temb': E.Temb(cfg, 1).param_table(),
              'vae_decoder': E.VaeDecoder(cfg, 1).param_table(), 'text_encoder': E.TextEncoder(cfg, 1).param_table()}
    for i, (k, t) in enumerate(tables.items()):
        Wt.save(os.path.join(d, f'{k}.sdodw'), Wt.synthetic_state_dict(t, seed=2000 + i, dtype=torch.float16))
    shutil.copy(os.path.join(ROOT, 'tests', 'golden', 'ctokenizer_synthetic.txt'), os.path.join(d, 'ctokenizer.txt'))
    print('containers written', flush=True)
    t0 = time.perf_counter()
    app = LibSdod(d + '/', latent_spatial=64, steps=20)
    assert app.status == 0, app.error()
    print(f'libsdod_setup: {time.perf_counter() - t0:.1f} s', flush=True)
    app.set_seed(7)
    for i in range(4):
        t0 = time.perf_counter()
        rc, img = app.generate('the horse riding a photograph of the astronaut', 7.5)
        assert rc == 0, app.error(rc)
        print(f'libsdod_generate_image #{i}: {(time.perf_counter() - t0) * 1e3:.1f} ms  {img.shape} {img.dtype}', flush=True)
    app.release()
finally:
    shutil.rmtree(d, ignore_errors=True)
