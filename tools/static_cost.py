#!/usr/bin/env python3
"""What the once-per-prompt part of the UNet graph costs (cross-attention K/V projection + the folded cross-attention's
per-prompt matrices): replay time with and without the static launch list.  python tools/static_cost.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stable-diffusion-on-device_amd'))
import torch  # noqa: E402
from sdod.amd import engine as E, weights as Wt  # noqa: E402

cfg = E.sd14_config(64, 64)
g = E.UNet(cfg, 2)
g.load_state_dict(Wt.synthetic_state_dict(g.param_table(), seed=1, dtype=torch.float16))
g.finalize()
g.execute(); g.execute(True); g.execute(True, True); torch.cuda.synchronize()
for name, fn in (('with static list', lambda: g.execute(True, False)), ('without', lambda: g.execute(True, True))) * 2:
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize(); print(f'{name:18s} {(time.perf_counter() - t0) * 50:.3f} ms per evaluation')
