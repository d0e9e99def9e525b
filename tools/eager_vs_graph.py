import sys, os, time
sys.path.insert(0, 'stable-diffusion-on-device_amd')
import torch
from sdod.amd import engine as E, weights as Wt
cfg = E.sd14_config(64, 64)
g = E.UNet(cfg, 2)
g.load_state_dict(Wt.synthetic_state_dict(g.param_table(), seed=1, dtype=torch.float16))
g.finalize()
g.execute(); g.execute(True); g.execute(True, True); torch.cuda.synchronize()
for name, fn in (('graph', lambda: g.execute(True, True)), ('eager', lambda: g.execute(False, True)), ('graph', lambda: g.execute(True, True)), ('eager', lambda: g.execute(False, True))):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); print(name, '%.3f ms' % ((time.perf_counter() - t0) * 50))
