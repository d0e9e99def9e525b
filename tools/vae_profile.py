import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'stable-diffusion-on-device_amd'))
import torch
from sdod.amd import engine as E, weights as Wt
cfg = E.sd14_config(64, 64)
g = E.VaeDecoder(cfg, 1)
g.load_state_dict(Wt.synthetic_state_dict(g.param_table(), seed=1236))
g.finalize()
ms = g.profile(iters=3)
tab = g.op_table(); det = g.op_details()
rows = sorted(zip(ms, tab, det), key=lambda r: -r[0])
print('total ms', sum(ms), 'launches', len(ms))
for t, (lab, fl, by), d in rows[:40]:
    print(f'{t*1e3:8.1f} us  {lab:14s} {d:50s} {fl/ (t*1e-3)/1e12 if fl else 0:7.1f} TF/s {by/(t*1e-3)/1e9:8.1f} GB/s')
