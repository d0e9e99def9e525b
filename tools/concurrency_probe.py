#!/usr/bin/env python3
"""Developer experiment (GPU box): does running the cond and uncond halves of the CFG batch as TWO concurrent batch-1
graph replays (two streams) beat ONE batch-2 replay?  Kernel launch/drain phases of one chain could overlap the
other chain's work.  usage: python tools/concurrency_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stable-diffusion-on-device_amd'))
import torch  # noqa: E402
from sdod.amd import engine as E, weights as Wt  # noqa: E402

cfg = E.sd14_config(64, 64)


def make(batch):
    g = E.UNet(cfg, batch)
    g.load_state_dict(Wt.synthetic_state_dict(g.param_table(), seed=1, dtype=torch.float16))
    g.finalize()
    g.x.normal_(); g.temb.normal_(); g.ctx.normal_()
    g.execute(False); g.execute(True); g.execute(True)
    torch.cuda.synchronize()
    return g


def timed(fn, iters=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


g2 = make(2)
print('one batch-2 replay            : %.3f ms' % timed(lambda: g2.execute(True, True)), flush=True)
ga, gb = make(1), make(1)
print('one batch-1 replay            : %.3f ms' % timed(lambda: ga.execute(True, True)), flush=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    with torch.cuda.stream(s1):
        ga.execute(True, True)
    with torch.cuda.stream(s2):
        gb.execute(True, True)


print('two batch-1 replays, 2 streams: %.3f ms' % timed(both), flush=True)


def serial():
    ga.execute(True, True); gb.execute(True, True)


print('two batch-1 replays, 1 stream : %.3f ms' % timed(serial), flush=True)
