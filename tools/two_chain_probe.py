#!/usr/bin/env python3
"""Would two independent batch-1 UNet chains on two streams beat one batch-2 chain?  (developer probe, GPU box only)
The cond / uncond halves of a guided step are independent; two half-size kernel chains could fill each other's ramp-up / drain
gaps.  Prints ms per guided step for: one batch-2 graph; two batch-1 graphs replayed concurrently on two streams."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stable-diffusion-on-device_amd'))
import torch  # noqa: E402
from sdod.amd import engine as E, weights as Wt  # noqa: E402


def build(batch):
    cfg = E.sd14_config(64, 64)
    g = E.UNet(cfg, batch)
    g.load_state_dict(Wt.synthetic_state_dict(g.param_table(), seed=1234))
    g.finalize()
    return g


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


def main():
    g2 = build(2)
    a, b = build(1), build(1)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    print(f'one batch-2 graph: {timeit(lambda: g2.execute()):.3f} ms per guided step', flush=True)

    def two():
        with torch.cuda.stream(s1):
            a.execute()
        with torch.cuda.stream(s2):
            b.execute()
    print(f'two batch-1 graphs on two streams: {timeit(two):.3f} ms per guided step', flush=True)
    print(f'one batch-1 graph alone: {timeit(lambda: a.execute()):.3f} ms', flush=True)


if __name__ == '__main__':
    main()
