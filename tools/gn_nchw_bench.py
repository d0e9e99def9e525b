#!/usr/bin/env python3
"""sdod.EfficientGN on torch's default (NCHW) layout: the NCHW kernel against the former route (torch .contiguous() transpose to
channels-last + NHWC kernel + view back), HIP-event timed over `reps` back-to-back calls.  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stable-diffusion-on-device_amd'))
import torch  # noqa: E402
from sdod.amd import ops  # noqa: E402

SHAPES = [(2, 320, 64, 64), (2, 640, 32, 32), (2, 1280, 16, 16), (2, 1280, 8, 8), (1, 128, 512, 512), (1, 512, 64, 64)]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    d = torch.device('cuda:0')
    print(f'{"shape":22s} {"MB (r+w)":>9s} {"nchw us":>9s} {"GB/s":>8s} {"transpose+nhwc us":>18s}')
    for shp in SHAPES:
        x = torch.randn(shp, device=d).half()
        w = torch.ones(shp[1], device=d); b = torch.zeros(shp[1], device=d)
        n, c = shp[0], shp[1]

        def old():
            xl = x.reshape(n, c, -1).permute(0, 2, 1).contiguous()
            return ops.group_norm_nhwc(xl, 32, w, b, 1e-5, True).permute(0, 2, 1).reshape(shp)

        t_new = timed(lambda: ops.group_norm_nchw(x, 32, w, b, 1e-5, True))
        t_old = timed(old)
        mb = 2 * x.numel() * 2 / 1e6
        print(f'{str(shp):22s} {mb:9.1f} {t_new:9.1f} {mb * 1e6 / (t_new * 1e-6) / 1e9:8.0f} {t_old:18.1f}', flush=True)


if __name__ == '__main__':
    main()
