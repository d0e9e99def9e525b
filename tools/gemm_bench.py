#!/usr/bin/env python3
"""Micro-benchmark of the GEMM / conv kernel on the UNet + VAE layer shapes (developer tool, GPU box only).
usage: python tools/gemm_bench.py [--tiles 0,1,2,3] [--iters 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stable-diffusion-on-device_amd'))
import torch  # noqa: E402
from sdod.amd import ops  # noqa: E402

# (name, kind, params)  rows: (M, N, K); conv: (n, h, w, cin, cout, stride, ups)
SHAPES = [
    ('conv 320->320 @64', 'conv', (2, 64, 64, 320, 320, 1, False)),
    ('conv 640->640 @32', 'conv', (2, 32, 32, 640, 640, 1, False)),
    ('conv 1280->1280 @16', 'conv', (2, 16, 16, 1280, 1280, 1, False)),
    ('conv 1280->1280 @8', 'conv', (2, 8, 8, 1280, 1280, 1, False)),
    ('conv 2560->1280 @8', 'conv', (2, 8, 8, 2560, 1280, 1, False)),
    ('conv 960->320 @64', 'conv', (2, 64, 64, 960, 320, 1, False)),
    ('conv 1920->640 @32', 'conv', (2, 32, 32, 1920, 640, 1, False)),
    ('upconv 640->640 @32->64', 'conv', (2, 32, 32, 640, 640, 1, True)),
    ('qkv 320 @64', 'rows', (8192, 960, 320)),
    ('proj 320 @64', 'rows', (8192, 320, 320)),
    ('ff1 320 @64', 'rows', (8192, 2560, 320)),
    ('ff2 320 @64', 'rows', (8192, 320, 1280)),
    ('ff1 640 @32', 'rows', (2048, 5120, 640)),
    ('ff2 640 @32', 'rows', (2048, 640, 2560)),
    ('ff1 1280 @16', 'rows', (512, 10240, 1280)),
    ('ff2 1280 @16', 'rows', (512, 1280, 5120)),
    ('kv ctx 1280', 'rows', (154, 2560, 768)),
    ('emb proj', 'rows', (2, 20160, 1280)),
    ('tiny 64x64x64', 'rows', (64, 64, 64)),
    ('tiny 64x64x640', 'rows', (64, 64, 640)),
    ('small M2048 N640 K640', 'rows', (2048, 640, 640)),
    ('small M512 N1280 K1280', 'rows', (512, 1280, 1280)),
    ('small M8192 N320 K320', 'rows', (8192, 320, 320)),
    ('vae conv 128->128 @512', 'conv', (1, 512, 512, 128, 128, 1, False)),
    ('vae conv 256->256 @256', 'conv', (1, 256, 256, 256, 256, 1, False)),
    ('vae conv 512->512 @128', 'conv', (1, 128, 128, 512, 512, 1, False)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tiles', default='0')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--only', default='')
    ap.add_argument('--split', type=int, default=0, help='force split-K factor (0 = auto)')
    ap.add_argument('--lib', default='', help='load lib/<name> instead of libsdod.so (ablation builds, see the Makefile)')
    ap.add_argument('--u8', action='store_true', help='affine-uint8 weight codes (config 5): the kernels stream bytes and expand on the fragment read')
    ap.add_argument('--cold', action='store_true', help='sweep the caches before every timed launch (weights from HBM)')
    args = ap.parse_args()
    if args.lib:
        import ctypes
        from sdod.amd import _lib
        _lib._cache['libsdod.so'] = ctypes.CDLL(os.path.join(_lib.LIB_DIR, args.lib), mode=ctypes.RTLD_GLOBAL)
    tiles = [int(t) for t in args.tiles.split(',')]
    d = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(0)
    scratch = torch.zeros(512 << 20, dtype=torch.uint8, device=d) if args.cold else None
    print(f'{"shape":28s} {"GFLOP":>8s} ' + ' '.join(f'{"t" + str(t) + " us":>9s} {"TF/s":>7s}' for t in tiles), flush=True)
    def weights(n, k):
        if not args.u8:
            return (torch.randn(n, k, generator=g) * k ** -0.5).half().to(d), {}
        q = torch.randint(0, 256, (n, k), generator=g, dtype=torch.uint8).to(d)
        return q, dict(w_scale=torch.full((n,), 2.0 / 255 * k ** -0.5).to(d), w_off=torch.zeros(n).to(d))

    for name, kind, prm in SHAPES:
        if args.only and args.only not in name:
            continue
        if kind == 'rows':
            m, n, k = prm
            a = torch.randn(m, k, generator=g).half().to(d)
            w, wkw = weights(n, k)
            bias = torch.randn(n).to(d)
            fl = 2.0 * m * n * k
            call = lambda t, it=0: ops.gemm(a, w, bias, tile=t, time_iters=it, split_k=args.split, cold_scratch=scratch, **wkw)
        else:
            nb, h, wd, cin, cout, stride, ups = prm
            a = torch.randn(nb, h, wd, cin, generator=g).half().to(d)
            w, wkw = weights(cout, 9 * cin)
            bias = torch.randn(cout).to(d)
            ho = (h * (2 if ups else 1)) // stride
            fl = 2.0 * nb * ho * ho * cout * 9 * cin
            call = lambda t, it=0: ops.gemm(a, w, bias, conv=dict(stride=stride, upsample=ups), tile=t, time_iters=it, split_k=args.split, cold_scratch=scratch, **wkw)
        row = f'{name:28s} {fl / 1e9:8.2f} '
        for t in tiles:
            try:
                us = call(t, args.iters) * 1e3   # timed inside the library: back-to-back launches, HIP events
            except Exception:                    # a tile that declines the shape (halo-patch tiles: geometry)
                row += f'{"--":>9s} {"--":>7s} '
                continue
            row += f'{us:9.1f} {fl / us / 1e6:7.1f} '
        print(row, flush=True)


if __name__ == '__main__':
    main()
