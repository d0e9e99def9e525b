#!/usr/bin/env python3
"""Where does a GEMM launch spend its time?  (developer tool, GPU box only)

Runs the LDS-DMA GEMM of lib/libsdod_stamp.so (make $(pwd)/lib/libsdod_stamp.so in stable-diffusion-on-device_amd:
gemm.hip compiled with -DSDOD_GEMM_STAMP) on the UNet's layer shapes and prints, per shape and tile, the wall-clock
phases of the workgroups from their s_memrealtime stamps (10 ns ticks):
  skew   = last workgroup entry - first workgroup entry (dispatch of the grid)
  pro    = entry -> prologue issued (index math, first STAGES-1 slabs + epilogue vectors requested)
  loop   = main loop incl. the final drain
  stage  = accumulators -> LDS tile
  store  = LDS tile -> global (residual read, stores retired)
  span   = first entry -> last retire (the kernel as the chip sees it); `event` = HIP-event time of the launch
medians over workgroups, microseconds.  usage: python tools/gemm_phases.py [--only substr] [--tiles 13,8]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stable-diffusion-on-device_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from sdod.amd import _lib  # noqa: E402

_STAMP_LIB = next((a.split('=', 1)[1] for a in sys.argv if a.startswith('--lib=')), 'libsdod_stamp.so')
_lib._cache['libsdod.so'] = ctypes.CDLL(os.path.join(_lib.LIB_DIR, _STAMP_LIB), mode=ctypes.RTLD_GLOBAL)
from sdod.amd import ops  # noqa: E402

# (name, kind, params, extras)  rows: (M, N, K); conv: (n, h, w, cin, cout)
SHAPES = [
    ('conv 320->320 @64', 'conv', (2, 64, 64, 320, 320)),
    ('conv 640->640 @32', 'conv', (2, 32, 32, 640, 640)),
    ('conv 1280->1280 @16', 'conv', (2, 16, 16, 1280, 1280)),
    ('conv 1280->1280 @8', 'conv', (2, 8, 8, 1280, 1280)),
    ('proj 320 @64 (M8192 N320 K320)', 'rows', (8192, 320, 320)),
    ('proj 640 @32 (M2048 N640 K640)', 'rows', (2048, 640, 640)),
    ('proj 1280 @16 (M512 N1280 K1280)', 'rows', (512, 1280, 1280)),
    ('qkv 320 @64 (M8192 N960 K320)', 'rows', (8192, 960, 320)),
    ('ff1 320 @64 (M8192 N2560 K320)', 'rows', (8192, 2560, 320)),
    ('ff2 320 @64 (M8192 N320 K1280)', 'rows', (8192, 320, 1280)),
    ('ff1 1280 @16 (M512 N10240 K1280)', 'rows', (512, 10240, 1280)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='')
    ap.add_argument('--lib', default='libsdod_stamp.so', help='developer library with the stamps compiled in')
    ap.add_argument('--tiles', default='13,9,8,21')
    ap.add_argument('--split', type=int, default=1)
    ap.add_argument('--residual', action='store_true')
    ap.add_argument('--geglu', action='store_true', help='rows shapes: fused GEGLU epilogue (N = 2 x hidden, [value | gate] 16-column blocks)')
    args = ap.parse_args()
    lib = _lib.hip()
    lib.sdod_gemm_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    d = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(0)
    tiles = [int(t) for t in args.tiles.split(',')]
    hdr = f'{"shape":34s} {"tile":>4s} {"WGs":>5s} {"event":>7s} {"span":>7s} {"skew":>6s} {"pro":>6s} {"loop":>7s} {"stage":>6s} {"store":>6s} {"wg med":>7s} {"wg max":>7s}'
    print(hdr, flush=True)
    for name, kind, prm in SHAPES:
        if args.only and args.only not in name:
            continue
        if kind == 'rows':
            m, n, k = prm
            a = torch.randn(m, k, generator=g).half().to(d)
            w = (torch.randn(n, k, generator=g) * k ** -0.5).half().to(d)
            kw = {}
            res_shape = (m, n)
        else:
            nb, h, wd, cin, cout = prm
            a = torch.randn(nb, h, wd, cin, generator=g).half().to(d)
            w = (torch.randn(cout, 9 * cin, generator=g) * (9 * cin) ** -0.5).half().to(d)
            kw = dict(conv=dict(stride=1))
            n = cout
            res_shape = (nb, h, wd, cout)
        bias = torch.randn(n).to(d)
        if args.residual:
            kw['residual'] = torch.randn(res_shape, generator=g).half().to(d)
        if args.geglu and kind == 'rows':
            kw['geglu'] = True
        for t in tiles:
            try:
                for _ in range(3):
                    ops.gemm(a, w, bias, tile=t, split_k=args.split, phase=1 if args.split > 1 else 0, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                _, desc = ops.gemm(a, w, bias, tile=t, split_k=args.split, phase=1 if args.split > 1 else 0, return_desc=True, **kw)
                e1.record()
                torch.cuda.synchronize()
            except Exception as ex:   # a tile that does not take the shape
                print(f'{name:34s} {t:4d}  -- {str(ex)[:60]}')
                continue
            tt, sp = ctypes.c_int(), ctypes.c_int()
            lib.sdod_gemm_plan(ctypes.byref(desc), ctypes.byref(tt), ctypes.byref(sp))
            # grid size from the plan: the stamp table is indexed by workgroup
            from math import ceil
            bm_, bn_, dma_ = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            lib.sdod_gemm_tile_shape(tt.value, ctypes.byref(bm_), ctypes.byref(bn_), ctypes.byref(dma_))
            bm, bn = (bm_.value, bn_.value) if dma_.value else (0, 0)
            if not bm:
                print(f'{name:34s} {t:4d}  -- tile {tt.value} is not an LDS-DMA tile')
                continue
            nwg = ceil(desc.M / bm) * ceil(desc.N / bn) * sp.value
            nwg = min(nwg, 8192)
            buf = np.zeros((nwg, 8), np.uint64)
            assert lib.sdod_gemm_stamps(buf.ctypes.data, nwg) == 0
            s = buf.astype(np.float64) * 0.01      # us
            t0 = s[:, 0].min()
            med = lambda x: float(np.median(x))
            print(f'{name:34s} {tt.value:4d} {nwg:5d} {e0.elapsed_time(e1) * 1e3:7.1f} {s[:, 4].max() - t0:7.1f} {s[:, 0].max() - t0:6.1f} '
                  f'{med(s[:, 1] - s[:, 0]):6.2f} {med(s[:, 2] - s[:, 1]):7.2f} {med(s[:, 3] - s[:, 2]):6.2f} {med(s[:, 4] - s[:, 3]):6.2f} '
                  f'{med(s[:, 4] - s[:, 0]):7.2f} {float((s[:, 4] - s[:, 0]).max()):7.2f}', flush=True)


if __name__ == '__main__':
    main()
