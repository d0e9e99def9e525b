#!/usr/bin/env python3
"""Where does a workgroup of the A-panel GEMM spend its time?  (developer tool, GPU box only; needs lib/libsdod_stamp.so:
make -C stable-diffusion-on-device_amd $(pwd)/stable-diffusion-on-device_amd/lib/libsdod_stamp.so)
s_memrealtime stamps (10 ns) of consumer wave 0 / loader wave 4 of every workgroup, medians in microseconds:
  panel = entry -> first barrier passed (row panel + first W slab in LDS)     loop0 = K loop of the first tile
  epi0  = epilogue of the first tile                                          tile1 = end of epilogue 0 -> end of the K loop of tile 1
  rest  = ... -> end of the last K loop                                       epiN  = last epilogue
  ldr   = entry -> the loader has nothing left in flight                      wg    = entry -> end of the last epilogue"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stable-diffusion-on-device_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from sdod.amd import _lib  # noqa: E402

_lib._cache['libsdod.so'] = ctypes.CDLL(os.path.join(_lib.LIB_DIR, 'libsdod_stamp.so'), mode=ctypes.RTLD_GLOBAL)
from sdod.amd import ops  # noqa: E402

CASES = [('ff1 @64', 8192, 320, 2560, True, 53), ('qkv @64', 8192, 320, 960, False, 53), ('ff1 @32', 2048, 640, 5120, True, 54),
         ('ff1 @16', 512, 1280, 10240, True, 55), ('ff1 @64 no geglu', 8192, 320, 2560, False, 53)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    lib = _lib.hip()
    lib.sdod_gemm_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    d = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    print(f'{"case":18s} {"WGs":>4s} {"event":>7s} {"span":>7s} {"panel":>6s} {"loop0":>6s} {"epi0":>6s} {"tile1":>6s} {"rest":>6s} {"epiN":>6s} {"ldr":>6s} {"wg":>6s} {"wg max":>6s}')
    for name, m, c, n, geglu, tile in CASES:
        if a.only and a.only not in name:
            continue
        x = (torch.randn(m, c, generator=g) + 0.5).half().to(d)
        w = (torch.randn(n, c, generator=g) * c ** -0.5).half().to(d)
        w, sv, tv = ops.ln_fold(w, (1 + 0.1 * torch.randn(c, generator=g)).to(d), (0.1 * torch.randn(c, generator=g)).to(d), torch.randn(n, generator=g).to(d))
        kw = dict(ln_s=sv, geglu=geglu, tile=tile)
        for _ in range(3):
            ops.gemm(x, w, tv, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        ops.gemm(x, w, tv, **kw)
        e1.record()
        torch.cuda.synchronize()
        bm = {53: 128, 54: 64, 55: 32}[tile]
        panels = -(-m // bm); tiles_n = -(-n // 128)
        best, tpg = -1.0, 1      # gemm.hip: panel_grid
        for t in range(tiles_n, 0, -1):
            wgs = panels * -(-tiles_n // t)
            score = wgs / (-(-wgs // 256) * 256) * t / (t + 1.5)
            if score > best * 1.0001:
                best, tpg = score, t
        nwg = panels * -(-tiles_n // tpg)
        buf = np.zeros((nwg, 8), np.uint64)
        assert lib.sdod_gemm_stamps(buf.ctypes.data, nwg) == 0
        s = buf.astype(np.float64) * 0.01
        t0 = s[:, 0].min()
        med = lambda v: float(np.median(v))
        print(f'{name:18s} {nwg:4d} {e0.elapsed_time(e1) * 1e3:7.1f} {s[:, 6].max() - t0:7.1f} {med(s[:, 1] - s[:, 0]):6.2f} {med(s[:, 2] - s[:, 1]):6.2f} '
              f'{med(s[:, 3] - s[:, 2]):6.2f} {med(s[:, 4] - s[:, 3]):6.2f} {med(s[:, 5] - s[:, 4]):6.2f} {med(s[:, 6] - s[:, 5]):6.2f} '
              f'{med(s[:, 7] - s[:, 0]):6.2f} {med(s[:, 6] - s[:, 0]):6.2f} {float((s[:, 6] - s[:, 0]).max()):6.2f}', flush=True)


if __name__ == '__main__':
    main()
