#!/usr/bin/env python3
"""GroupNorm(+SiLU) on the UNet / VAE shapes, timed the way a launch list replays them: `reps` launches captured into
one device graph, replayed, HIP-event time / reps (= kernel + launch boundary, no Python in between).
Prints us per launch and algorithmic GB/s (2 * N*HW*C * 2 bytes).  usage: python tools/gn_bench.py [--reps 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'stable-diffusion-on-device_amd'))
import torch  # noqa: E402
from sdod.amd import _lib, ops  # noqa: E402

# (n, hw, c, count per UNet evaluation)
SHAPES = [(2, 4096, 320, 13), (2, 4096, 640, 2), (2, 4096, 960, 1), (2, 1024, 640, 11), (2, 1024, 320, 1), (2, 1024, 960, 1),
          (2, 1024, 1280, 1), (2, 1024, 1920, 1), (2, 256, 1280, 11), (2, 256, 2560, 2), (2, 256, 1920, 1), (2, 256, 640, 1),
          (2, 64, 1280, 12), (2, 64, 2560, 3),
          (4, 4096, 320, 0), (1, 4096, 320, 0), (1, 4096, 512, 0), (1, 16384, 512, 0), (1, 65536, 256, 0), (1, 262144, 128, 0)]


def graph_time(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    args = ap.parse_args()
    lib = _lib.hip()
    d = torch.device('cuda:0')
    tot = 0.0
    print(f'{"shape":26s} {"launches":>8s} {"us":>8s} {"GB/s":>8s} {"x count":>8s}')
    for n, hw, c, cnt in SHAPES:
        x = (torch.randn(n, hw, c) * 2 + 1).half().to(d)
        w = torch.randn(c).to(d); b = torch.randn(c).to(d)
        y = torch.empty_like(x)
        us = graph_time(lambda: ops.group_norm_nhwc(x, 32, w, b, 1e-5, True, out=y), args.reps)
        by = 2.0 * n * hw * c * 2
        tot += us * cnt
        print(f'n{n} hw{hw:6d} c{c:5d}         {lib.sdod_group_norm_launches(hw, c, 32, 0):8d} {us:8.2f} {by / us / 1e3:8.1f} {us * cnt:8.1f}', flush=True)
    print(f'sum over one UNet evaluation: {tot / 1e3:.3f} ms')
    # GroupNorm with the producing conv's split-K reduce folded in (sdod_group_norm_reduce_nhwc) next to the two launches
    # it replaces (splitk_reduce + GroupNorm); (n, side, cin, cout, splits, count per UNet evaluation)
    print(f'\n{"fused reduce + GroupNorm":34s} {"fused us":>9s} {"reduce us":>10s} {"gn us":>8s}')
    for n, side, cin, cout, split, cnt in [(2, 32, 640, 640, 3, 10), (2, 16, 1280, 1280, 6, 10), (2, 8, 1280, 1280, 6, 11), (2, 32, 320, 320, 3, 1)]:
        hw = side * side
        x = torch.randn(n, side, side, cin).half().to(d)
        wt = (torch.randn(cout, 9 * cin) * (9 * cin) ** -0.5).half().to(d)
        bias = torch.randn(cout).to(d)
        temb = torch.randn(n, cout).half().to(d)
        gw = torch.randn(cout).to(d); gb = torch.randn(cout).to(d)
        kw = dict(row_bias=temb, rows_per_img=hw, conv=dict(stride=1), split_k=split, tile=28)
        out, desc = ops.gemm(x, wt, bias, phase=1, return_desc=True, **kw)
        fused = graph_time(lambda: ops.group_norm_reduce(desc, n, hw, 32, gw, gb, 1e-5, True), args.reps)
        desc2 = type(desc).from_buffer_copy(desc); desc2.phase = 2
        red = graph_time(lambda: _lib.check(lib.sdod_gemm_f16(__import__('ctypes').byref(desc2), ops._stream())), args.reps)
        y = torch.empty(n, hw, cout, dtype=torch.float16, device=d)
        gn = graph_time(lambda: ops.group_norm_nhwc(out.reshape(n, hw, cout), 32, gw, gb, 1e-5, True, out=y), args.reps)
        print(f'n{n} hw{hw:5d} c{cout:5d} x{split}              {fused:9.2f} {red:10.2f} {gn:8.2f}   (x{cnt})', flush=True)


if __name__ == '__main__':
    main()
