#!/usr/bin/env python3
"""Regenerate the shipped GEMM tile table (stable-diffusion-on-device_amd/tune/gfx950.tune) on an MI355X.

Builds every graph of the configurations the benches and the GPU tests use with the shipped table ignored
(SDOD_TUNE_DEFAULT=0) and SDOD_TUNE_CACHE pointing at a fresh file, so every distinct GEMM shape is timed cold-cache once
(engine.hip: emit_gemm) and its pick appended.  Copy the result over tune/gfx950.tune and commit it: from then on
every process builds the same launch lists without timing anything.

usage (GPU box):  python tools/make_tune_cache.py gpurun_out/gfx950.tune [--quick | --only-quant | --only-rows3 | --only-ln | --only-new]
NEVER run under a profiler (timing noise would be baked into the picks)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'stable-diffusion-on-device_amd'))

out = os.path.abspath(sys.argv[1])
quick = '--quick' in sys.argv
only_quant = '--only-quant' in sys.argv   # keep the shipped picks of every fp16 shape, re-time the uint8-weight shapes only
only_rows3 = '--only-rows3' in sys.argv   # re-time only the 3x3 stride-1 convolutions on images whose rows are multiples of 3 (config 5)
only_ln = '--only-ln' in sys.argv         # keep every shipped pick but those of the LayerNorm-folded Linears (key flag bit 30): re-time those
only_n160 = '--only-n160' in sys.argv     # re-time the fp16 rows-mode shapes whose N is a multiple of 160 (candidates for the 160-wide tiles 56..58)
only_geglu = '--only-geglu' in sys.argv   # re-time the LayerNorm-folded GEGLU projections (key flags: bit 1 and bit 30)
only_softmax = '--only-softmax' in sys.argv # re-time the score GEMMs of the folded cross-attention (key flag bit 28)
only_tail = '--only-tail' in sys.argv     # re-time the 3x3 convolutions with a fused 1x1 skip (key flags: tc0 in bits 2..13)
only_new = '--only-new' in sys.argv       # keep EVERY shipped pick, time only the shapes the table does not have yet (a new fusion's GEMMs)
if os.path.exists(out):
    os.remove(out)
os.makedirs(os.path.dirname(out), exist_ok=True)
if only_quant or only_rows3 or only_ln or only_new or only_n160 or only_geglu or only_softmax or only_tail:
    shipped = os.path.join(ROOT, 'stable-diffusion-on-device_amd', 'tune', 'gfx950.tune')
    with open(shipped) as f, open(out, 'w') as g:
        for line in f:
            v = line.split()
            if len(v) not in (13, 15):
                continue
            # key fields (engine.hip: key_of): 0 a_mode, 6 stride, 8 ksize, 9 h_in, 10 flags (bit 29 = uint8 weights, bit 30 = LayerNorm fold)
            if only_new:
                drop = False
            elif only_tail:
                drop = v[0] == '1' and ((int(v[10]) >> 2) & 4095) != 0 and not (int(v[10]) & (1 << 29))
            elif only_softmax:
                drop = int(v[10]) & (1 << 28)
            elif only_geglu:
                drop = (int(v[10]) & 2) and (int(v[10]) & (1 << 30))
            elif only_n160:
                drop = v[0] == '0' and int(v[2]) % 160 == 0 and not (int(v[10]) & ((1 << 29) | (1 << 28) | 2))   # (not uint8, not the score GEMMs, not GEGLU: those have their own switches)
            elif only_ln:
                drop = int(v[10]) & (1 << 30)
            else:
                drop = (int(v[10]) & (1 << 29)) if only_quant else (v[0] == '1' and v[6] == '1' and v[8] == '3' and int(v[9]) % 3 == 0)
            if not drop:
                g.write(line)
os.environ['SDOD_TUNE_CACHE'] = out
os.environ['SDOD_TUNE_DEFAULT'] = '0'

import torch  # noqa: E402
from sdod.amd import engine as E, weights as Wt  # noqa: E402

T0 = time.time()


def build(cls, cfg, batch, seed, label, quant=False):
    g = cls(cfg, batch)
    sd = Wt.synthetic_state_dict(g.param_table(), seed=seed)
    g.load_state_dict(Wt.quantize_state_dict(sd) if quant else sd)
    g.finalize()
    torch.cuda.synchronize()
    n = sum(1 for _ in open(out)) if os.path.exists(out) else 0
    print(f'[{time.time() - T0:6.1f}s] {label}: {g.stats()["launches"]} launches; table now {n} shapes', flush=True)
    del g


if only_quant:
    for hw in (96, 24):
        cq = E.sd21_config(hw, hw)
        cq.weight_quant = 1
        build(E.UNet, cq, 2, 2100, f'sd21 unet {hw}x{hw} b2, uint8 weights', quant=True)
        build(E.Temb, cq, 1, 2101, 'sd21 temb b1, uint8 weights', quant=True)
        build(E.Temb, cq, 20, 2101, 'sd21 temb b20, uint8 weights', quant=True)
    print(f'done: {out}')
    sys.exit(0)

if only_ln:
    c64 = E.sd14_config(64, 64)
    for b in (2, 4, 1):
        build(E.UNet, c64, b, 1234, f'sd14 unet 64x64 b{b}')
    build(E.UNet, E.sd21_config(96, 96), 2, 2100, 'sd21 unet 96x96 b2')
    c16 = E.sd14_config(16, 16)
    for b in (2, 4, 1):
        build(E.UNet, c16, b, 1234, f'sd14 unet 16x16 b{b}')
    build(E.UNet, E.sd21_config(24, 24), 2, 2100, 'sd21 unet 24x24 b2')
    print(f'done: {out}')
    sys.exit(0)

if only_rows3:
    c96 = E.sd21_config(96, 96)
    build(E.UNet, c96, 2, 2100, 'sd21 unet 96x96 b2')
    for hw in (96, 24):
        cq = E.sd21_config(hw, hw)
        cq.weight_quant = 1
        build(E.UNet, cq, 2, 2100, f'sd21 unet {hw}x{hw} b2, uint8 weights', quant=True)
    build(E.VaeDecoder, c96, 1, 1236, 'vae 96x96')
    build(E.UNet, E.sd21_config(24, 24), 2, 2100, 'sd21 unet 24x24 b2')
    print(f'done: {out}')
    sys.exit(0)

# headline (configs 2/3), config 4's per-rank share (batch 4), latency mode (batch 1)
c64 = E.sd14_config(64, 64)
for b in (2, 4, 1):
    build(E.UNet, c64, b, 1234, f'sd14 unet 64x64 b{b}')
build(E.VaeDecoder, c64, 1, 1236, 'sd14 vae 64x64')
build(E.TextEncoder, c64, 2, 1237, 'clip-L b2')
build(E.TextEncoder, c64, 1, 1237, 'clip-L b1')
for b in (1, 2, 4):
    build(E.Temb, c64, b, 1235, f'temb b{b}')
build(E.Temb, c64, 20, 1235, 'temb b20')
build(E.Temb, c64, 50, 1235, 'temb b50')
if not quick:
    # config 5 (SD v2.1-768) and the reduced sizes the GPU tests run
    c96 = E.sd21_config(96, 96)
    build(E.UNet, c96, 2, 2100, 'sd21 unet 96x96 b2')
    for hw in (96, 24):      # config 5 with the weights kept uint8 in HBM (sdod_model_config.weight_quant)
        cq = E.sd21_config(hw, hw)
        cq.weight_quant = 1
        build(E.UNet, cq, 2, 2100, f'sd21 unet {hw}x{hw} b2, uint8 weights', quant=True)
        build(E.Temb, cq, 1, 2101, f'sd21 temb b1, uint8 weights', quant=True)
        build(E.Temb, cq, 20, 2101, f'sd21 temb b20, uint8 weights', quant=True)
    build(E.VaeDecoder, c96, 1, 1236, 'vae 96x96')
    build(E.TextEncoder, c96, 2, 2102, 'openclip-H b2')
    build(E.Temb, c96, 1, 2101, 'sd21 temb b1')
    build(E.Temb, c96, 20, 2101, 'sd21 temb b20')
    c16 = E.sd14_config(16, 16)
    for b in (2, 4, 1):
        build(E.UNet, c16, b, 1234, f'sd14 unet 16x16 b{b}')
    build(E.VaeDecoder, c16, 1, 1236, 'vae 16x16')
    c24 = E.sd21_config(24, 24)
    build(E.UNet, c24, 2, 2100, 'sd21 unet 24x24 b2')
print(f'done: {out}')
