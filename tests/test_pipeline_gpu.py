"""End-to-end parity of the sampler loops + decode on the GPU against the CPU oracle, same synthetic weights,
injected x_T, guidance 7.5.  Reduced latent (16x16) keeps the CPU side in tens of seconds; the full 64x64 loop is
covered by properties (finite, deterministic, sharding-invariant) and by bench.py's in-run parity block.

Stated tolerances (fp16 GPU vs fp32 CPU): final latent rel-L2 <= 2e-2; uint8 image within 2 LSB on
>= 99% of pixels (measured: max diff 1 LSB, 100% within 2 LSB).
Scheduler index sequences exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope='module')
def rig():
    from oracle import sd_torch as S
    from sdod.amd import engine as E, weights as Wt
    from sdod.amd.pipeline import Txt2Img
    cfg = E.sd14_config(16, 16)
    tables = {'unet': E.UNet(cfg, 2).param_table(), 'temb': E.Temb(cfg, 1).param_table(),
              'vae': E.VaeDecoder(cfg, 1).param_table(), 'text': E.TextEncoder(cfg, 1).param_table()}
    sds = {k: Wt.synthetic_state_dict(t, seed=1234 + i) for i, (k, t) in enumerate(tables.items())}
    pipe = Txt2Img(state_dicts=sds, images_per_gpu=1, latent_hw=16)
    with torch.device('meta'):
        unet, vae, clip = S.UNetModel(), S.AutoencoderKLDecode(), S.ClipTextModel()
    unet.load_state_dict({**sds['unet'], **sds['temb']}, assign=True)
    vae.load_state_dict(sds['vae'], assign=True)
    clip.load_state_dict(sds['text'], assign=True)
    return pipe, unet.eval(), vae.eval(), clip.eval()


def _ctx(pipe, clip):
    ids_u = np.full(77, 49407, np.int64); ids_u[0] = 49406
    ids_c = ids_u.copy(); ids_c[1:9] = [320, 1125, 539, 550, 18376, 6765, 320, 4558]
    ctx2 = pipe.encode_tokens(ids_u, ids_c)
    with torch.no_grad():
        ref = clip(torch.from_numpy(np.stack([ids_u, ids_c])))
    assert rel_l2(ctx2.float().cpu(), ref) <= 5e-3
    return ctx2, ref


def test_plms_20_steps_matches_oracle(rig):
    from oracle import pipeline_oracle as PO
    pipe, unet, vae, clip = rig
    ctx2, ref_ctx = _ctx(pipe, clip)
    x_T = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(42))
    tr_gpu, tr_cpu = [], []
    z = pipe.sample_plms(ctx2, x_T, steps=20, guidance=7.5, trace=tr_gpu)
    # the oracle consumes the SAME fp16-rounded conditioning the GPU used
    c16 = ctx2.float().cpu()
    z_ref = PO.plms_sample(unet, c16[0:1], c16[1:2], x_T, steps=20, scale=7.5, trace=tr_cpu)
    assert tr_gpu == tr_cpu                      # timestep / index sequence: exact
    r = rel_l2(z.cpu(), z_ref)
    print('plms final latent rel-L2', r)
    assert torch.isfinite(z).all() and r <= 2e-2, r
    img = pipe.decode(z, mode=1).cpu().numpy()
    img_ref = PO.decode_u8(vae, z_ref, mode=1)
    diff = np.abs(img.astype(np.int32) - img_ref.astype(np.int32))
    frac = float((diff <= 2).mean())
    print('uint8 image: max diff', int(diff.max()), 'mean', float(diff.mean()), 'within 2 LSB', frac)
    assert img.shape == (1, 128, 128, 3) and frac >= 0.99, frac


def test_dpm_driver_loop_matches_oracle(rig, oracle_lib):
    from oracle import pipeline_oracle as PO
    pipe, unet, vae, clip = rig
    ctx2, _ = _ctx(pipe, clip)
    x_T = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(43))
    z = pipe.sample_dpm(ctx2, x_T, steps=20, guidance=7.5)
    c16 = ctx2.float().cpu()
    z_ref = PO.dpm_sample(unet, oracle_lib, c16[0:1], c16[1:2], x_T, steps=20, guidance=7.5)
    r = rel_l2(z.cpu(), z_ref)
    print('dpm final latent rel-L2', r)
    assert torch.isfinite(z).all() and r <= 2e-2, r
    img = pipe.decode(z, mode=0).cpu().numpy()
    img_ref = PO.decode_u8(vae, z_ref, mode=0, oracle_lib=oracle_lib)
    frac = float((np.abs(img.astype(np.int32) - img_ref.astype(np.int32)) <= 2).mean())
    print('dpm uint8 within 2 LSB', frac)
    assert frac >= 0.99, frac


def test_pipeline_is_deterministic_and_sharding_invariant(rig):
    """same (seed, image index) -> same bits regardless of which rank/batch slot computes it"""
    from sdod.amd.pipeline import initial_latent, shard_images
    pipe, unet, vae, clip = rig
    ctx2, _ = _ctx(pipe, clip)
    assert shard_images(16, 3, 8) == [6, 7] and shard_images(5, 2, 4) == [4] and shard_images(5, 3, 4) == []
    x = initial_latent(42, 7, (4, 16, 16))
    assert torch.equal(x, initial_latent(42, 7, (4, 16, 16)))
    a = pipe.generate(ctx2, x, steps=4, guidance=7.5, sampler='plms')
    b = pipe.generate(ctx2, x, steps=4, guidance=7.5, sampler='plms')
    assert torch.equal(a, b)
    # the whole trajectory replayed as ONE device graph: same kernels on the same buffers -> same bits, also for new inputs
    c = pipe.generate_graphed(ctx2, x, steps=4, guidance=7.5, sampler='plms').clone()
    assert torch.equal(a, c)
    x2 = initial_latent(42, 8, (4, 16, 16))
    d = pipe.generate_graphed(ctx2, x2, steps=4, guidance=7.5, sampler='plms').clone()
    assert torch.equal(d, pipe.generate(ctx2, x2, steps=4, guidance=7.5, sampler='plms')) and not torch.equal(d, a)
    e = pipe.generate_graphed(ctx2, x, steps=3, guidance=5.0, sampler='dpm').clone()
    assert torch.equal(e, pipe.generate(ctx2, x, steps=3, guidance=5.0, sampler='dpm'))
    # throughput form: sampling graph on the current stream, decode graph on a side stream under the NEXT image's sampling --
    # three images back to back, each identical to the serial result (the latent hand-off and the output buffer are the
    # only shared state: an image is read after ITS event and before the next decode starts)
    outs = []
    for xi in (x, x2, x):
        img, ev = pipe.generate_pipelined(ctx2, xi, steps=4, guidance=7.5, sampler='plms')
        ev.synchronize()
        outs.append(img.clone())
    assert torch.equal(outs[0], a) and torch.equal(outs[1], d) and torch.equal(outs[2], a)
    # ... and without waiting in between (the stream order alone must keep image i's latent intact until its decode has it)
    evs = []
    for xi in (x, x2):
        img, ev = pipe.generate_pipelined(ctx2, xi, steps=4, guidance=7.5, sampler='plms')
        evs.append(ev)
    evs[-1].synchronize()
    assert torch.equal(img, d)


def test_config4_dpm_50_steps_two_images_per_gpu(rig, oracle_lib):
    """BASELINE config 4 on one rank's share: 50-step DPM, 2 images per GPU (UNet batch 4 = 2 x (uncond, cond)), distinct
    x_T per image index; every image must match the oracle run of that image alone (sharding changes nothing)."""
    from oracle import pipeline_oracle as PO
    from sdod.amd.pipeline import Txt2Img, initial_latent, shard_images
    pipe1, unet, vae, clip = rig
    pipe = Txt2Img(state_dicts=pipe1._sd, images_per_gpu=2, latent_hw=16, with_text_encoder=False)
    ctx2, _ = _ctx(pipe1, clip)
    mine = shard_images(16, 3, 8)                     # rank 3 of 8 owns images 6 and 7
    x_T = torch.cat([initial_latent(7, i, (4, 16, 16)) for i in mine], 0)
    z = pipe.sample_dpm(ctx2, x_T, steps=50, guidance=7.5)
    assert z.shape == (2, 4, 16, 16) and torch.isfinite(z).all()
    c16 = ctx2.float().cpu()
    z_ref = PO.dpm_sample(unet, oracle_lib, c16[0:1], c16[1:2], x_T, steps=50, guidance=7.5)   # images are independent rows
    for j in range(2):
        r = rel_l2(z[j:j + 1].cpu(), z_ref[j:j + 1])
        print(f'config 4, image {mine[j]}: dpm-50 final latent rel-L2', r)
        assert r <= 2e-2, r
    img = pipe.decode(z, mode=0)
    assert img.shape == (2, 128, 128, 3) and img.dtype == torch.uint8


# ------------------------------------------------------------------ BASELINE config 5: SD v2.1 shapes, int8 weights, PLMS
@pytest.fixture(scope='module')
def rig21():
    """SD v2.1 UNet (64-wide heads, context 1024, Linear transformer projections) with every conv / linear weight stored
    as per-tensor affine uint8 (the reference's QNN weight format); the oracle computes in fp32 on the SAME dequantised
    values.  Latent 24x24 keeps the CPU side short; the 96x96 evaluation is test_config5_unet_step_at_768px."""
    from oracle import sd_torch as S
    from sdod.amd import engine as E, weights as Wt
    cfg = E.sd21_config(24, 24)
    tables = {'unet': E.UNet(cfg, 2).param_table(), 'temb': E.Temb(cfg, 1).param_table()}
    sds = {k: Wt.quantize_state_dict(Wt.synthetic_state_dict(t, seed=2100 + i)) for i, (k, t) in enumerate(tables.items())}
    nq = sum(isinstance(v, Wt.QuantU8) for sd in sds.values() for v in sd.values())
    assert nq == 282                                  # every conv / linear weight of the UNET + TEMB graphs
    with torch.device('meta'):
        unet = S.UNetModel(context_dim=1024, head_dim=64, use_linear=True)
    deq = {k: (v.dequantize() if isinstance(v, Wt.QuantU8) else v) for sd in sds.values() for k, v in sd.items()}
    unet.load_state_dict(deq, assign=True)
    return sds, unet.eval()


def test_config5_sd21_int8_weights_plms_matches_oracle(rig21):
    from oracle import pipeline_oracle as PO
    from sdod.amd.pipeline import Txt2Img
    sds, unet = rig21
    pipe = Txt2Img(state_dicts=sds, images_per_gpu=1, latent_hw=24, model='sd21', with_vae=False, with_text_encoder=False)
    g = torch.Generator().manual_seed(77)
    ctx2 = torch.randn(2, 77, 1024, generator=g).half().cuda()           # stands in for the OpenCLIP-H tower (tests/test_engine_gpu.py)
    x_T = torch.randn(1, 4, 24, 24, generator=g)
    tr_gpu, tr_cpu = [], []
    z = pipe.sample_plms(ctx2, x_T, steps=20, guidance=7.5, trace=tr_gpu)
    c = ctx2.float().cpu()
    z_ref = PO.plms_sample(unet, c[0:1], c[1:2], x_T, steps=20, scale=7.5, trace=tr_cpu, parameterization='v')
    assert tr_gpu == tr_cpu
    r = rel_l2(z.cpu(), z_ref)
    print('config 5 (sd21 shapes, u8 weights, v-prediction PLMS) final latent rel-L2', r)
    assert torch.isfinite(z).all() and r <= 2e-2, r


def test_config5_unet_step_at_768px_uint8_weights_in_hbm(rig21):
    """the same evaluation with weight_quant = 1: conv / linear weights STAY affine uint8 in HBM and the GEMMs stream the codes
    (sdod_gemm_desc.wq); weight arena ~0.87 GB instead of 1.73 GB.  Same oracle, same tolerance."""
    from sdod.amd import engine as E
    sds, unet = rig21
    cfg = E.sd21_config(96, 96)
    cfg.weight_quant = 1
    g = E.UNet(cfg, 2)
    g.load_state_dict(sds['unet'])
    g.finalize()
    te = E.Temb(cfg, 1)
    te.load_state_dict(sds['temb'])
    te.finalize()
    gen = torch.Generator().manual_seed(78)
    x = torch.randn(2, 4, 96, 96, generator=gen)
    ctx = torch.randn(2, 77, 1024, generator=gen).half()
    te.t.copy_(torch.tensor([601.0]))
    te.execute()
    g.x.copy_(x); g.ctx.copy_(ctx); g.temb.copy_(te.out.expand(2, -1))
    g.execute(True)
    out = g.eps.float().cpu().permute(0, 3, 1, 2)
    with torch.no_grad():
        ref = unet(x, torch.tensor([601.0, 601.0]), ctx.float())
    r = rel_l2(out, ref)
    st = g.stats()
    print('config 5 UNet step @96x96, uint8 weights in HBM: rel-L2', r, st)
    assert torch.isfinite(out).all() and r <= 1e-2, r
    assert st['weight_bytes'] < 0.95e9, st          # 866 M one-byte codes + fp32 vectors (fp16 build: 1.73 GB)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.execute(True); ev0.record()
    for _ in range(5):
        g.execute(True, static_unchanged=True)
    ev1.record(); torch.cuda.synchronize()
    print('config 5 UNet step @96x96, uint8 weights: %.2f ms per replay' % (ev0.elapsed_time(ev1) / 5))


def test_config5_unet_step_weight_quant_where_it_pays(rig21):
    """weight_quant = 2 (sdod_engine.h): the same uint8 checkpoint, but only the blocks whose GEMMs have <= 128 rows keep their
    codes in HBM and stream them; every other block is dequantised once at load and built as in an fp16 graph (LayerNorm fold,
    fused skip conv, composed ff.net.2 + proj_out).  At 24x24 latents, batch 2, that is the 6x6 / 3x3 levels (72 / 18 rows)
    against the 24x24 / 12x12 levels (1152 / 288 rows): both kinds of block in one graph.  Same oracle and tolerance as the
    other two settings; the weight arena sits between theirs; the launch list is shorter than with every tensor uint8."""
    from sdod.amd import engine as E
    sds, unet = rig21
    gen = torch.Generator().manual_seed(79)
    x = torch.randn(2, 4, 24, 24, generator=gen)
    ctx = torch.randn(2, 77, 1024, generator=gen).half()
    with torch.no_grad():
        ref = unet(x, torch.tensor([601.0, 601.0]), ctx.float())
    outs, stats, launches = {}, {}, {}
    for wq in (0, 1, 2):
        cfg = E.sd21_config(24, 24)
        cfg.weight_quant = wq
        g = E.UNet(cfg, 2)
        g.load_state_dict(sds['unet'])
        g.finalize()
        te = E.Temb(cfg, 1)
        te.load_state_dict(sds['temb'])
        te.finalize()
        te.t.copy_(torch.tensor([601.0]))
        te.execute()
        g.x.copy_(x); g.ctx.copy_(ctx); g.temb.copy_(te.out.expand(2, -1))
        g.execute(True)
        outs[wq] = g.eps.float().cpu().permute(0, 3, 1, 2)
        stats[wq], launches[wq] = g.stats(), len(g.op_table())
        r = rel_l2(outs[wq], ref)
        print(f'config 5 UNet step @24x24, weight_quant = {wq}: rel-L2 {r:.3e}, {stats[wq]["weight_bytes"] / 1e9:.3f} GB of weights, {launches[wq]} launches')
        assert torch.isfinite(outs[wq]).all() and r <= 1e-2, (wq, r)
    assert stats[1]['weight_bytes'] < stats[2]['weight_bytes'] < stats[0]['weight_bytes']
    assert launches[0] <= launches[2] < launches[1]


def test_config5_unet_step_at_768px(rig21):
    """one batch-2 UNet evaluation at the full 96x96 latent of SD v2.1-768 (L = 9216 tokens at 64-wide heads)"""
    from sdod.amd import engine as E
    sds, unet = rig21
    cfg = E.sd21_config(96, 96)
    g = E.UNet(cfg, 2)
    g.load_state_dict(sds['unet'])
    g.finalize()
    te = E.Temb(cfg, 1)
    te.load_state_dict(sds['temb'])
    te.finalize()
    gen = torch.Generator().manual_seed(78)
    x = torch.randn(2, 4, 96, 96, generator=gen)
    ctx = torch.randn(2, 77, 1024, generator=gen).half()
    te.t.copy_(torch.tensor([601.0]))
    te.execute()
    g.x.copy_(x); g.ctx.copy_(ctx); g.temb.copy_(te.out.expand(2, -1))
    g.execute(True)
    out = g.eps.float().cpu().permute(0, 3, 1, 2)
    with torch.no_grad():
        ref = unet(x, torch.tensor([601.0, 601.0]), ctx.float())
    r = rel_l2(out, ref)
    print('config 5 UNet step @96x96 rel-L2', r, g.stats())
    assert torch.isfinite(out).all() and r <= 1e-2, r
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.execute(True); ev0.record()
    for _ in range(5):
        g.execute(True, static_unchanged=True)
    ev1.record(); torch.cuda.synchronize()
    print('config 5 UNet step @96x96, fp16 image of the uint8 weights: %.2f ms per replay' % (ev0.elapsed_time(ev1) / 5))


def test_uint8_encoding_with_a_positive_offset_is_rejected_at_load():
    """the kernels fold the zero point into the code expansion (q + offset must be an exact fp16 integer): a graph that keeps
    its weights uint8 refuses an encoding outside [-1024, 0] instead of computing something else (engine.hip: set_param)"""
    from sdod.amd import engine as E, weights as Wt
    cfg = E.sd21_config(24, 24)
    cfg.weight_quant = 1
    g = E.Temb(cfg, 1)
    sd = Wt.quantize_state_dict(Wt.synthetic_state_dict(g.param_table(), seed=5))
    g.load_state_dict(sd)                         # the QNN-style encoding loads
    name = next(k for k, v in sd.items() if isinstance(v, Wt.QuantU8))
    bad = dict(sd)
    bad[name] = Wt.QuantU8(sd[name].q, sd[name].scale, 7)
    g2 = E.Temb(cfg, 1)
    with pytest.raises(Exception, match='offset must be in'):
        g2.load_state_dict(bad)
