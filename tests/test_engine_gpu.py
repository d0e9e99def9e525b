"""GPU parity of the four graphs (engine C ABI, include/sdod_engine.h) against the PyTorch-CPU fp32 oracle
(oracle/sd_torch.py) on the same seeded synthetic weights and inputs.

Stated fp16 tolerances: temb / CLIP rel-L2 <= 5e-3; one UNet evaluation rel-L2 <= 1e-2;
VAE decode rel-L2 <= 1e-2.  Oracle parity is "unpinned" by reference tests at this boundary (no ldm, no
checkpoint offline); its structural known answers are checked in tests/test_oracle_models.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope='module')
def unet_oracle():
    from oracle import sd_torch as S
    return S.build(S.UNetModel, seed=1234)


@pytest.fixture(scope='module')
def vae_oracle():
    from oracle import sd_torch as S
    return S.build(S.AutoencoderKLDecode, seed=1235)


@pytest.fixture(scope='module')
def clip_oracle():
    from oracle import sd_torch as S
    return S.build(S.ClipTextModel, seed=1236)


def test_param_tables_match_oracle_state_dicts(unet_oracle, vae_oracle, clip_oracle):
    """every engine parameter exists in the ldm/HF-named oracle state dict with the same shape, and vice versa"""
    from sdod.amd import engine as E
    cfg = E.sd14_config(16, 16)
    for cls, orc, extra in ((E.UNet, unet_oracle, E.Temb), (E.VaeDecoder, vae_oracle, None), (E.TextEncoder, clip_oracle, None)):
        g = cls(cfg, 1)
        table = dict(g.param_table())
        if extra is not None:
            table.update(dict(extra(cfg, 1).param_table()))
        sd = {k: tuple(v.shape) for k, v in orc.state_dict().items()}
        assert set(table) == set(sd), (sorted(set(table) ^ set(sd))[:6])
        for k, shp in table.items():
            assert tuple(shp) == sd[k], (k, shp, sd[k])
        del g


def test_temb_graph(unet_oracle):
    from oracle import sd_torch as S
    from sdod.amd import engine as E
    cfg = E.sd14_config()
    g = E.Temb(cfg, 4)
    g.load_state_dict(unet_oracle.state_dict())
    g.finalize()
    t = torch.tensor([999.0, 949.05, 49.949936, 1.0])
    g.t.copy_(t)
    g.execute()
    torch.cuda.synchronize()
    with torch.no_grad():
        emb = unet_oracle.time_embed(S.timestep_embedding(t, 320))
        # the graph's output = emb pushed through every ResBlock's emb_layers (SiLU -> Linear), concatenated in
        # the order the UNet visits them
        blocks = [m[0] for m in list(unet_oracle.input_blocks) + [unet_oracle.middle_block] + list(unet_oracle.output_blocks)
                  if isinstance(m[0], S.ResBlock)]
        blocks.insert(blocks.index(unet_oracle.middle_block[0]) + 1, unet_oracle.middle_block[2])
        ref = torch.cat([b.emb_layers(emb) for b in blocks], dim=1)
    assert g.out.shape == ref.shape, (g.out.shape, ref.shape)
    r = rel_l2(g.out.float().cpu(), ref)
    assert r <= 5e-3, r


def test_text_encoder_graph(clip_oracle):
    from sdod.amd import engine as E
    cfg = E.sd14_config()
    g = E.TextEncoder(cfg, 2)
    g.load_state_dict(clip_oracle.state_dict())
    g.finalize()
    ids = torch.randint(0, 49408, (2, 77), generator=torch.Generator().manual_seed(5))
    ids[0, 10:] = 49407
    g.ids.copy_(ids.int())
    g.execute()
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = clip_oracle(ids)
    r = rel_l2(g.out.float().cpu(), ref)
    print(f'clip rel-L2 {r:.3e}')
    assert torch.isfinite(g.out).all()
    assert r <= 5e-3, r
    # replay through a captured hipGraph gives identical bits
    first = g.out.clone()
    g.execute(use_hip_graph=True); g.execute(use_hip_graph=True)
    torch.cuda.synchronize()
    assert torch.equal(first, g.out)


def _unet_case(unet_oracle, hw, batch, seed):
    from oracle import sd_torch as S
    from sdod.amd import engine as E
    cfg = E.sd14_config(hw, hw)
    g = E.UNet(cfg, batch)
    g.load_state_dict(unet_oracle.state_dict())
    g.finalize()
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, 4, hw, hw, generator=gen)
    t = torch.tensor([999.0] * batch) if seed % 2 == 0 else torch.tensor([251.0] * batch)
    ctx = torch.randn(batch, 77, 768, generator=gen)
    with torch.no_grad():
        ref = unet_oracle(x, t, ctx.half().float())
    tg = E.Temb(cfg, batch)
    tg.load_state_dict(unet_oracle.state_dict())
    tg.finalize()
    tg.t.copy_(t); tg.execute()
    g.x.copy_(x); g.temb.copy_(tg.out); g.ctx.copy_(ctx.half())
    g.execute()
    torch.cuda.synchronize()
    out = g.eps.float().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(out).all()
    r = rel_l2(out, ref)
    print(f'unet {hw}x{hw} b{batch} eager rel-L2 {r:.3e}')
    eager = g.eps.clone()
    g.execute(use_hip_graph=True); g.execute(use_hip_graph=True, static_unchanged=True)
    torch.cuda.synchronize()
    assert torch.equal(eager, g.eps), 'hipGraph replay differs from eager execution'
    return r, g.stats()


def test_unet_graph_small_latent(unet_oracle):
    r, st = _unet_case(unet_oracle, 16, 2, 10)
    print('unet 16x16 rel-L2', r, st)
    assert r <= 1e-2, r


def test_unet_graph_config2_full_size(unet_oracle):
    """BASELINE.json configs[1]: one UNet denoise step, 4x64x64 latent, batch 2 (cond+uncond)"""
    r, st = _unet_case(unet_oracle, 64, 2, 11)
    print('unet 64x64 rel-L2', r, st)
    assert r <= 1e-2, r
    assert abs(st['flops'] / 1.607e12 - 1) < 0.02, st   # 1.607 TFLOP per batch-2 evaluation


def test_unet_graph_config4_batch4_full_size(unet_oracle):
    """BASELINE.json configs[3]: per-rank share of the 16-image DPM job -- 2 images per GPU = a batch-4 UNet evaluation at
    64x64 (the M = 16384 / 4096 / 1024 / 256 GEMM shapes of tune/gfx950.tune), against the fp32 oracle"""
    r, st = _unet_case(unet_oracle, 64, 4, 13)
    print('unet 64x64 batch 4 rel-L2', r, st)
    assert r <= 1e-2, r
    assert abs(st['flops'] / (2 * 1.607e12) - 1) < 0.02, st


def test_vae_decoder_graph(vae_oracle):
    from sdod.amd import engine as E
    cfg = E.sd14_config(16, 16)
    g = E.VaeDecoder(cfg, 1)
    g.load_state_dict(vae_oracle.state_dict())
    g.finalize()
    z = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(12)) * 0.18215 * 4
    with torch.no_grad():
        ref = vae_oracle(z)
    g.z.copy_(z)
    g.execute()
    torch.cuda.synchronize()
    out = g.img.float().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(out).all()
    r = rel_l2(out, ref)
    print('vae 16x16 rel-L2', r, g.stats())
    assert r <= 1e-2, r


def test_openclip_text_tower_graph_sd21():
    """config 5's conditioning: OpenCLIP ViT-H/14 text tower (open_clip names, fused in_proj, erf GELU, penultimate block +
    ln_final) against the fp32 oracle on the same synthetic weights; prompts padded with id 0 as open_clip's tokenizer does"""
    from oracle import sd_torch as S
    from sdod.amd import engine as E, weights as Wt
    cfg = E.sd21_config()
    g = E.TextEncoder(cfg, 2)
    sd = Wt.synthetic_state_dict(g.param_table(), seed=2150)
    g.load_state_dict(sd)
    g.finalize()
    with torch.device('meta'):
        ref = S.OpenClipTextModel(layers=23, run_layers=23)
    ref.load_state_dict(sd, assign=True)
    ids = np.zeros((2, 77), np.int64)
    ids[:, 0] = 49406
    ids[0, 1] = 49407
    ids[1, 1:9] = [320, 1125, 539, 550, 18376, 6765, 320, 4558]; ids[1, 9] = 49407
    g.ids.copy_(torch.from_numpy(ids.astype(np.int32)))
    g.execute(True); g.execute(True)
    with torch.no_grad():
        want = ref.eval()(torch.from_numpy(ids))
    out = g.out.float().cpu()
    r = rel_l2(out, want)
    print('OpenCLIP-H text tower rel-L2', r, g.stats())
    assert out.shape == (2, 77, 1024) and torch.isfinite(out).all() and r <= 5e-3, r
