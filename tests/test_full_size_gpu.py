"""Full-size parity on the GPU (BASELINE.json configs[2] and the VAE halves of configs[2]/[4]): the sizes bench.py
and the C API actually run, against the fp32 CPU oracle on the same synthetic weights and injected x_T.

  * config 3 end to end: 64x64 latent, 20-step PLMS, guidance 7.5, whole-trajectory graph AND the per-step path ->
    final latent rel-L2 <= 2e-2, 512x512 uint8 image within 2 LSB on >= 99 % of pixels (measured round 1: 1.3e-3,
    max 1 LSB), scheduler trace exact;
  * VAE decoder 64x64 -> 512x512 on its own (float output, rel-L2 <= 1e-2): the 256^2 / 512^2 GroupNorm maps
    (65,536 and 262,144 pixels) only exist at this size;
  * config 5's VAE decode 96x96 -> 768x768.
The oracle side costs ~2-3 minutes of host CPU in total (16 threads)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

IDS_U = np.full(77, 49407, np.int64); IDS_U[0] = 49406
IDS_C = IDS_U.copy(); IDS_C[1:9] = [320, 1125, 539, 550, 18376, 6765, 320, 4558]


def rel_l2(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope='module')
def full_rig():
    from oracle import sd_torch as S
    from sdod.amd import engine as E, weights as Wt
    from sdod.amd.pipeline import Txt2Img
    cfg = E.sd14_config(64, 64)
    tables = {'unet': E.UNet(cfg, 2).param_table(), 'temb': E.Temb(cfg, 1).param_table(),
              'vae': E.VaeDecoder(cfg, 1).param_table(), 'text': E.TextEncoder(cfg, 1).param_table()}
    sds = {k: Wt.synthetic_state_dict(t, seed=1234 + i) for i, (k, t) in enumerate(tables.items())}
    pipe = Txt2Img(state_dicts=sds, images_per_gpu=1, latent_hw=64)
    with torch.device('meta'):
        unet, vae = S.UNetModel(), S.AutoencoderKLDecode()
    unet.load_state_dict({**sds['unet'], **sds['temb']}, assign=True)
    vae.load_state_dict(sds['vae'], assign=True)
    return pipe, unet.eval(), vae.eval()


def test_config3_full_size_plms_20_steps_512px(full_rig):
    from oracle import pipeline_oracle as PO
    from sdod.amd.pipeline import initial_latent
    pipe, unet, vae = full_rig
    ctx2 = pipe.encode_tokens(IDS_U, IDS_C)
    x_T = initial_latent(42, 0)
    img = pipe.generate_graphed(ctx2, x_T, steps=20, guidance=7.5, sampler='plms').clone()
    tr_gpu, tr_cpu = [], []
    z = pipe.sample_plms(ctx2, x_T, steps=20, guidance=7.5, trace=tr_gpu)
    img_steps = pipe.decode(z, mode=1)
    assert torch.equal(img, img_steps), 'trajectory graph and per-step path disagree'
    c = ctx2.float().cpu()
    z_ref = PO.plms_sample(unet, c[0:1], c[1:2], x_T, steps=20, scale=7.5, trace=tr_cpu)
    assert tr_gpu == tr_cpu
    r = rel_l2(z.cpu(), z_ref)
    img_ref = PO.decode_u8(vae, z_ref, mode=1)
    d = np.abs(img.cpu().numpy().astype(np.int32) - img_ref.astype(np.int32))
    print(f'config 3 full size: final latent rel-L2 {r:.3e}; uint8 max diff {int(d.max())}, within 1 LSB {float((d <= 1).mean()):.5f}')
    assert img.shape == (1, 512, 512, 3) and img.dtype == torch.uint8
    assert torch.isfinite(z).all() and r <= 2e-2, r
    assert float((d <= 2).mean()) >= 0.99


def _vae_case(hw, seed):
    from oracle import sd_torch as S
    from sdod.amd import engine as E
    vae = S.build(S.AutoencoderKLDecode, seed=1235)
    cfg = E.sd14_config(hw, hw)
    g = E.VaeDecoder(cfg, 1)
    g.load_state_dict(vae.state_dict())
    g.finalize()
    z = torch.randn(1, 4, hw, hw, generator=torch.Generator().manual_seed(seed)) * 0.18215 * 4
    with torch.no_grad():
        ref = vae(z)
    g.z.copy_(z)
    g.execute()
    torch.cuda.synchronize()
    eager = g.img.clone()
    g.execute(True)
    torch.cuda.synchronize()
    assert torch.equal(eager, g.img), 'hipGraph replay differs from eager execution'
    out = g.img.float().cpu().permute(0, 3, 1, 2)
    assert out.shape == (1, 3, 8 * hw, 8 * hw) and torch.isfinite(out).all()
    return rel_l2(out, ref), g.stats()


def test_vae_decoder_64_to_512_vs_oracle():
    r, st = _vae_case(64, 12)
    print('vae 64x64 -> 512x512 rel-L2', r, st)
    assert r <= 1e-2, r
    assert abs(st['flops'] / 2.5145e12 - 1) < 0.02, st     # 2,514.5 GFLOP per image


def test_config5_vae_decoder_96_to_768_vs_oracle():
    r, st = _vae_case(96, 13)
    print('vae 96x96 -> 768x768 rel-L2', r, st)
    assert r <= 1e-2, r


def test_config5_chain_768px_uint8_weights_vs_oracle():
    """BASELINE.json configs[4] as ONE chain at its full size: OpenCLIP ViT-H/14 text tower -> 5-step
    v-prediction PLMS (6 batch-2 UNet evaluations at the 96x96 latent, guidance 7.5) with the UNet's conv / linear weights kept
    affine uint8 in HBM (the reference's `quantize=8` path, todlc.py:105-108) -> VAE decode 96 -> 768 -> uint8, against the
    fp32 oracle on the SAME dequantised weights (qnn_context.cpp:1018-1033 arithmetic), injected x_T.  Tolerances as config 3:
    final latent rel-L2 <= 2e-2, >= 99 % of the 768x768x3 pixels within 2 LSB, scheduler trace exact."""
    from oracle import pipeline_oracle as PO, sd_torch as S
    from sdod.amd import engine as E, weights as Wt
    from sdod.amd.pipeline import Txt2Img, initial_latent
    cfg = E.sd21_config(96, 96)
    tables = {'unet': E.UNet(cfg, 2).param_table(), 'temb': E.Temb(cfg, 1).param_table(),
              'vae': E.VaeDecoder(cfg, 1).param_table(), 'text': E.TextEncoder(cfg, 1).param_table()}
    sds = {k: Wt.synthetic_state_dict(t, seed=2100 + i) for i, (k, t) in enumerate(tables.items())}
    for k in ('unet', 'temb'):
        sds[k] = Wt.quantize_state_dict(sds[k])
    pipe = Txt2Img(state_dicts=sds, images_per_gpu=1, latent_hw=96, model='sd21', weight_quant=True)
    assert pipe.unet.stats()['weight_bytes'] < 0.95e9          # the codes, not their fp16 image, are what sits in HBM
    ids_c = np.zeros(77, np.int64); ids_c[0] = 49406; ids_c[1:6] = (320, 1125, 539, 320, 2368); ids_c[6] = 49407   # open_clip pads with 0
    ids_u = np.zeros(77, np.int64); ids_u[0] = 49406; ids_u[1] = 49407
    ctx2 = pipe.encode_tokens(ids_u, ids_c)
    x_T = initial_latent(45, 0, (4, 96, 96))
    tr_gpu, tr_cpu = [], []
    z = pipe.sample_plms(ctx2, x_T, steps=5, guidance=7.5, trace=tr_gpu)
    img = pipe.decode(z, mode=1)
    torch.cuda.synchronize()
    assert img.shape == (1, 768, 768, 3) and img.dtype == torch.uint8

    deq = lambda sd: {k: (v.dequantize() if isinstance(v, Wt.QuantU8) else v) for k, v in sd.items()}
    with torch.device('meta'):
        unet = S.UNetModel(context_dim=1024, head_dim=64, use_linear=True)
        vae = S.AutoencoderKLDecode()
        clip = S.OpenClipTextModel(layers=23, run_layers=23)
    unet.load_state_dict({**deq(sds['unet']), **deq(sds['temb'])}, assign=True)
    vae.load_state_dict(sds['vae'], assign=True)
    clip.load_state_dict(sds['text'], assign=True)
    with torch.no_grad():
        c = clip(torch.from_numpy(np.stack([ids_u, ids_c])))
    rc = rel_l2(ctx2.float().cpu(), c)
    z_ref = PO.plms_sample(unet.eval(), c[0:1], c[1:2], x_T, steps=5, scale=7.5, trace=tr_cpu, parameterization='v')
    assert tr_gpu == tr_cpu
    r = rel_l2(z.cpu(), z_ref)
    img_ref = PO.decode_u8(vae.eval(), z_ref, mode=1)
    d = np.abs(img.cpu().numpy().astype(np.int32) - img_ref.astype(np.int32))
    print(f'config 5 chain @768px, uint8 UNet weights: context rel-L2 {rc:.3e}; final latent rel-L2 {r:.3e}; '
          f'uint8 max diff {int(d.max())}, within 2 LSB {float((d <= 2).mean()):.5f}')
    assert rc <= 5e-3 and torch.isfinite(z).all() and r <= 2e-2, (rc, r)
    assert float((d <= 2).mean()) >= 0.99
