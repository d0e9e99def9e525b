"""GPU test of the drop-in C API (include/libsdod.h), driven exactly as the reference's simple_app.cpp drives it:
setup(models_dir, 4, spatial, 8, steps, log, use_htp) -> generate_image(prompt, 7.5, &buf, &len) -> release.
Weights are synthetic .sdodw containers written to a temp models_dir (no checkpoint offline); the reduced latent
(16x16 -> 128x128 image) keeps the CPU oracle fast.  x_T is injected (RNG streams are not portable)."""
import ctypes
import os
import shutil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def models(tmp_path_factory, golden_dir):
    from sdod.amd import engine as E, weights as Wt
    d = tmp_path_factory.mktemp('models')
    cfg = E.sd14_config(16, 16)
    tables = {'unet': E.UNet(cfg, 2).param_table(), 'temb': E.Temb(cfg, 1).param_table(),
              'vae_decoder': E.VaeDecoder(cfg, 1).param_table(), 'text_encoder': E.TextEncoder(cfg, 1).param_table()}
    sds = {}
    for i, (k, t) in enumerate(tables.items()):
        sds[k] = Wt.synthetic_state_dict(t, seed=2000 + i, dtype=torch.float16)
        Wt.save(str(d / f'{k}.sdodw'), sds[k])
    shutil.copy(os.path.join(golden_dir, 'ctokenizer_synthetic.txt'), d / 'ctokenizer.txt')
    return str(d), sds


def test_generate_image_through_c_api_matches_python_loop_and_oracle(models, oracle_lib):
    from oracle import pipeline_oracle as PO, sd_torch as S
    from sdod.amd.host import LibSdod, Tokenizer
    from sdod.amd.pipeline import Txt2Img
    mdir, sds = models
    prompt = 'A photograph of an astronaut riding a horse'
    app = LibSdod(mdir + '/', latent_channels=4, latent_spatial=16, upscale_factor=8, steps=20, log_level=1, use_htp=1)
    assert app.status == 0, app.error()
    x_T = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(42))
    assert app.set_initial_latent(x_T.numpy()) == 0
    rc, img = app.generate(prompt, 7.5)
    assert rc == 0, app.error(rc)
    assert img.shape == (128, 128, 3) and img.dtype == np.uint8

    # (1) same image as the Python host loop over the same engine (DPM-Solver++ + reference CFG + mode-0 uint8).  Not
    # bit-identical by construction: the driver encodes cond / uncond as two batch-1 text-encoder runs, the Python loop
    # as one batch-2 run, and the per-shape tile autotuner may pick different accumulation orders for M=77 and M=154.
    tok = Tokenizer(mdir + '/ctokenizer.txt')
    pipe = Txt2Img(models_dir=mdir, images_per_gpu=1, latent_hw=16, tokenizer=tok)
    ctx2 = pipe.encode_prompt(prompt)
    z = pipe.sample_dpm(ctx2, x_T, steps=20, guidance=7.5)
    img_py = pipe.decode(z, mode=0).cpu().numpy()[0]
    d_py = np.abs(img.astype(np.int32) - img_py.astype(np.int32))
    print('C API vs Python loop: max diff', int(d_py.max()), 'differing pixels', float((d_py > 0).mean()))
    assert d_py.max() <= 2 and float((d_py <= 1).mean()) >= 0.99

    # (2) within tolerance of the CPU oracle of the driver loop (context.cpp:292-403)
    with torch.device('meta'):
        unet, vae, clip = S.UNetModel(), S.AutoencoderKLDecode(), S.ClipTextModel()
    unet.load_state_dict({k: v.float() for k, v in {**sds['unet'], **sds['temb']}.items()}, assign=True)
    vae.load_state_dict({k: v.float() for k, v in sds['vae_decoder'].items()}, assign=True)
    clip.load_state_dict({k: v.float() for k, v in sds['text_encoder'].items()}, assign=True)
    ids = np.stack([tok.encode(''), tok.encode(prompt)]).astype(np.int64)
    with torch.no_grad():
        c = clip(torch.from_numpy(ids))
    z_ref = PO.dpm_sample(unet.eval(), oracle_lib, c[0:1], c[1:2], x_T, steps=20, guidance=7.5)
    img_ref = PO.decode_u8(vae.eval(), z_ref, mode=0, oracle_lib=oracle_lib)[0]
    diff = np.abs(img.astype(np.int32) - img_ref.astype(np.int32))
    print('C API vs oracle: max diff', int(diff.max()), 'within 2 LSB', float((diff <= 2).mean()))
    assert float((diff <= 2).mean()) >= 0.99

    # (3) buffer reuse / ownership rules (libsdod.h:84-116): too small -> INVALID_ARGUMENT, larger -> reused, size written back
    L = app.lib
    small = (ctypes.c_ubyte * 16)()
    p = ctypes.cast(small, ctypes.POINTER(ctypes.c_ubyte)); n = ctypes.c_uint(16)
    assert L.libsdod_generate_image(app.ctx, prompt.encode(), 7.5, ctypes.byref(p), ctypes.byref(n)) == 2
    assert b'too small' in L.libsdod_get_last_error_extra_info(2, app.ctx)
    big = (ctypes.c_ubyte * (128 * 128 * 3 + 100))()
    p = ctypes.cast(big, ctypes.POINTER(ctypes.c_ubyte)); n = ctypes.c_uint(len(big))
    app.set_initial_latent(x_T.numpy())
    assert L.libsdod_generate_image(app.ctx, prompt.encode(), 7.5, ctypes.byref(p), ctypes.byref(n)) == 0
    assert n.value == 128 * 128 * 3 and np.array_equal(np.frombuffer(big, np.uint8, n.value).reshape(128, 128, 3), img)

    # (4) seeded generator: same seed -> same image, other seed -> different image; set_steps accepts other counts
    app.set_seed(7); _, a = app.generate(prompt, 7.5)
    app.set_seed(7); _, b = app.generate(prompt, 7.5)
    app.set_seed(8); _, c2 = app.generate(prompt, 7.5)
    assert np.array_equal(a, b) and not np.array_equal(a, c2)
    assert app.set_steps(8) == 0
    app.set_seed(7); rc, d8 = app.generate(prompt, 1.0)          # guidance 1: conditional branch only (context.cpp:359-360)
    assert rc == 0 and d8.shape == (128, 128, 3)
    assert app.set_steps(0) == 2                                  # INVALID_ARGUMENT, context stays usable

    # (5) ref-counting and use-after-release (libsdod.cpp:146-161)
    assert L.libsdod_ref_context(app.ctx) == 0
    assert app.release() == 0 and app.release() == 0
    assert app.release() == 1                                     # released: INVALID_CONTEXT, handle still detectable
    assert b'released' in L.libsdod_get_last_error_extra_info(1, None)


def test_generate_image_at_the_sample_apps_size_through_c_api(models, oracle_lib, capfd):
    """The C boundary at the size the reference's sample app runs (simple_app.cpp:9-33: 4, 64, 8, 20 -> a 512x512 image):
    setup -> set_initial_latent -> generate, the 512^2 uint8 image against the CPU oracle of the driver loop
    (context.cpp:292-403: DPM-Solver++(2M), reference CFG, mode-0 uint8), same tolerance as the 16x16 test above; and the
    four phase timers the reference logs at INFO (context.cpp:331, :381, :398, :402), here from stream events"""
    from oracle import pipeline_oracle as PO, sd_torch as S
    from sdod.amd.host import LibSdod, Tokenizer
    mdir, sds = models
    prompt = 'A photograph of an astronaut riding a horse'
    app = LibSdod(mdir + '/', latent_channels=4, latent_spatial=64, upscale_factor=8, steps=20, log_level=2, use_htp=1)
    assert app.status == 0, app.error()
    x_T = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(43))
    assert app.set_initial_latent(x_T.numpy()) == 0
    rc, img = app.generate(prompt, 7.5)            # first call: eager passes + graph capture
    assert rc == 0, app.error(rc)
    assert app.set_initial_latent(x_T.numpy()) == 0
    capfd.readouterr()
    rc, img2 = app.generate(prompt, 7.5)
    assert rc == 0 and img.shape == (512, 512, 3) and img.dtype == np.uint8
    assert np.array_equal(img, img2), 'same latent, same prompt: the replayed graphs must reproduce the eager image'
    log = capfd.readouterr().out
    import re
    its = [float(v) for v in re.findall(r'Single iteration took ([0-9.]+)ms', log)]
    cond = re.findall(r'Conditioning took ([0-9.]+)ms', log); dec = re.findall(r'Decoding took ([0-9.]+)ms', log)
    tot = re.findall(r'Image generation took ([0-9.]+)ms', log)
    assert len(its) == 20 and len(cond) == 1 and len(dec) == 1 and len(tot) == 1, log
    print(f'C API 512x512: conditioning {cond[0]} ms, iteration median {sorted(its)[10]:.2f} ms, decoding {dec[0]} ms, image {tot[0]} ms')
    # device-side phase times add up to (at most) the host's wall clock of the call, and decode no longer contains the sampler loop
    assert float(cond[0]) + sum(its) + float(dec[0]) <= float(tot[0]) * 1.02 + 1.0
    assert float(dec[0]) < 0.5 * float(tot[0]) and 1.0 < sorted(its)[10] < 50.0
    assert app.release() == 0

    tok = Tokenizer(mdir + '/ctokenizer.txt')
    with torch.device('meta'):
        unet, vae, clip = S.UNetModel(), S.AutoencoderKLDecode(), S.ClipTextModel()
    unet.load_state_dict({k: v.float() for k, v in {**sds['unet'], **sds['temb']}.items()}, assign=True)
    vae.load_state_dict({k: v.float() for k, v in sds['vae_decoder'].items()}, assign=True)
    clip.load_state_dict({k: v.float() for k, v in sds['text_encoder'].items()}, assign=True)
    ids = np.stack([tok.encode(''), tok.encode(prompt)]).astype(np.int64)
    with torch.no_grad():
        c = clip(torch.from_numpy(ids))
    z_ref = PO.dpm_sample(unet.eval(), oracle_lib, c[0:1], c[1:2], x_T, steps=20, guidance=7.5)
    img_ref = PO.decode_u8(vae.eval(), z_ref, mode=0, oracle_lib=oracle_lib)[0]
    diff = np.abs(img.astype(np.int32) - img_ref.astype(np.int32))
    print('C API 512x512 vs oracle: max diff', int(diff.max()), 'within 2 LSB', float((diff <= 2).mean()))
    assert float((diff <= 2).mean()) >= 0.99


def test_setup_failure_reports_through_error_table(tmp_path):
    from sdod.amd.host import LibSdod
    app = LibSdod(str(tmp_path), latent_spatial=16, steps=20)
    assert app.status == 2                                        # missing ctokenizer.txt -> INVALID_ARGUMENT
    desc, extra = app.error()
    assert desc == 'Invalid argument' and 'ctokenizer.txt' in extra
    assert app.ctx.value is not None and app.release() == 0       # *context was set and must still be released
