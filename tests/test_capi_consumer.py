"""A compiled C consumer of include/libsdod.h (tests/capi_consumer.c), the boundary the reference's own sample application
crosses (csrc/libsdod/test/simple_app.cpp:7-36): every other boundary test goes through ctypes, this one through gcc and
the dynamic linker, as a fresh process.
  * CPU: it compiles with -Wall -Werror as C99 against the header, links against lib/libsdod.so, and a models_dir
    without files fails the way the sample app expects (status -> description + extra info, handle released, exit 1);
  * GPU: setup -> generate -> output.bin -> release against synthetic .sdodw containers; the image equals the one the
    ctypes path produces for the same seedless run only in size (x_T is drawn from std::random_device, context.cpp:16),
    so the file is checked for size and for not being constant."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'stable-diffusion-on-device_amd', 'lib')


@pytest.fixture(scope='module')
def consumer(tmp_path_factory):
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(LIB, 'libsdod.so')):
        ge.build()
    exe = str(tmp_path_factory.mktemp('capi') / 'capi_consumer')
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Wextra', '-Werror', '-I', os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'tests', 'capi_consumer.c'), '-o', exe, '-L', LIB, '-lsdod', f'-Wl,-rpath,{LIB}'])
    return exe


def test_c_consumer_links_and_fails_gracefully_without_models(consumer, tmp_path):
    r = subprocess.run([consumer, str(tmp_path), '64', '20', str(tmp_path / 'output.bin')], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1, (r.returncode, r.stdout, r.stderr)
    assert 'Initialization error: Invalid argument;' in r.stdout and 'ctokenizer.txt' in r.stdout
    assert not (tmp_path / 'output.bin').exists()


@pytest.mark.gpu
def test_c_consumer_generates_an_image(consumer, tmp_path):
    from sdod.amd import engine as E, weights as Wt
    cfg = E.sd14_config(16, 16)
    for stem, cls, batch, seed in (('unet', E.UNet, 2, 1234), ('temb', E.Temb, 1, 1235), ('vae_decoder', E.VaeDecoder, 1, 1236),
                                   ('text_encoder', E.TextEncoder, 1, 1237)):
        Wt.save(str(tmp_path / f'{stem}.sdodw'), Wt.synthetic_state_dict(cls(cfg, batch).param_table(), seed=seed))
    with open(os.path.join(ROOT, 'tests', 'golden', 'ctokenizer_synthetic.txt'), 'rb') as f:
        (tmp_path / 'ctokenizer.txt').write_bytes(f.read())
    out = tmp_path / 'output.bin'
    r = subprocess.run([consumer, str(tmp_path), '16', '4', str(out), 'abc abc'], capture_output=True, text=True, timeout=600)
    print(r.stdout[-1500:], r.stderr[-1500:])
    assert r.returncode == 0, r.returncode
    img = np.fromfile(out, np.uint8)
    assert img.size == 3 * 128 * 128 and img.std() > 0          # show_output.py:5-6 layout: HWC uint8
