"""CPU tests (no GPU, no compute kernels): the C-ABI library loads, exports every symbol include/*.h declares, and its
host-side pieces (C++ tokenizer, DPM solver tables, libsdod handle/error conventions) match the reference-generated
golden vectors bit for bit."""
import ctypes
import json
import os
import re

import numpy as np
import pytest


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__ as ge
    so = os.path.join(ge.PKG, 'lib', 'libsdod.so')
    if not os.path.exists(so):
        ge.build()
    from sdod.amd import _lib
    return _lib.load('libsdod.so')


def test_every_declared_symbol_is_exported(lib):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    declared = set()
    for h in ('sdod_hip.h', 'sdod_engine.h', 'sdod_host.h', 'libsdod.h'):
        text = open(os.path.join(root, 'include', h)).read()
        declared |= set(re.findall(r'(?:SDOD_API|LIBSDOD_API)\s+[\w\s\*]+?\b((?:sdod|libsdod)_\w+)\s*\(', text))
    assert len(declared) >= 55, sorted(declared)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    from sdod.amd import _lib, engine, host
    bound = set(_lib.HIP_SYMBOLS + engine.ENGINE_SYMBOLS + host.HOST_SYMBOLS + host.LIBSDOD_SYMBOLS)
    assert not (declared - bound), declared - bound     # the Python side binds what the headers declare


def test_cpp_dpm_solver_bit_exact_vs_reference_golden(lib, golden_dir):
    from sdod.amd.host import DpmSolver
    for steps in (20, 50, 1, 2, 3, 8, 100):          # the path's step counts, then edge counts (all from the compiled reference)
        g = json.load(open(os.path.join(golden_dir, f'dpm_steps{steps}.json')))
        s = DpmSolver(1000, 0.00085, 0.0120)
        mts = s.prepare(steps)
        assert np.array_equal(mts.view(np.uint32), np.array(g['model_ts_bits'], np.uint32))
        for name in ('ts', 'log_alphas', 'lambdas', 'sigmas', 'alphas', 'phis', 'i2rs', 'all_t', 'all_log_alpha'):
            assert np.array_equal(s.table(name).view(np.uint32), np.array(g[name + '_bits'], np.uint32)), name
        x = np.array(g['x0_bits'], np.uint32).view(np.float32).copy()
        y_prev = np.zeros_like(x)
        for i, rec in enumerate(g['trajectory']):
            e = np.array(rec['eps_bits'], np.uint32).view(np.float32).copy()
            s.update_host(i, x, e, y_prev)
            assert np.array_equal(x.view(np.uint32), np.array(rec['x_bits'], np.uint32)), f'step {i}'
        assert s.coef(0)['order'] == 1                                                                    # quirk Q8 kept:
        assert all(s.coef(i)['order'] == 2 for i in {1, steps - 1} if 1 <= i < steps)                     # order 2 from step 1 on


def test_cpp_tokenizer_matches_reference_golden_and_oracle(lib, golden_dir):
    from oracle.tokenizer_oracle import TokenizerOracle
    from sdod.amd.host import Tokenizer
    g = json.load(open(os.path.join(golden_dir, 'tokenizer_synthetic.json')))
    vocab = os.path.join(golden_dir, g['vocab'])
    tok = Tokenizer(vocab)
    orc = TokenizerOracle(vocab, canonical_ws=True)
    assert (tok.start_token, tok.end_token) == (orc.start_token, orc.end_token) == (554, 555)
    for c in g['cases']:                       # ids produced by the reference's own tokenizer.cpp
        assert tok.encode(c['text']).tolist() == c['ids'], c['text']
    # beyond the reference's domain: Q3 inputs, non-blank whitespace, non-ASCII -- canonical semantics (oracle)
    for text in ['aabc', 'tthe', 'a\nb\r\nc', 'line one\n\nline two', 'café CafÉ', 'über straße', 'x' * 400,
                 'the ' * 90, "it's they'll we've", '']:
        assert tok.encode(text).tolist() == orc.tokenize(text), repr(text)
    assert tok.encode('A photograph of an astronaut riding a horse').tolist() == tok.encode('a  photograph of an astronaut riding a horse ').tolist()
    with pytest.raises(Exception):
        tok.encode(b'\xff\xfe'.decode('latin-1').encode('latin-1').decode('latin-1') and '\udcff')   # invalid UTF-8


def test_tokenizer_missing_file_is_an_invalid_argument(lib):
    from sdod.amd._lib import SdodError
    from sdod.amd.host import Tokenizer
    with pytest.raises(SdodError) as ei:
        Tokenizer('/nonexistent/ctokenizer.txt')
    assert ei.value.code == 2


def test_libsdod_handle_and_error_conventions_without_gpu(lib):
    """argument validation happens before any device work, so these paths run on a CPU-only box
    (reference semantics: libsdod.cpp:48-63, :66-72, :187-209)"""
    from sdod.amd import host
    L = host._host()
    assert L.libsdod_get_error_description(0) == b'No error'
    assert L.libsdod_get_error_description(2) == b'Invalid argument'
    assert L.libsdod_get_error_description(6) is None and L.libsdod_get_error_description(-1) is None
    assert L.libsdod_setup(None, b'.', 4, 64, 8, 20, 1, 1) == 2                      # context == NULL
    ctx = ctypes.c_void_p(1234)
    assert L.libsdod_setup(ctypes.byref(ctx), b'.', 4, 64, 8, 20, 1, 1) == 2         # *context != NULL
    ctx = ctypes.c_void_p()
    assert L.libsdod_setup(ctypes.byref(ctx), b'.', 4, 64, 8, 20, 9, 1) == 2         # invalid log level
    assert ctx.value is None
    info = L.libsdod_get_last_error_extra_info(2, None)
    assert info is not None and b'Invalid log_level' in info and b'capi.cpp' in info
    for fn in (L.libsdod_release, L.libsdod_ref_context):
        assert fn(None) == 1                                                          # INVALID_CONTEXT
    assert L.libsdod_set_steps(None, 20) == 1 and L.libsdod_set_log_level(None, 1) == 1
    assert b'context is nullptr' in L.libsdod_get_last_error_extra_info(1, None)
    fake = (ctypes.c_uint * 8)(0xdeadbeef, 1, 1, 0, 0, 0, 0, 0)
    assert L.libsdod_release(ctypes.cast(fake, ctypes.c_void_p)) == 1                 # magic mismatch
    assert b'magic' in L.libsdod_get_last_error_extra_info(1, None)
    assert L.libsdod_get_last_error_extra_info(99, None) is None


def test_weight_container_roundtrip(tmp_path):
    import torch
    from sdod.amd import weights as Wt
    sd = Wt.synthetic_state_dict([('a.weight', (8, 4, 3, 3)), ('a.bias', (8,)), ('n.weight', (8,)), ('tok.embedding.weight', (10, 4))], seed=3)
    assert abs(float(sd['n.weight'].mean()) - 1.0) < 0.2 and float(sd['tok.embedding.weight'].abs().max()) < 0.2
    again = Wt.synthetic_state_dict([('a.weight', (8, 4, 3, 3)), ('a.bias', (8,)), ('n.weight', (8,)), ('tok.embedding.weight', (10, 4))], seed=3)
    assert all(torch.equal(sd[k], again[k]) for k in sd)
    sd['h.weight'] = torch.randn(4, 4).half()
    p = tmp_path / 'w.sdodw'
    Wt.save(str(p), sd)
    back = Wt.load(str(p))
    assert list(back) == list(sd)
    assert all(torch.equal(sd[k], back[k]) and sd[k].dtype == back[k].dtype for k in sd)


def _conv_desc(n, h, w, c0, cout, c1=0, ups=False, stride=1, tail=0, tile=0, split=0):
    """a sdod_gemm_desc for a 3x3 convolution with fake (non-null, never dereferenced) pointers: planning calls only"""
    from sdod.amd._lib import GemmDesc
    d = GemmDesc()
    d.a = d.w = d.out = 0x1000
    d.a2 = 0x1000 if c1 else None
    d.a_mode = 1; d.ksize = 3; d.stride = stride; d.upsample = 1 if ups else 0
    d.n_img, d.h_in, d.w_in, d.c0, d.c1 = n, h, w, c0, c1
    ho, wo = (h << ups) // stride, (w << ups) // stride
    d.M, d.N, d.K = n * ho * wo, cout, 9 * (c0 + c1) + tail
    d.ldw, d.ldo = d.K, cout
    if tail:
        d.k_tail = 9 * (c0 + c1); d.t0 = 0x1000; d.tc0 = tail
    d.tile, d.split_k = tile, split
    return d


def test_halo_patch_tile_planning_without_a_gpu():
    """host side of conv_halo_kernel (gemm.hip: halo_geometry / make_plan): K slices are whole 64-channel chunks, a fused 1x1
    tail is one more slice, the in-kernel reduce needs counters, and geometries the tile cannot hold fall back to the generic
    plan (the launch would refuse them) -- sdod_gemm_plan / sdod_gemm_fixup / sdod_gemm_workspace_bytes only, no launches"""
    import ctypes
    from sdod.amd import _lib
    lib = _lib.hip()
    t, s = ctypes.c_int(), ctypes.c_int()

    def plan(d):
        assert lib.sdod_gemm_plan(ctypes.byref(d), ctypes.byref(t), ctypes.byref(s)) == 0
        return t.value, s.value

    # 64x64 x 320 -> 320 on the 128x80 tile (38): 5 chunks; asking for 2 slices gives ceil(5/3) = 2, for 8 at most 5
    assert plan(_conv_desc(2, 64, 64, 320, 320, tile=38, split=2)) == (38, 2)
    assert plan(_conv_desc(2, 64, 64, 320, 320, tile=38, split=8)) == (38, 5)
    assert plan(_conv_desc(2, 64, 64, 320, 320, tile=38, split=1)) == (38, 1)
    # the fused 1x1 skip (tail) is ONE extra slice
    assert plan(_conv_desc(2, 16, 16, 1280, 1280, tail=2560, tile=41, split=3)) == (41, 4)
    # workspace = slices x M x N fp32
    d = _conv_desc(2, 16, 16, 1280, 1280, tile=41, split=3)
    assert lib.sdod_gemm_workspace_bytes(ctypes.byref(d)) == 3 * 512 * 1280 * 4
    # in-kernel reduce: only with counters, a split plan, phase 0
    assert lib.sdod_gemm_fixup(ctypes.byref(d)) == 0
    d.fix_counters = 0x2000
    assert lib.sdod_gemm_fixup(ctypes.byref(d)) == 1
    d.phase = 1
    assert lib.sdod_gemm_fixup(ctypes.byref(d)) == 0
    d1 = _conv_desc(2, 64, 64, 320, 320, tile=38, split=1); d1.fix_counters = 0x2000
    assert lib.sdod_gemm_fixup(ctypes.byref(d1)) == 0
    assert lib.sdod_gemm_fixup_counters() >= 16384
    # nearest-2x upsampling is taken (patch cut from the source), stride 2 is not: the generic plan answers instead
    assert plan(_conv_desc(2, 32, 32, 640, 640, ups=True, tile=39, split=1)) == (39, 1)
    tile, splits = plan(_conv_desc(2, 32, 32, 640, 640, stride=2, tile=39, split=3))
    assert tile == 39 and splits == 3            # generic K split (90 slabs / 3), not chunk-granular
    # tile metadata for profilers: SPEC column 2 marks conv_halo_kernel<BM, BN, WM, WN, STAGES>
    info = (ctypes.c_int * 7)()
    assert lib.sdod_gemm_tile_info(38, info) == 0 and list(info) == [128, 80, 4, 1, 4, 2, 1]
    assert lib.sdod_gemm_num_tiles() >= 52
    # 96- / 192-row tiles (49..52) hold images whose rows are multiples of 3 (SD v2.1-768: 96 / 48 / 24 / 12): chunk-granular
    # plans there, the generic K split on a 64-pixel image (the launch would refuse it)
    assert lib.sdod_gemm_tile_info(51, info) == 0 and list(info) == [192, 80, 4, 1, 4, 2, 1]
    assert plan(_conv_desc(2, 96, 96, 320, 320, tile=52, split=8)) == (52, 5)          # 5 chunks at most
    assert plan(_conv_desc(2, 24, 24, 1280, 1280, tile=49, split=3)) == (49, 3)        # ceil(20 / 7) chunks per slice -> 3 slices
    assert plan(_conv_desc(2, 64, 64, 320, 320, tile=52, split=8)) == (52, 8)          # 45 slabs / 8: not chunk-granular


# the geometries of tests/test_kernels_gpu.py::test_conv3x3_halo_patch_tiles (n, h, w, cin, cout) and its tile list
HALO_TILES = list(range(37, 46)) + [49, 50, 51, 52]
HALO_CASES = [(2, 64, 64, 128, 320), (2, 32, 32, 192, 160), (2, 16, 16, 256, 256), (2, 8, 8, 320, 192), (1, 8, 8, 64, 64),
              (3, 8, 8, 128, 80), (1, 128, 128, 64, 128), (5, 16, 16, 64, 48),
              (2, 96, 96, 64, 160), (2, 48, 48, 128, 80), (2, 24, 24, 192, 128), (1, 12, 12, 64, 200), (3, 12, 12, 128, 64)]


def test_every_halo_tile_takes_some_geometry_of_the_gpu_test_matrix():
    """the GPU halo tests skip a (tile, geometry) pair the tile declines; a planner regression that declined everything would
    turn them green-by-skip.  sdod_gemm_halo_ok is the same host predicate the launch uses: every tile must take at least two
    of the listed geometries, the 64 / 128 / 256-row tiles none of the rows-multiple-of-3 images that do not fit them"""
    import ctypes
    from sdod.amd import _lib
    lib = _lib.hip()
    takes = {t: [c for c in HALO_CASES if lib.sdod_gemm_halo_ok(ctypes.byref(_conv_desc(c[0], c[1], c[2], c[3], c[4])), t)] for t in HALO_TILES}
    for t, cs in takes.items():
        assert len(cs) >= 2, (t, cs)
    assert (2, 64, 64, 128, 320) in takes[38] and (2, 64, 64, 128, 320) in takes[37]
    assert (2, 96, 96, 64, 160) in takes[49] and (2, 96, 96, 64, 160) not in takes[38]
    assert (2, 24, 24, 192, 128) in takes[51] and (2, 24, 24, 192, 128) in takes[52]
    # not a halo tile / not a 3x3 stride-1 convolution: never
    assert lib.sdod_gemm_halo_ok(ctypes.byref(_conv_desc(2, 64, 64, 128, 320)), 28) == 0
    assert lib.sdod_gemm_halo_ok(ctypes.byref(_conv_desc(2, 64, 64, 128, 320, stride=2)), 38) == 0


def test_xcd_panel_choice_minimises_the_bytes_the_eight_l2s_fetch():
    """sdod_gemm_xcd_panels: panels * A + (8 / panels) * W is smallest (gemm.hip: xcd_panels / tile_of); a forced value wins"""
    import ctypes
    from sdod.amd import _lib
    from sdod.amd._lib import GemmDesc
    lib = _lib.hip()

    def rows(m, n, k, tile, xcd=0):
        d = GemmDesc()
        d.a = d.w = d.out = 0x1000
        d.M, d.N, d.K, d.lda, d.ldw, d.ldo = m, n, k, k, k, n
        d.tile, d.split_k, d.xcd_panels = tile, 1, xcd
        return d

    # weight-heavy small-M Linear (16x16 level): A 1.3 MB, W 3.3 MB -> 4 panels x 2 bands: 4 A + 2 W = 11.8 MB (m-major: 27.5)
    assert lib.sdod_gemm_xcd_panels(ctypes.byref(rows(512, 1280, 1280, 28))) == 4
    # activation-heavy projection (64x64 level): A 5.2 MB, W 0.2 MB -> m-major, as before
    assert lib.sdod_gemm_xcd_panels(ctypes.byref(rows(8192, 320, 320, 31))) == 1
    # comparable operands: A 2.6 MB, W 0.8 MB over 32 x 10 tiles: 2 panels would fetch 8.5 MB against 9.2 -- not worth cutting
    # every output row between XCDs (the choice moves only for a saving of 20 % or more)
    assert lib.sdod_gemm_xcd_panels(ctypes.byref(rows(2048, 640, 640, 28))) == 1
    assert lib.sdod_gemm_xcd_panels(ctypes.byref(rows(2048, 640, 640, 28, xcd=2))) == 2
    # never more panels than n-tiles
    assert lib.sdod_gemm_xcd_panels(ctypes.byref(rows(128, 128, 8192, 28))) <= 2
    assert lib.sdod_gemm_xcd_panels(ctypes.byref(rows(8192, 320, 320, 31, xcd=4))) == 2   # 64x160 tile: two n-tiles only
    assert lib.sdod_gemm_xcd_panels(ctypes.byref(rows(8192, 2560, 320, 14, xcd=4))) == 4
    # deep convolution: the image is 0.16 MB, the weights 29 MB
    assert lib.sdod_gemm_xcd_panels(ctypes.byref(_conv_desc(2, 8, 8, 1280, 1280, tile=42, split=5))) == 8


def test_apanel_tile_eligibility_without_a_gpu():
    """gemm_apanel_kernel (tiles 53..55) takes the transformer's short-K Linears whose row panel fits LDS, nothing else"""
    import ctypes
    from sdod.amd import _lib
    from sdod.amd._lib import GemmDesc
    lib = _lib.hip()

    def rows(m, n, k, **kw):
        d = GemmDesc()
        d.a = d.w = d.out = 0x1000
        d.M, d.N, d.K, d.lda, d.ldw, d.ldo = m, n, k, k, k, n
        for key, v in kw.items():
            setattr(d, key, v)
        return d

    ok = lambda d, t: lib.sdod_gemm_panel_ok(ctypes.byref(d), t)
    assert ok(rows(8192, 2560, 320, geglu=1, ldo=1280), 53) == 1 and ok(rows(8192, 960, 320, ln=1), 53) == 1
    assert ok(rows(2048, 5120, 640), 54) == 1 and ok(rows(512, 10240, 1280), 55) == 1
    assert ok(rows(2048, 5120, 640), 53) == 0                        # 128 rows x 640: 160 KB panel
    assert ok(rows(512, 10240, 1280), 54) == 0
    assert ok(rows(8192, 320, 128), 53) == 0                         # two slabs of K: the ring kernel's business
    assert ok(rows(8192, 2560, 320, split_k=2), 53) == 0 and ok(rows(8192, 2560, 320, wq=1), 53) == 0
    assert ok(rows(8192, 2560, 320, bias_on_m=1), 53) == 0 and ok(rows(8192, 324, 320), 53) == 0
    assert ok(_conv_desc(2, 64, 64, 320, 320), 53) == 0 and ok(rows(8192, 2560, 320), 28) == 0
    t, s = ctypes.c_int(), ctypes.c_int()
    d = rows(8192, 2560, 320, tile=53, split_k=0)
    assert lib.sdod_gemm_plan(ctypes.byref(d), ctypes.byref(t), ctypes.byref(s)) == 0 and (t.value, s.value) == (53, 1)
    info = (ctypes.c_int * 7)()
    assert lib.sdod_gemm_tile_info(54, info) == 0 and list(info)[:4] == [64, 128, 2, 2] and info[5] == 3
    assert lib.sdod_gemm_num_tiles() == 61
    assert lib.sdod_gemm_tile_info(57, info) == 0 and list(info)[:6] == [128, 160, 2, 2, 3, 1]   # ring tiles for the softmax-epilogue GEMM


def test_plan_of_the_folded_cross_attention_gemms_without_a_gpu():
    """host side of sdod_gemm_desc::{softmax_cols, w_img_stride} (include/sdod_hip.h): the plan only ever names a ring tile whose
    waves own 80 columns for the softmax epilogue, a tile whose rows divide the image for per-image weights, never split-K with
    per-image vectors; the A-panel tiles refuse both (sdod_gemm_plan is host-only)"""
    import ctypes
    from sdod.amd import _lib
    from sdod.amd._lib import GemmDesc
    lib = _lib.hip()

    def rows(m, n, k, **kw):
        d = GemmDesc()
        d.a = d.w = d.out = 0x1000
        d.M, d.N, d.K, d.lda, d.ldw, d.ldo = m, n, k, k, k, n
        for key, v in kw.items():
            setattr(d, key, v)
        return d

    def plan(d):
        t, s = ctypes.c_int(), ctypes.c_int()
        assert lib.sdod_gemm_plan(ctypes.byref(d), ctypes.byref(t), ctypes.byref(s)) == 0
        return t.value, s.value

    wave80 = {21, 22, 31, 35, 48, 56, 57, 58, 59, 60}
    info = (ctypes.c_int * 7)()
    for t in wave80:                                            # "80 columns per wave" is a property of the tile table
        assert lib.sdod_gemm_tile_info(t, info) == 0 and info[1] // info[3] == 80, t
    score = dict(ln=1, softmax_cols=80, w_img_stride=640 * 320, vec_img_stride=640)
    for tile in list(range(1, lib.sdod_gemm_num_tiles() + 1)):
        t, s = plan(rows(8192, 640, 320, rows_per_img=4096, tile=tile, split_k=0, **score))
        assert t in wave80 and s == 1, (tile, t, s)
        assert lib.sdod_gemm_tile_info(t, info) == 0 and 4096 % info[0] == 0
    # the 8x8 level: 64 rows per image -- a 128-row tile would straddle two weight matrices
    for tile in (57, 23, 14, 9, 53):
        t, s = plan(rows(128, 640, 1280, rows_per_img=64, tile=tile, split_k=0, **score))
        assert t in wave80 and lib.sdod_gemm_tile_info(t, info) == 0 and 64 % info[0] == 0, (tile, t)
    # second GEMM: per-image weights, shared bias -- any ring tile whose rows divide the image, split-K allowed
    out = dict(w_img_stride=320 * 640)
    t, s = plan(rows(8192, 320, 640, rows_per_img=4096, tile=31, split_k=0, **out))
    assert t == 31
    t, s = plan(rows(128, 1280, 640, rows_per_img=64, tile=23, split_k=2, **out))
    assert lib.sdod_gemm_tile_info(t, info) == 0 and 64 % info[0] == 0 and info[5] == 1 and s == 2
    t, s = plan(rows(8192, 320, 640, rows_per_img=4096, tile=53, split_k=0, **out))
    assert lib.sdod_gemm_tile_info(t, info) == 0 and info[5] != 3     # not an A-panel tile
    assert lib.sdod_gemm_panel_ok(ctypes.byref(rows(8192, 2560, 320, w_img_stride=2560 * 320, rows_per_img=4096)), 53) == 0
    assert lib.sdod_gemm_panel_ok(ctypes.byref(rows(8192, 640, 320, softmax_cols=80)), 53) == 0


def test_group_norm_workspace_layout_keeps_the_barrier_lines_out_of_reach():
    """one grow-only workspace serves every (n, groups); the grid barrier's counter lines sit at a FIXED offset in
    front, so the partial sums of no layout can land on the lines another layout uses"""
    import ctypes
    from sdod.amd import _lib
    lib = _lib.hip()
    offs = {}
    for n, g in [(1, 16), (2, 32), (1, 32), (4, 32), (5, 8), (16, 32)]:
        v = [ctypes.c_size_t() for _ in range(5)]
        assert lib.sdod_group_norm_layout(n, g, *[ctypes.byref(x) for x in v]) == 0
        sync, part, stats, shift, end = [x.value for x in v]
        assert sync == 0 and part >= 26 * 128 and part % 128 == 0      # 25 counter lines + the sticky timeout word
        assert part < stats < shift < end == lib.sdod_group_norm_workspace_bytes(n, g)
        assert stats - part == n * 1024 * g * 2 * 4 and shift - stats == n * g * 2 * 4 and end - shift == n * g * 4
        offs[(n, g)] = part
    assert len(set(offs.values())) == 1                                 # the same barrier region for every layout
    assert lib.sdod_group_norm_layout(0, 32, None, None, None, None, None) != 0
    assert lib.sdod_group_norm_status() == 0 and lib.sdod_group_norm_clear_error() == 0   # no device: nothing to report


def test_group_norm_path_selection_without_a_gpu():
    """sdod_group_norm_path: which kernel a shape gets; without a device there is no CU count, so the grid-barrier kernel is
    never chosen here (path 0 is covered by the GPU suite)"""
    from sdod.amd import _lib
    lib = _lib.hip()
    assert lib.sdod_group_norm_path(2, 1024, 640, 0, 32, 0) == 1        # (image, group) one-launch kernel
    assert lib.sdod_group_norm_path(2, 256, 1280, 640, 32, 0) == 1      # concat source
    assert lib.sdod_group_norm_path(1, 100, 64, 0, 32, 1) in (2, 3)     # fp32: LDS / two-pass kernels
    assert lib.sdod_group_norm_path(2, 64, 30, 0, 32, 0) == -1          # channels not divisible by groups
    assert lib.sdod_group_norm_workspace_bytes(2, 32) == (1024 + 2 * 1024 * 32 * 2 + 2 * 32 * 3) * 4


def test_argument_checks_answer_before_any_launch():
    """shape constraints of the kernel entry points are host-side checks: they answer (status + message) without a device"""
    import ctypes
    from sdod.amd import _lib
    lib = _lib.hip()
    buf = (ctypes.c_char * 64)()
    P = ctypes.c_void_p
    for c in (12, 4096):   # not a multiple of 8; wider than the 64-lanes-per-row kernel holds (6 x 16-byte chunks per lane)
        assert lib.sdod_layer_norm_f16(ctypes.cast(buf, P), ctypes.cast(buf, P), None, None, 4, c, 1e-5, None) != 0
        assert b'LayerNorm needs' in lib.sdod_hip_last_error()
    assert lib.sdod_layer_norm_f16(None, None, None, None, 4, 64, 1e-5, None) != 0
    assert b'null pointer' in lib.sdod_hip_last_error()


def test_shipped_tile_table_is_well_formed():
    """tune/gfx950.tune (the table every process builds its launch lists from): 14 key fields + pick per line, no duplicate
    keys, tile ids the library knows, split factors in range; halo-patch picks only on 3x3 stride-1 convolutions whose
    geometry the tile holds and A-panel picks only on GEMMs the panel kernel takes (sdod_gemm_halo_ok / sdod_gemm_panel_ok on
    a descriptor rebuilt from the key) -- a stale or hand-edited table fails here, not on the GPU box"""
    import ctypes
    import os
    from sdod.amd import _lib
    from sdod.amd._lib import GemmDesc
    lib = _lib.hip()
    path = os.path.join(os.path.dirname(_lib.LIB_DIR), 'tune', 'gfx950.tune')
    assert os.path.exists(path), path
    ntiles = lib.sdod_gemm_num_tiles()
    seen = set()
    info = (ctypes.c_int * 7)()
    n_halo = n_panel = 0
    for ln, line in enumerate(open(path), 1):
        v = line.split()
        assert len(v) == 15, (ln, line)
        v = [int(x) for x in v]
        key = tuple(v[:14])
        assert key not in seen, f'duplicate key on line {ln}'
        seen.add(key)
        tile, split = v[14] % 1000, v[14] // 1000
        assert 1 <= tile <= ntiles and 1 <= split <= 64, (ln, v[14])
        assert lib.sdod_gemm_tile_info(tile, info) == 0
        a_mode, M, N, K, c0, c1, stride, ups, ksize, h_in, flags, lda, w_in, n_img = key
        if info[5] in (2, 3):
            d = GemmDesc()
            d.a = d.w = d.out = 0x1000
            d.a2 = 0x1000 if c1 else None
            d.a_mode, d.M, d.N, d.K, d.c0, d.c1, d.stride, d.upsample, d.ksize = a_mode, M, N, K, c0, c1, stride, ups, ksize
            d.h_in, d.w_in, d.n_img, d.lda, d.ldw = h_in, w_in, n_img, lda, K
            d.geglu = 1 if flags & 2 else 0
            d.ldo = N // 2 if d.geglu else N
            d.ln = 1 if flags & (1 << 30) else 0
            d.wq = 1 if flags & (1 << 29) else 0
            d.tc0, d.tc1 = (flags >> 2) & 4095, (flags >> 14) & 4095
            if d.tc0:
                d.k_tail = K - d.tc0 - d.tc1; d.t0 = 0x1000; d.t1 = 0x1000 if d.tc1 else None
            if flags & 1:
                d.residual = 0x1000; d.ldr = d.ldo
            d.tile, d.split_k = tile, split
        if info[5] == 2:   # conv_halo_kernel: the key must be a 3x3 stride-1 convolution (a_mode 1, stride 1, ksize 3) the tile holds
            assert a_mode == 1 and stride == 1 and ksize == 3 and w_in > 0 and n_img > 0, (ln, line)
            assert lib.sdod_gemm_halo_ok(ctypes.byref(d), tile) == 1, (ln, line)
            n_halo += 1
        if info[5] == 3:   # gemm_apanel_kernel
            d.split_k = 1
            assert a_mode == 0 and split == 1 and lib.sdod_gemm_panel_ok(ctypes.byref(d), tile) == 1, (ln, line)
            n_panel += 1
    assert len(seen) >= 600 and n_halo >= 100, (len(seen), n_halo, n_panel)
