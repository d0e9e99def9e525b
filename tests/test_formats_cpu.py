"""Host-side data formats either side of the hot path: checkpoint -> weight containers, the tokenizer
vocabulary file, output.bin / PNG, and the per-op latency tables.  No GPU, no compute calls."""
import gzip
import json
import os
import struct
import zlib

import numpy as np
import pytest
import torch


# ------------------------------------------------------------------ 8f-1 checkpoint loader
def _ldm_checkpoint_on_meta():
    """the key set + shapes of a CompVis sd-v1-x checkpoint, as meta tensors (public ldm / HF layouts restated in
    oracle/sd_torch.py, whose parameter totals match the known model sizes)"""
    from oracle import sd_torch as S
    with torch.device('meta'):
        unet, vae, clip = S.UNetModel(), S.AutoencoderKLDecode(), S.ClipTextModel()
    sd = {}
    sd.update({'model.diffusion_model.' + k: v for k, v in unet.state_dict().items()})
    sd.update({'first_stage_model.' + k: v for k, v in vae.state_dict().items()})
    sd.update({'cond_stage_model.transformer.' + k: v for k, v in clip.state_dict().items()})
    return sd, unet, vae, clip


def test_checkpoint_split_covers_every_parameter_exactly():
    from sdod.amd import convert
    sd, unet, vae, clip = _ldm_checkpoint_on_meta()
    junk = {'model_ema.decay': torch.empty(()), 'first_stage_model.encoder.conv_in.weight': torch.empty(128, 3, 3, 3, device='meta'),
            'cond_stage_model.transformer.text_model.embeddings.position_ids': torch.empty(1, 77, dtype=torch.int64, device='meta'),
            'alphas_cumprod': torch.empty(1000, device='meta')}
    parts, unused = convert.split_state_dict({**sd, **junk})
    assert sorted(unused) == sorted(junk)                        # every real parameter is consumed, only junk is left
    assert set(parts) == {'unet', 'temb', 'vae', 'text'}
    # UNet weights are shared out between the UNET and TEMB graphs with nothing lost or duplicated
    assert set(parts['unet']) | set(parts['temb']) == set(unet.state_dict())
    assert not set(parts['unet']) & set(parts['temb'])
    assert all(k.startswith('time_embed.') or '.emb_layers.' in k for k in parts['temb'])
    assert set(parts['vae']) == set(vae.state_dict()) and set(parts['text']) == set(clip.state_dict())
    assert all(t.dtype == torch.float16 for p in parts.values() for t in p.values())
    n = lambda names, m: sum(m.state_dict()[k].numel() for k in names)
    assert n(list(parts['unet']) + list(parts['temb']), unet) == 859_520_964       # SURVEY Appendix B known answers
    assert n(parts['vae'], vae) == 49_490_199 and n(parts['text'], clip) == 123_060_480


def test_checkpoint_split_reports_missing_and_mismatched():
    from sdod.amd import convert
    sd, *_ = _ldm_checkpoint_on_meta()
    tables = convert.parameter_tables()
    short = dict(sd)
    del short['model.diffusion_model.out.2.bias']
    with pytest.raises(KeyError, match='out.2.bias'):
        convert.split_state_dict(short, tables)
    bad = dict(sd)
    bad['first_stage_model.decoder.conv_in.weight'] = torch.empty(512, 4, 1, 1, device='meta')
    with pytest.raises(ValueError, match='decoder.conv_in.weight'):
        convert.split_state_dict(bad, tables)
    # transformers >= 5 checkpoints drop the `text_model.` level: still accepted
    flat = {(k.replace('cond_stage_model.transformer.text_model.', 'cond_stage_model.transformer.')): v for k, v in sd.items()}
    parts, _ = convert.split_state_dict(flat, tables)
    assert len(parts['text']) == 196


@pytest.mark.parametrize('ext', ['.safetensors', '.ckpt'])
def test_convert_writes_the_four_containers(tmp_path, monkeypatch, ext):
    """file -> models_dir round trip on a miniature table (the full-size one is 2 GB: covered by key/shape above)"""
    from sdod.amd import convert, weights as Wt
    tables = {'unet': [('out.2.weight', (4, 8, 3, 3)), ('out.2.bias', (4,))], 'temb': [('time_embed.0.weight', (8, 4))],
              'vae': [('post_quant_conv.weight', (4, 4, 1, 1))], 'text': [('text_model.final_layer_norm.weight', (16,))]}
    monkeypatch.setattr(convert, 'parameter_tables', lambda cfg=None: tables)
    g = torch.Generator().manual_seed(5)
    sd = {convert.GRAPHS[gr][0] + n: torch.randn(s, generator=g) for gr, t in tables.items() for n, s in t}
    sd['model_ema.num_updates'] = torch.zeros(())
    src = str(tmp_path / ('model' + ext))
    if ext == '.safetensors':
        from safetensors.torch import save_file
        save_file(sd, src)
    else:
        torch.save({'state_dict': sd, 'global_step': 7}, src)
    out = tmp_path / 'models'
    written = convert.convert(src, str(out))
    assert sorted(os.path.basename(p) for p in written) == ['temb.sdodw', 'text_encoder.sdodw', 'unet.sdodw', 'vae_decoder.sdodw']
    back = Wt.load(str(out / 'unet.sdodw'))
    assert list(back) == ['out.2.weight', 'out.2.bias']
    assert torch.equal(back['out.2.weight'], sd['model.diffusion_model.out.2.weight'].half())
    assert torch.equal(Wt.load(str(out / 'text_encoder.sdodw'))['text_model.final_layer_norm.weight'],
                       sd['cond_stage_model.transformer.text_model.final_layer_norm.weight'].half())


# ------------------------------------------------------------------ 8f-2 tokenizer vocabulary file
def test_tokenizer_file_matches_reference_format_fixture(tmp_path, golden_dir):
    """tests/golden/ctokenizer_synthetic.txt was written in the format of gen_tokenizer_file.py:33-42 (oracle/gen_golden.py)
    for the synthetic merge list; the product generator must emit the same bytes from a CLIP-style .gz and a HF directory"""
    from oracle.tokenizer_oracle import SYNTHETIC_MERGES
    from sdod.amd import tokenizer_file as TF
    want = open(os.path.join(golden_dir, 'ctokenizer_synthetic.txt'), 'rb').read()
    lines = ['"bpe_simple_vocab_16e6.txt#version: 0.2'] + [f'{a} {b}' for a, b in SYNTHETIC_MERGES] + ['never used', '']
    gz = tmp_path / 'bpe.txt.gz'
    with gzip.open(gz, 'wb') as f:
        f.write('\n'.join(lines).encode('utf-8'))
    out = TF.generate(str(gz), str(tmp_path / 'c1.txt'), limit=len(SYNTHETIC_MERGES))
    assert open(out, 'rb').read() == want
    hf = tmp_path / 'hf'
    hf.mkdir()
    (hf / 'merges.txt').write_text('\n'.join(['#version: 0.2'] + [f'{a} {b}' for a, b in SYNTHETIC_MERGES]) + '\n', encoding='utf-8')
    vocab = {tok: i for i, tok in enumerate(TF.vocabulary(list(SYNTHETIC_MERGES)))}
    (hf / 'vocab.json').write_text(json.dumps(vocab), encoding='utf-8')
    out2 = TF.generate(str(hf), str(tmp_path / 'c2.txt'))
    assert open(out2, 'rb').read() == want
    vocab['th'], vocab['in'] = vocab['in'], vocab['th']           # a vocab.json that disagrees is caught
    (hf / 'vocab.json').write_text(json.dumps(vocab), encoding='utf-8')
    with pytest.raises(ValueError, match='disagrees'):
        TF.generate(str(hf), str(tmp_path / 'c3.txt'))


def test_byte_symbol_table_known_answers():
    from sdod.amd import tokenizer_file as TF
    sym = TF.byte_symbols()
    assert len(sym) == 256 and len(set(sym)) == 256
    assert sym[0] == '!' and sym[93] == '~' and sym[94] == '¡' and sym[187] == 'ÿ'
    assert sym[188] == 'Ā' and sym[255] == 'Ń'           # byte 0 -> U+0100 ... byte 0xAD -> U+0143
    assert TF.CLIP_MERGES == 48894                                   # 512 + 48894 + SOT + EOT = 49408 ids
    assert len(TF.vocabulary([('a', 'b')])) == 515
    with pytest.raises(ValueError):
        TF.write(os.devnull, [('a b', 'c')])


def test_generated_file_drives_the_c_tokenizer(tmp_path):
    """end to end: generator -> ctokenizer.txt -> libsdod tokenizer ids (SOT/EOT = line count, +1)"""
    from oracle.tokenizer_oracle import SYNTHETIC_MERGES
    from sdod.amd import tokenizer_file as TF
    from sdod.amd.host import Tokenizer
    path = TF.write(str(tmp_path / 'ctokenizer.txt'), list(SYNTHETIC_MERGES))
    tok = Tokenizer(path)
    ids = list(tok.encode('the horse'))
    v = {t: i for i, t in enumerate(TF.vocabulary(list(SYNTHETIC_MERGES)))}
    sot, eot = v['<|startoftext|>'], v['<|endoftext|>']
    assert ids[:4] == [sot, v['the</w>'], v['horse</w>'], eot] and len(ids) == 77 and set(ids[3:]) == {eot}


# ------------------------------------------------------------------ 8f-3 output.bin / PNG
def test_output_bin_and_png_roundtrip(tmp_path):
    from sdod.amd import image_io as IO
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (24, 40, 3), dtype=np.uint8)
    p = IO.write_output_bin(str(tmp_path / 'output.bin'), torch.from_numpy(img)[None])     # [1,H,W,3] as generate() returns
    assert os.path.getsize(p) == 24 * 40 * 3
    assert np.array_equal(IO.read_output_bin(p, 24, 40), img)
    with pytest.raises(ValueError):
        IO.read_output_bin(p)                                                               # not 512x512
    q = IO.write_png(str(tmp_path / 'o.png'), img)
    raw = open(q, 'rb').read()
    assert raw[:8] == b'\x89PNG\r\n\x1a\n' and raw[12:16] == b'IHDR' and raw[-8:-4] == b'IEND'
    assert struct.unpack('>II', raw[16:24]) == (40, 24)
    assert np.array_equal(IO.read_png(q), img)
    with pytest.raises(ValueError):
        IO.write_png(str(tmp_path / 'bad.png'), img.astype(np.float32))


@pytest.mark.parametrize('ftype', [1, 2, 3, 4])
def test_png_reader_undoes_every_scanline_filter(tmp_path, ftype):
    from sdod.amd import image_io as IO
    rng = np.random.default_rng(ftype)
    img = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    flat = img.reshape(5, 21).astype(np.int32)
    rows = bytearray()
    for y in range(5):
        rows.append(ftype)
        for x in range(21):
            a = flat[y, x - 3] if x >= 3 else 0
            b = flat[y - 1, x] if y else 0
            c = flat[y - 1, x - 3] if (y and x >= 3) else 0
            if ftype == 1:
                pred = a
            elif ftype == 2:
                pred = b
            elif ftype == 3:
                pred = (a + b) // 2
            else:
                pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            rows.append((flat[y, x] - pred) & 255)
    ch = lambda tag, body: struct.pack('>I', len(body)) + tag + body + struct.pack('>I', zlib.crc32(tag + body) & 0xFFFFFFFF)
    p = tmp_path / 'f.png'
    p.write_bytes(b'\x89PNG\r\n\x1a\n' + ch(b'IHDR', struct.pack('>IIBBBBB', 7, 5, 8, 2, 0, 0, 0)) +
                  ch(b'IDAT', zlib.compress(bytes(rows))) + ch(b'IEND', b''))
    assert np.array_equal(IO.read_png(str(p)), img)


# ------------------------------------------------------------------ 8f-4 per-op latency tables
def test_latency_tables_by_launch_and_by_op_type():
    from sdod.amd import analyze as A
    table = [('gemm_t14', 1e9, 1e6), ('group_norm', 0, 1e6), ('attn_d40', 1e9, 1e6), ('gemm_t3', 1e8, 1e5), ('gemm_t8_splitk', 1e8, 1e5)]
    details = ['conv3 M8192 N320 K2880 x1', 'n2 hw4096 c320', 'B2 h8 lq4096 lk4096', 'rows M8192 N320 K320 x1', 'conv1 M128 N4 K64 x2']
    ms = [0.040, 0.020, 0.130, 0.010, 0.005]
    slow, by_type, total = A.summarize(ms, table, details, top=2)
    assert [s[0].split()[1] for s in slow] == ['attn_d40', 'gemm_t14'] and abs(total - 205.0) < 1e-6
    assert [r[0] for r in by_type][:2] == ['Attention (QK^T softmax PV)', 'Conv 3x3 (implicit GEMM)']
    assert abs(sum(r[2] for r in by_type) - 100.0) < 0.05 and sum(r[3] for r in by_type) == 5
    assert A.op_type('gemm_t3', 'rows M1 N1 K64 x1') == 'Linear / MatMul' and A.op_type('gemm_t8', 'conv1 ...') == 'Conv 1x1'
    text = A.format_table(by_type, ['Op.', 'Latency (us)', '% Latency', 'Launches'])
    assert 'GroupNorm(+SiLU)' in text and 'Launches' in text


# ------------------------------------------------------------------ config 5: uint8 affine weights (QNN format) + SD 2.1 shapes
def test_u8_weight_format_matches_reference_dequant_semantics(tmp_path, oracle_lib):
    """weights.QuantU8 = the reference's QNN weight encoding real = (q + offset) * scale, q unsigned, offset <= 0
    (qnn_context.cpp:1018-1033); its dequantisation must equal the C oracle of that function bit for bit"""
    import ctypes
    from sdod.amd import weights as Wt
    g = torch.Generator().manual_seed(11)
    w = torch.randn(64, 96, generator=g) * 0.05 + 0.01
    q = Wt.quantize_u8(w)
    assert q.q.dtype == torch.uint8 and q.offset <= 0 and q.scale > 0 and tuple(q.shape) == (64, 96)
    deq = q.dequantize()
    assert float((deq - w).abs().max()) <= 0.5001 * q.scale                          # round-to-nearest on a 255-step grid
    assert int(q.q.min()) == 0 and int(q.q.max()) == 255                             # the min/max range is used fully
    ref = np.zeros(w.numel(), np.float32)
    raw = np.ascontiguousarray(q.q.numpy().reshape(-1))
    oracle_lib.oracle_dequant_u8.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_uint, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_float]
    oracle_lib.oracle_dequant_u8(ref.ctypes.data, raw.ctypes.data, q.offset, q.scale, raw.size, 0, 0, 1.0)
    assert np.array_equal(deq.numpy().reshape(-1), ref)
    zero = Wt.quantize_u8(torch.zeros(4, 4))                                           # degenerate range
    assert torch.equal(zero.dequantize(), torch.zeros(4, 4))
    pos = Wt.quantize_u8(torch.rand(8, 8, generator=g) + 1.0)                          # all-positive tensor: 0 stays representable
    assert pos.offset == 0
    sd = Wt.quantize_state_dict({'a.weight': w, 'a.bias': torch.randn(64, generator=g), 'n.weight': torch.ones(64)})
    assert isinstance(sd['a.weight'], Wt.QuantU8) and not isinstance(sd['a.bias'], Wt.QuantU8)
    p = tmp_path / 'q.sdodw'
    Wt.save(str(p), sd)
    back = Wt.load(str(p))
    assert isinstance(back['a.weight'], Wt.QuantU8) and torch.equal(back['a.weight'].q, q.q)
    assert back['a.weight'].offset == q.offset and np.float32(back['a.weight'].scale) == np.float32(q.scale)
    assert torch.equal(back['a.bias'], sd['a.bias'])
    assert os.path.getsize(p) < w.numel() * 2                                          # 1 byte per weight on disk


def test_sd21_unet_parameter_table_matches_public_architecture():
    """SD v2.1 UNet (config 5): 64-wide heads, context 1024, Linear transformer projections -> 865,910,724 parameters"""
    from oracle import sd_torch as S
    from sdod.amd import engine as E
    cfg = E.sd21_config()
    assert (cfg.latent_h, cfg.context_dim, cfg.num_heads, cfg.head_dim, cfg.linear_proj) == (96, 1024, 0, 64, 1)
    table = dict(E.UNet(cfg, 2).param_table() + E.Temb(cfg, 1).param_table())
    with torch.device('meta'):
        ref = S.UNetModel(context_dim=1024, head_dim=64, use_linear=True)
    want = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    assert table == want
    assert sum(int(np.prod(s)) for s in table.values()) == 865_910_724


def test_sd21_text_tower_table_and_checkpoint_prefixes():
    """OpenCLIP ViT-H/14 text tower as SD2.x uses it: 23 of the checkpoint's 24 blocks + ln_final; 352,984,064 parameters in
    the 24-block tower (354,032,640 with the unused text projection), open_clip key names under cond_stage_model.model."""
    from oracle import sd_torch as S
    from sdod.amd import convert, engine as E
    cfg = E.sd21_config()
    assert (cfg.text_arch, cfg.text_layers, cfg.text_heads) == (1, 23, 16)
    table = dict(E.TextEncoder(cfg, 2).param_table())
    with torch.device('meta'):
        ref = S.OpenClipTextModel()
    want = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    assert sum(v.numel() for v in ref.state_dict().values()) == 352_984_064
    assert all(table[k] == want[k] for k in table) and len(table) == 280
    assert sorted(k for k in want if k not in table) == sorted(k for k in want if k.startswith('transformer.resblocks.23.'))
    # a v2 checkpoint is split with the open_clip prefix; the last block and the projection stay unused
    with torch.device('meta'):
        unet, vae = S.UNetModel(context_dim=1024, head_dim=64, use_linear=True), S.AutoencoderKLDecode()
    sd = {'model.diffusion_model.' + k: v for k, v in unet.state_dict().items()}
    sd.update({'first_stage_model.' + k: v for k, v in vae.state_dict().items()})
    sd.update({'cond_stage_model.model.' + k: v for k, v in ref.state_dict().items()})
    sd['cond_stage_model.model.text_projection'] = torch.empty(1024, 1024, device='meta')
    parts, unused = convert.split_state_dict(sd, convert.parameter_tables(cfg))
    assert len(parts['text']) == 280 and len(unused) == 13 and 'cond_stage_model.model.text_projection' in unused


def _container(entries):
    """raw .sdodw bytes from (name, dtype, dims, offset, nbytes) records; payload region = 4 KiB of zeros"""
    head = b'SDODW001' + struct.pack('<Q', len(entries))
    for name, dt, dims, off, nb in entries:
        nm = name.encode()
        head += struct.pack('<I', len(nm)) + nm + struct.pack('<II', dt, len(dims))
        head += b''.join(struct.pack('<Q', d & (2 ** 64 - 1)) for d in dims) + struct.pack('<QQ', off & (2 ** 64 - 1), nb & (2 ** 64 - 1))
    return head + bytes(4096)


@pytest.mark.parametrize('entry,what', [
    (('time_embed.0.bias', 1, [1280], 2 ** 64 - 64, 128), 'out of range'),          # offset + nbytes wraps around in u64
    (('time_embed.0.bias', 1, [1280], 512, 1 << 40), 'out of range'),               # payload runs past the end of the file
    (('time_embed.0.bias', 1, [1280], 512, 64), 'does not match its dims'),         # truncated payload: set_param would over-read
    (('time_embed.0.bias', 2, [1280], 512, 1280), 'does not match its dims'),       # affine uint8 without its {scale, offset} prefix
    (('time_embed.0.bias', 1, [0], 512, 0), 'bad dims'),
    (('time_embed.0.bias', 1, [-5], 512, 20), 'bad dims'),
    (('time_embed.0.bias', 7, [128], 512, 512), 'bad dtype'),
])
def test_malformed_weight_container_is_rejected_before_any_read(tmp_path, entry, what):
    """Graph::load_file validates every record against the file size and its own dims (a truncated / hostile
    container must yield LIBSDOD_INVALID_ARGUMENT, not an out-of-bounds read inside libsdod_setup).  No GPU needed: the
    record is rejected before device memory is claimed."""
    from sdod.amd import engine as E
    from sdod.amd._lib import SdodError
    path = tmp_path / 'temb.sdodw'
    path.write_bytes(_container([entry]))
    g = E.Temb(E.sd14_config(64, 64), 1)
    with pytest.raises(SdodError) as ei:
        g.load_file(str(path))
    assert ei.value.code == 2 and what in str(ei.value), str(ei.value)
