"""CPU tests (no GPU): the oracle's host arithmetic against the golden vectors generated from the REFERENCE itself
(tests/golden/*, made by oracle/gen_golden.py from /root/reference's dpm_solver.cpp / tokenizer.cpp / sdod.EfficientGN),
plus hand-derivable known answers for the fragments of context.cpp / qnn_context.cpp that cannot be compiled here."""
import json
import math
import os

import numpy as np
import pytest

TABLES = {'ts': 0, 'log_alphas': 1, 'lambdas': 2, 'sigmas': 3, 'alphas': 4, 'phis': 5, 'i2rs': 6, 'model_ts': 7,
          'all_t': 8, 'all_log_alpha': 9}


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize('steps', [20, 50, 1, 2, 3, 8, 100])
def test_oracle_dpm_tables_and_trajectory_bit_exact(oracle_lib, golden_dir, steps):
    g = json.load(open(os.path.join(golden_dir, f'dpm_steps{steps}.json')))
    h = oracle_lib.oracle_dpm_create(1000, 0.00085, 0.0120)
    oracle_lib.oracle_dpm_prepare(h, steps)
    for name, w in TABLES.items():
        n = 1000 if w >= 8 else steps + 1
        out = np.zeros(n, np.float32)
        assert oracle_lib.oracle_dpm_table(h, w, out.ctypes.data) == n
        assert np.array_equal(bits(out), np.array(g[name + '_bits'], np.uint32)), name
    x = np.array(g['x0_bits'], np.uint32).view(np.float32).copy()
    for s, rec in enumerate(g['trajectory']):
        e = np.array(rec['eps_bits'], np.uint32).view(np.float32).copy()
        oracle_lib.oracle_dpm_update(h, s, x.ctypes.data, e.ctypes.data, x.size)
        assert np.array_equal(bits(x), np.array(rec['x_bits'], np.uint32)), f'x step {s}'
        assert np.array_equal(bits(e), np.array(rec['y_bits'], np.uint32)), f'y step {s}'
    oracle_lib.oracle_dpm_destroy(h)


def test_dpm_known_answers_from_survey(golden_dir):
    """values quoted in SURVEY.md 8a row S (printed by the reference's test_dpm harness)"""
    g = json.load(open(os.path.join(golden_dir, 'dpm_steps20.json')))
    f = lambda k: np.array(g[k], np.uint32).view(np.float32)
    mts = f('model_ts_bits')
    assert mts[0] == 999.0 and abs(mts[1] - 949.049988) < 1e-4 and abs(mts[19] - 49.949936) < 1e-4 and abs(mts[20] + 0.000066) < 1e-5
    assert abs(f('sigmas_bits')[0] - 0.997668) < 1e-6 and abs(f('alphas_bits')[0] - 0.068260) < 1e-6
    assert abs(f('lambdas_bits')[0] + 2.682101) < 1e-5 and abs(f('phis_bits')[1] + 0.253662) < 1e-5
    assert abs(f('i2rs_bits')[2] - 0.465400) < 1e-5 and abs(f('sigmas_bits')[20] - 0.029152) < 1e-6


def test_oracle_tokenizer_matches_reference_golden(golden_dir):
    from oracle.tokenizer_oracle import TokenizerOracle
    g = json.load(open(os.path.join(golden_dir, 'tokenizer_synthetic.json')))
    tok = TokenizerOracle(os.path.join(golden_dir, g['vocab']), canonical_ws=False)
    assert len(g['cases']) > 400
    for c in g['cases']:
        assert tok.tokenize(c['text'], g['context_len']) == c['ids'], c['text']
    assert tok.tokenize('abc')[:3] == [554, 544, 555]          # SOT, merged "abc</w>", EOT with the synthetic vocab


def test_oracle_tokenizer_q3_case_terminates_canonically(golden_dir):
    """reference bug Q3: [a, a, b, c</w>] with merge (a, b) hangs there; canonical CLIP gives [a, ab, c</w>] -> then abc"""
    from oracle.tokenizer_oracle import TokenizerOracle
    tok = TokenizerOracle(os.path.join(golden_dir, 'ctokenizer_synthetic.txt'))
    assert tok.diverges_from_reference('aabc')
    ids = tok.tokenize('aabc')
    assert ids[0] == tok.start_token and ids[-1] == tok.end_token and len(ids) == 77
    assert ids[1] == tok.tokens['a'] and ids[2] == tok.tokens['abc</w>']


def test_timestep_features_known_answers(oracle_lib):
    out = np.zeros(320, np.float32)
    oracle_lib.oracle_timestep_features(999.0, 320, out.ctypes.data)
    for j in (0, 1, 7, 159):
        f = math.exp(-math.log(10000.0) * j / 160)
        assert abs(out[j] - math.cos(999.0 * f)) < 2e-4 and abs(out[160 + j] - math.sin(999.0 * f)) < 2e-4
    oracle_lib.oracle_timestep_features(0.0, 320, out.ctypes.data)
    assert np.all(out[:160] == 1.0) and np.all(out[160:] == 0.0)


def test_cfg_dequant_uint8_known_answers(oracle_lib):
    ec = np.array([1.0, -2.0, 0.5], np.float32); eu = np.array([0.0, 1.0, 0.5], np.float32)
    e = np.zeros(3, np.float32)
    oracle_lib.oracle_cfg_combine(e.ctypes.data, ec.ctypes.data, eu.ctypes.data, 7.5, 3)
    assert np.allclose(e, eu + 7.5 * (ec - eu), atol=1e-6)
    oracle_lib.oracle_cfg_combine(e.ctypes.data, ec.ctypes.data, eu.ctypes.data, 1.0, 3)
    assert np.array_equal(e, ec)                                    # context.cpp:359-360: g == 1 skips the uncond term
    q = np.array([0, 128, 255], np.uint8); out = np.zeros(3, np.float32)
    oracle_lib.oracle_dequant_u8(out.ctypes.data, q.ctypes.data, -128, 0.5, 3, 0, 0, 0.0)
    assert np.array_equal(out, np.array([-64.0, 0.0, 63.5], np.float32))   # real = (q + offset) * scale
    oracle_lib.oracle_dequant_u8(out.ctypes.data, q.ctypes.data, -128, 0.5, 3, 1, 1, 2.0)
    assert np.array_equal(out, np.array([-192.0, 0.0, 190.5], np.float32))  # += 2 * real
    img = np.array([-0.5, 0.0, 0.5, 0.999, 1.0, 1.5, 0.00392], np.float32); u8 = np.zeros(7, np.uint8)
    oracle_lib.oracle_to_uint8(u8.ctypes.data, img.ctypes.data, 7)
    assert u8.tolist() == [0, 0, 127, 254, 255, 255, 0]             # truncating cast after the clamp


def test_oracle_model_structure_known_answers():
    """parameter totals of the public SD v1.4 graphs -- structural known-answer test of the oracle"""
    import torch
    from oracle import sd_torch as S
    with torch.device('meta'):
        assert S.count_params(S.UNetModel()) == 859_520_964
        assert S.count_params(S.AutoencoderKLDecode()) == 49_490_199
        assert S.count_params(S.ClipTextModel()) == 123_060_480


def test_oracle_clip_agrees_with_transformers_implementation():
    import torch
    from oracle import sd_torch as S
    transformers = pytest.importorskip('transformers')
    small = dict(vocab=1000, d=128, layers=2, heads=2, inter=512, max_pos=77)
    c = S.build(S.ClipTextModel, seed=5, **small)
    cfg = transformers.CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=512, num_hidden_layers=2,
                                      num_attention_heads=2, max_position_embeddings=77, hidden_act='quick_gelu')
    hf = transformers.CLIPTextModel(cfg).eval()
    sd = c.state_dict()
    if not any(k.startswith('text_model.') for k in hf.state_dict()):
        sd = {k[len('text_model.'):]: v for k, v in sd.items()}
    res = hf.load_state_dict(sd, strict=False)
    assert not res.missing_keys and not res.unexpected_keys
    ids = torch.randint(0, 1000, (2, 77), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        assert float((c(ids) - hf(input_ids=ids).last_hidden_state).abs().max()) < 1e-4


def test_oracle_openclip_tower_agrees_with_transformers_implementation():
    """config 5's conditioning oracle (oracle/sd_torch.py: OpenClipTextModel -- open_clip key names, fused in_proj, erf GELU,
    penultimate block + ln_final) against an independent implementation of the same architecture: transformers.CLIPTextModel
    with hidden_act='gelu', weights mapped key by key; the penultimate output is HF's hidden_states[-2] through its final
    LayerNorm (until now only the CLIP-L oracle had such a cross-check)"""
    import torch
    from oracle import sd_torch as S
    transformers = pytest.importorskip('transformers')
    d, layers, heads = 128, 3, 4
    c = S.build(S.OpenClipTextModel, seed=6, vocab=1000, d=d, layers=layers, heads=heads, max_pos=77, run_layers=layers - 1)
    cfg = transformers.CLIPTextConfig(vocab_size=1000, hidden_size=d, intermediate_size=4 * d, num_hidden_layers=layers,
                                      num_attention_heads=heads, max_position_embeddings=77, hidden_act='gelu')
    hf = transformers.CLIPTextModel(cfg).eval()
    osd = c.state_dict()
    m = {'embeddings.token_embedding.weight': osd['token_embedding.weight'],
         'embeddings.position_embedding.weight': osd['positional_embedding'],
         'final_layer_norm.weight': osd['ln_final.weight'], 'final_layer_norm.bias': osd['ln_final.bias']}
    for i in range(layers):
        o, h = f'transformer.resblocks.{i}.', f'encoder.layers.{i}.'
        wq, wk, wv = osd[o + 'attn.in_proj_weight'].chunk(3, 0)
        bq, bk, bv = osd[o + 'attn.in_proj_bias'].chunk(3, 0)
        for n, w, b in (('q_proj', wq, bq), ('k_proj', wk, bk), ('v_proj', wv, bv)):
            m[h + f'self_attn.{n}.weight'] = w; m[h + f'self_attn.{n}.bias'] = b
        m[h + 'self_attn.out_proj.weight'] = osd[o + 'attn.out_proj.weight']; m[h + 'self_attn.out_proj.bias'] = osd[o + 'attn.out_proj.bias']
        for a, b in (('ln_1', 'layer_norm1'), ('ln_2', 'layer_norm2'), ('mlp.c_fc', 'mlp.fc1'), ('mlp.c_proj', 'mlp.fc2')):
            m[h + b + '.weight'] = osd[o + a + '.weight']; m[h + b + '.bias'] = osd[o + a + '.bias']
    if any(k.startswith('text_model.') for k in hf.state_dict()):
        m = {'text_model.' + k: v for k, v in m.items()}
    res = hf.load_state_dict(m, strict=False)
    assert not res.unexpected_keys and all('position_ids' in k for k in res.missing_keys), res
    ids = torch.randint(1, 999, (2, 77), generator=torch.Generator().manual_seed(2))
    ids[:, 0] = 998; ids[0, 9] = 999; ids[0, 10:] = 0; ids[1, 40] = 999; ids[1, 41:] = 0     # open_clip pads with 0 after EOT
    with torch.no_grad():
        out = hf(input_ids=ids, output_hidden_states=True)
        tm = hf.text_model if hasattr(hf, 'text_model') else hf
        want = tm.final_layer_norm(out.hidden_states[-2])
        got = c(ids)
    assert got.shape == want.shape == (2, 77, d)
    assert float((got - want).abs().max()) < 1e-4
    # and with every block run (run_layers = layers) it is HF's last_hidden_state
    c.run_layers = layers
    with torch.no_grad():
        assert float((c(ids) - out.last_hidden_state).abs().max()) < 1e-4


def test_philox_oracle_matches_published_known_answer_vectors():
    """oracle/philox_oracle.py (the checker of sdod_randn_f32) against the Philox4x32-10 known-answer vectors distributed
    with the algorithm's reference implementation (Random123 `kat_vectors`: zero, all-ones and the pi-digits inputs)."""
    import numpy as np
    from oracle.philox_oracle import philox4x32_10, randn
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kat:
        got = philox4x32_10(np.array([ctr], np.uint32), np.array([key], np.uint32))[0]
        assert [int(v) for v in got] == want
    w, z = randn(10, 0, 0)                       # counter (0, 0): the first block is the zero-input vector above
    assert [int(v) for v in w[0]] == kat[0][2] and z.shape == (10,) and np.isfinite(z).all()
    _, big = randn(1 << 18, 7, 1)
    assert abs(big.mean()) < 1e-2 and abs(big.var() - 1.0) < 1e-2
