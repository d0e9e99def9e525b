"""Thin torch-tensor wrappers over the kernel-level C ABI (include/sdod_hip.h).

torch supplies device memory and the current stream only; every computation below is a call into
lib/libsdod.so.  Activations are NHWC fp16; a torch NCHW tensor in channels_last memory format
IS that layout, so the conversions are views."""
import ctypes

import torch

from . import _lib
from ._lib import GemmDesc, check

ACT = {None: 0, 'none': 0, 'silu': 1, 'gelu': 2, 'quick_gelu': 3}


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _req(t, dtype=None, name='tensor'):
    if not t.is_cuda:
        raise ValueError(f'{name} must live on the GPU')
    if not t.is_contiguous():
        raise ValueError(f'{name} must be contiguous')
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f'{name} must be {dtype}, got {t.dtype}')
    return t


_ws = {}


def workspace(nbytes, device, tag='default'):
    """Grow-only fp32 scratch per (device, tag); reused across calls on the same stream."""
    key = (device, tag)
    buf = _ws.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        # zeroed: the one-launch GroupNorm keeps its grid-barrier words at the end of its workspace (include/sdod_hip.h)
        buf = torch.zeros((max(nbytes, 1) + 3) // 4, dtype=torch.float32, device=device)
        _ws[key] = buf
    return buf


def gemm(a, w, bias=None, *, residual=None, row_bias=None, rows_per_img=0, act=None, alpha=1.0, out=None,
         conv=None, a2=None, bias_on_m=False, split_k=0, tile=0, time_iters=0, geglu=False, tail=None, bias2=None,
         ln_s=None, ln_eps=1e-5, cold_scratch=None, phase=0, return_desc=False, w_scale=None, w_off=None, fixup=False,
         xcd=0, softmax_cols=0):
    """out = act(alpha * A @ W^T + bias + row_bias) + residual.

    a: fp16 [M, K] (rows mode) or NHWC [N, H, W, C0] with conv=dict(stride=1|2, upsample=bool) (3x3 pad 1);
    a2: optional second NHWC source concatenated on channels; w: fp16 [Nout, K] (conv: K = 9*(C0+C1), KRSC)."""
    lib = _lib.hip()
    _req(a, torch.float16, 'a')
    d = GemmDesc()
    if w_scale is not None:   # affine-uint8 weight codes: real = (q + offset) * scale, w_off = offset + 128 per output column
        _req(w, torch.uint8, 'w'); _req(w_scale, torch.float32, 'w_scale'); _req(w_off, torch.float32, 'w_off')
        d.wq = 1; d.w_scale = _p(w_scale); d.w_off = _p(w_off)
    else:
        _req(w, torch.float16, 'w')
    per_image = w.dim() == 3   # [n_img, Nout, K]: the rows of image i (rows_per_img each) multiply w[i]; bias / ln_s may be [n_img, Nout]
    nout, k = w.shape[-2:]
    if per_image:
        assert conv is None and rows_per_img > 0 and w.is_contiguous()
        d.w_img_stride = nout * k
        d.rows_per_img = rows_per_img
        if (bias is not None and bias.dim() == 2) or (ln_s is not None and ln_s.dim() == 2):
            assert (bias is None or bias.dim() == 2) and (ln_s is None or ln_s.dim() == 2)
            d.vec_img_stride = nout
    d.softmax_cols = softmax_cols
    if conv is None:
        m = a.shape[0]
        assert a.shape[1] == k, (a.shape, w.shape)
        d.a_mode = 0
        d.lda = k
        out_shape = (m, nout)
    else:
        n_img, h, wd, c0 = a.shape
        c1 = 0
        if a2 is not None:
            _req(a2, torch.float16, 'a2')
            assert a2.shape[:3] == a.shape[:3]
            c1 = a2.shape[3]
        stride = int(conv.get('stride', 1)); ups = 1 if conv.get('upsample', False) else 0
        ks = int(conv.get('ksize', 3))
        hup, wup = h << ups, wd << ups
        ho, wo = (hup + 2 * (ks // 2) - ks) // stride + 1, (wup + 2 * (ks // 2) - ks) // stride + 1
        d.ksize = ks
        m = n_img * ho * wo
        d.a_mode = 1
        d.n_img, d.h_in, d.w_in, d.c0, d.c1 = n_img, h, wd, c0, c1
        d.stride, d.upsample = stride, ups
        d.a2 = _p(a2)
        out_shape = (n_img, ho, wo, nout)
    nvis = nout // 2 if geglu else nout
    if geglu:
        out_shape = out_shape[:-1] + (nvis,)
    if out is None:
        out = torch.empty(out_shape, dtype=torch.float16, device=a.device)
    else:
        _req(out, torch.float16, 'out')
        assert out.numel() == m * nvis
    d.a, d.w, d.out = _p(a), _p(w), _p(out)
    d.M, d.N, d.K = m, nout, k
    d.ldw, d.ldo = k, nvis
    d.geglu = 1 if geglu else 0
    if tail is not None:      # (t0, t1 or None): NHWC tensors read by the 1x1 tail segment; w holds [main K | tail K] columns
        t0, t1 = tail
        d.t0 = _p(t0); d.tc0 = t0.shape[-1]
        d.t1 = _p(t1); d.tc1 = t1.shape[-1] if t1 is not None else 0
        d.k_tail = k - d.tc0 - d.tc1
    if bias2 is not None:
        _req(bias2, torch.float32, 'bias2'); d.bias2 = _p(bias2)
    if ln_s is not None:      # LayerNorm folded into this Linear: w / bias / ln_s come from ln_fold()
        _req(ln_s, torch.float32, 'ln_s'); d.ln = 1; d.ln_s = _p(ln_s); d.ln_eps = ln_eps
    if bias is not None:
        _req(bias, torch.float32, 'bias'); d.bias = _p(bias)
    if row_bias is not None:
        _req(row_bias, torch.float16, 'row_bias'); d.row_bias = _p(row_bias); d.rows_per_img = rows_per_img
        d.ld_row_bias = row_bias.shape[-1]
    if residual is not None:
        _req(residual, torch.float16, 'residual'); assert residual.numel() == m * nout
        d.residual = _p(residual); d.ldr = nout
    d.act = ACT[act]
    d.alpha = alpha
    d.bias_on_m = 1 if bias_on_m else 0
    d.split_k = split_k
    d.tile = tile
    d.xcd_panels = xcd        # 0 = per-shape choice; 1 / 2 / 4 / 8 = n-tile panels of the XCD-aware tile order (speed only)
    need = lib.sdod_gemm_workspace_bytes(ctypes.byref(d))
    if need:
        ws = workspace(need, a.device, 'gemm')
        d.workspace = _p(ws); d.workspace_bytes = ws.numel() * 4
        if fixup:   # split-K reduced inside the GEMM launch where the tile supports it (zeroed counters, see include/sdod_hip.h)
            d.fix_counters = _p(workspace(lib.sdod_gemm_fixup_counters() * 4, a.device, 'gemm_fixup'))
    if time_iters:
        ms = ctypes.c_float()
        if cold_scratch is not None:   # cold weights, warm activations: what the launch meets inside a graph replay
            check(lib.sdod_gemm_time_cold(ctypes.byref(d), _stream(), min(time_iters, 16), _p(cold_scratch),
                                          cold_scratch.numel() * cold_scratch.element_size(), ctypes.byref(ms), None))
        else:
            check(lib.sdod_gemm_time(ctypes.byref(d), _stream(), time_iters, ctypes.byref(ms)))
        return ms.value
    d.phase = phase           # split-K only: 1 = partial slabs only (a GroupNorm may finish the job, group_norm_reduce)
    check(lib.sdod_gemm_f16(ctypes.byref(d), _stream()))
    if return_desc:
        d._keep = (a, w, bias, residual, row_bias, out, a2, bias2)   # the descriptor holds raw pointers
        return out, d
    return out


class GnReduce(ctypes.Structure):
    """mirror of `struct sdod_gn_reduce`"""
    _fields_ = [('partial', ctypes.c_void_p), ('splits', ctypes.c_int), ('slab_floats', ctypes.c_size_t), ('bias', ctypes.c_void_p),
                ('bias2', ctypes.c_void_p), ('row_bias', ctypes.c_void_p), ('ld_row_bias', ctypes.c_int), ('residual', ctypes.c_void_p),
                ('ldr', ctypes.c_int), ('x_out', ctypes.c_void_p), ('alpha', ctypes.c_float), ('act', ctypes.c_int),
                ('M', ctypes.c_int), ('N', ctypes.c_int)]


def group_norm_reduce(desc, n, hw, groups, weight=None, bias=None, eps=1e-5, silu=False, x2=None):
    """GroupNorm(+SiLU) of the output of a split-K GEMM launched with phase=1 (desc from gemm(..., return_desc=True)): the
    reduce + epilogue of that GEMM and the normalisation in one launch.  Returns y; x lands in the GEMM's `out`."""
    lib = _lib.hip()
    red = GnReduce()
    check(lib.sdod_gemm_reduce_info(ctypes.byref(desc), ctypes.byref(red)))
    c0 = desc.N
    c1 = x2.shape[-1] if x2 is not None else 0
    dev = desc._keep[0].device
    y = torch.empty((n, hw, c0 + c1), dtype=torch.float16, device=dev)
    if weight is not None:
        weight = weight.detach().to(torch.float32).contiguous(); bias = bias.detach().to(torch.float32).contiguous()
    check(lib.sdod_group_norm_reduce_nhwc(ctypes.byref(red), _p(x2), _p(y), _p(weight), _p(bias), n, hw, c0, c1, groups, eps,
                                          1 if silu else 0, _stream()))
    return y


def group_norm_nhwc(x, groups, weight=None, bias=None, eps=1e-5, silu=False, x2=None, out=None):
    """x: [N, ..., C] channels-last fp16/fp32 (optionally concatenated with x2 on C)."""
    lib = _lib.hip()
    _req(x, None, 'x')
    assert x.dtype in (torch.float16, torch.float32)
    n, c0 = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c0)
    c1 = 0
    if x2 is not None:
        _req(x2, x.dtype, 'x2'); c1 = x2.shape[-1]
    if out is None:
        out = torch.empty(x.shape[:-1] + (c0 + c1,), dtype=x.dtype, device=x.device)
    if weight is not None:
        weight = weight.detach().to(torch.float32).contiguous(); bias = bias.detach().to(torch.float32).contiguous()
    ws = workspace(lib.sdod_group_norm_workspace_bytes(n, groups), x.device, 'gn')
    check(lib.sdod_group_norm_nhwc(_p(x), _p(x2), _p(out), _p(weight), _p(bias), n, hw, c0, c1, groups, eps,
                                   1 if silu else 0, 0 if x.dtype == torch.float16 else 1, _p(ws), _stream()))
    return out


_NCHW_DTYPES = {torch.float16: 0, torch.float32: 1, torch.bfloat16: 3}


def group_norm_nchw(x, groups, weight=None, bias=None, eps=1e-5, silu=False):
    """torch-semantics entry used by sdod.EfficientGN: x is [N, C, *] fp16 / bf16 / fp32; returns a tensor of the same shape
    and memory format.  No layout copy on either side: a channels_last tensor whose channel count the NHWC kernels take IS
    their layout (the permute is a view); anything else that is dense goes to the NCHW kernel, where a group is one contiguous
    slab (any channel count); only a tensor that is neither is made contiguous first."""
    if x.dtype not in _NCHW_DTYPES:
        raise TypeError('EfficientGN HIP kernels support float16, bfloat16 and float32')
    n, c = x.shape[0], x.shape[1]
    x = x.detach()
    if (x.dim() == 4 and c % 8 == 0 and x.dtype != torch.bfloat16 and not x.is_contiguous()
            and x.is_contiguous(memory_format=torch.channels_last)):
        y = group_norm_nhwc(x.permute(0, 2, 3, 1), groups, weight, bias, eps, silu)     # [N, H, W, C] view -> NHWC kernels
        return y.permute(0, 3, 1, 2)                                                      # logical NCHW, channels_last strides
    lib = _lib.hip()
    if not x.is_contiguous():
        x = x.contiguous()
    _req(x, None, 'x')
    out = torch.empty_like(x)
    if weight is not None:
        weight = weight.detach().to(torch.float32).contiguous(); bias = bias.detach().to(torch.float32).contiguous()
    spatial = x.numel() // (n * c)
    ws = workspace(lib.sdod_group_norm_nchw_workspace_bytes(n, groups), x.device, 'gn_nchw')
    check(lib.sdod_group_norm_nchw(_p(x), _p(out), _p(weight), _p(bias), n, c, spatial, groups, eps, 1 if silu else 0,
                                   _NCHW_DTYPES[x.dtype], _p(ws), _stream()))
    return out


def ln_fold(w, gamma, beta, bias=None):
    """fold LayerNorm(gamma, beta) into Linear(w, bias): returns (w_folded fp16, s fp32, t fp32); w is modified in place"""
    lib = _lib.hip()
    _req(w, torch.float16, 'w'); _req(gamma, torch.float32, 'gamma'); _req(beta, torch.float32, 'beta')
    n, k = w.shape
    s = torch.empty(n, dtype=torch.float32, device=w.device); t = torch.empty_like(s)
    check(lib.sdod_ln_fold_f16(_p(w), n, k, k, _p(gamma), _p(beta), _p(bias), _p(s), _p(t), _stream()))
    return w, s, t


def layer_norm(x, weight, bias, eps=1e-5, out=None):
    lib = _lib.hip()
    _req(x, torch.float16, 'x')
    c = x.shape[-1]; m = x.numel() // c
    if out is None:
        out = torch.empty_like(x)
    check(lib.sdod_layer_norm_f16(_p(x), _p(out), _p(weight), _p(bias), m, c, eps, _stream()))
    return out


def attention(q, k, v, heads, scale=None, causal=False, out=None):
    """q: [B, Lq, heads*d], k/v: [B, Lk, heads*d] fp16 (row strides = last-dim size)."""
    lib = _lib.hip()
    for t, nme in ((q, 'q'), (k, 'k'), (v, 'v')):
        _req(t, torch.float16, nme)
    b, lq, c = q.shape
    lk = k.shape[1]
    d = c // heads
    if scale is None:
        scale = d ** -0.5
    if out is None:
        out = torch.empty_like(q)
    check(lib.sdod_attention_f16(_p(q), _p(k), _p(v), _p(out), b, heads, lq, lk, d, q.shape[2], k.shape[2], v.shape[2],
                                 out.shape[2], scale, 1 if causal else 0, _stream()))
    return out


def xattn_fold(kv, k_off, v_off, n_img, L, wq, sq, tq, wo, heads, scale=None):
    """once-per-prompt part of the folded cross-attention (include/sdod_hip.h: sdod_xattn_fold_f16).  kv: fp16 [n_img * L, ld]
    holding K at column k_off and V at v_off; wq: LayerNorm-folded to_q weight with its fold vectors sq, tq (ln_fold); wo: to_out
    weight.  Returns (w1 [n_img, heads*80, C], s1, t1 [n_img, heads*80], w2 [n_img, C, heads*80])."""
    lib = _lib.hip()
    _req(kv, torch.float16, 'kv'); _req(wq, torch.float16, 'wq'); _req(wo, torch.float16, 'wo')
    _req(sq, torch.float32, 'sq'); _req(tq, torch.float32, 'tq')
    c = wq.shape[0]; d = c // heads
    if scale is None:
        scale = d ** -0.5
    w1 = torch.empty(n_img, heads * 80, c, dtype=torch.float16, device=kv.device)
    w2 = torch.empty(n_img, c, heads * 80, dtype=torch.float16, device=kv.device)
    s1 = torch.empty(n_img, heads * 80, dtype=torch.float32, device=kv.device); t1 = torch.empty_like(s1)
    check(lib.sdod_xattn_fold_f16(_p(kv), kv.shape[-1], k_off, v_off, n_img, L, _p(wq), wq.shape[1], _p(sq), _p(tq), _p(wo), wo.shape[1],
                                  heads, d, scale, _p(w1), _p(s1), _p(t1), _p(w2), _stream()))
    return w1, s1, t1, w2


def attention_strided(q, k, v, out, batch, heads, lq, lk, d, ldq, ldk, ldv, ldo, scale, causal=False):
    """raw form for packed qkv buffers: pointers may be column offsets into wider rows"""
    lib = _lib.hip()
    check(lib.sdod_attention_f16(q, k, v, out, batch, heads, lq, lk, d, ldq, ldk, ldv, ldo, scale, 1 if causal else 0,
                                 _stream()))


def softmax_rows(x, out=None):
    lib = _lib.hip()
    _req(x, torch.float16, 'x')
    n = x.shape[-1]; m = x.numel() // n
    if out is None:
        out = torch.empty_like(x)
    check(lib.sdod_softmax_rows_f16(_p(x), _p(out), m, n, _stream()))
    return out


def geglu(x, out=None):
    lib = _lib.hip()
    _req(x, torch.float16, 'x')
    c = x.shape[-1] // 2; m = x.numel() // (2 * c)
    if out is None:
        out = torch.empty(x.shape[:-1] + (c,), dtype=torch.float16, device=x.device)
    check(lib.sdod_geglu_f16(_p(x), _p(out), m, c, _stream()))
    return out


def activation(x, act, out=None):
    lib = _lib.hip()
    _req(x, torch.float16, 'x')
    if out is None:
        out = torch.empty_like(x)
    check(lib.sdod_act_f16(_p(x), _p(out), x.numel(), ACT[act], _stream()))
    return out


def add(a, b, out=None):
    lib = _lib.hip()
    _req(a, torch.float16, 'a'); _req(b, torch.float16, 'b')
    if out is None:
        out = torch.empty_like(a)
    check(lib.sdod_add_f16(_p(a), _p(b), _p(out), a.numel(), _stream()))
    return out


def concat_channels(a, b):
    lib = _lib.hip()
    _req(a, torch.float16, 'a'); _req(b, torch.float16, 'b')
    c0, c1 = a.shape[-1], b.shape[-1]
    rows = a.numel() // c0
    out = torch.empty(a.shape[:-1] + (c0 + c1,), dtype=torch.float16, device=a.device)
    check(lib.sdod_concat_channels_f16(_p(a), _p(b), _p(out), rows, c0, c1, _stream()))
    return out


def im2col3x3_small(x, kpad=64):
    lib = _lib.hip()
    _req(x, torch.float16, 'x')
    n, h, w, c = x.shape
    out = torch.empty((n * h * w, kpad), dtype=torch.float16, device=x.device)
    check(lib.sdod_im2col3x3_small_f16(_p(x), _p(out), n, h, w, c, kpad, _stream()))
    return out


def latent_im2col(x, kpad=64, scale=1.0):
    """NCHW fp32 latent -> im2col matrix [n*h*w, kpad] fp16 of a 3x3 pad-1 convolution (the UNet's input conv as a K = 64 GEMM)"""
    lib = _lib.hip()
    _req(x, torch.float32, 'x')
    n, c, h, w = x.shape
    out = torch.empty((n * h * w, kpad), dtype=torch.float16, device=x.device)
    check(lib.sdod_latent_im2col_f16(_p(x), _p(out), n, h, w, c, kpad, scale, _stream()))
    return out


def conv_in(x, w, bias, scale=1.0):
    """the UNet's input convolution in one launch (include/sdod_hip.h: sdod_conv_in_f16): x NCHW fp32 [n, c, h, w], w fp16
    [cout, 64] (k = tap * c + channel, zero beyond 9 c), bias fp32 [cout] -> NHWC fp16 [n, h, w, cout]"""
    lib = _lib.hip()
    _req(x, torch.float32, 'x'); _req(w, torch.float16, 'w'); _req(bias, torch.float32, 'bias')
    n, c, h, wd = x.shape
    cout = w.shape[0]
    assert w.shape[1] == 64
    out = torch.empty((n, h, wd, cout), dtype=torch.float16, device=x.device)
    check(lib.sdod_conv_in_f16(_p(x), _p(w), _p(bias), _p(out), n, h, wd, c, cout, scale, _stream()))
    return out


def nchw_f32_to_nhwc_f16(x, scale=1.0):
    lib = _lib.hip()
    _req(x, torch.float32, 'x')
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // (n * c)
    out = torch.empty((n,) + tuple(x.shape[2:]) + (c,), dtype=torch.float16, device=x.device)
    check(lib.sdod_nchw_f32_to_nhwc_f16(_p(x), _p(out), n, c, hw, scale, _stream()))
    return out


def nhwc_f16_to_nchw_f32(x):
    lib = _lib.hip()
    _req(x, torch.float16, 'x')
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    out = torch.empty((n, c) + tuple(x.shape[1:-1]), dtype=torch.float32, device=x.device)
    check(lib.sdod_nhwc_f16_to_nchw_f32(_p(x), _p(out), n, c, hw, _stream()))
    return out


def embedding(ids, table, pos):
    lib = _lib.hip()
    _req(ids, torch.int32, 'ids'); _req(table, torch.float16, 'table'); _req(pos, torch.float16, 'pos')
    rows = ids.numel(); seq = pos.shape[0]; c = table.shape[1]
    out = torch.empty(tuple(ids.shape) + (c,), dtype=torch.float16, device=ids.device)
    check(lib.sdod_embedding_f16(_p(ids), _p(table), _p(pos), _p(out), rows, seq, c, _stream()))
    return out


def timestep_features(t, dim=320):
    lib = _lib.hip()
    _req(t, torch.float32, 't')
    out = torch.empty((t.numel(), dim), dtype=torch.float16, device=t.device)
    check(lib.sdod_timestep_features_f16(_p(t), _p(out), t.numel(), dim, _stream()))
    return out


def cfg_combine(eps_nhwc, guidance, uncond_first=True, mode=1):
    lib = _lib.hip()
    _req(eps_nhwc, torch.float16, 'eps')
    n2, c = eps_nhwc.shape[0], eps_nhwc.shape[-1]
    n = n2 // 2
    spatial = tuple(eps_nhwc.shape[1:-1])
    hw = eps_nhwc.numel() // (n2 * c)
    out = torch.empty((n, c) + spatial, dtype=torch.float32, device=eps_nhwc.device)
    check(lib.sdod_cfg_combine(_p(eps_nhwc), _p(out), n, c, hw, guidance, 1 if uncond_first else 0, mode, _stream()))
    return out


def plms_update(eps_nhwc, x, old, coefs, div, ddim, guidance, mode=1, v_coef=None, stage=None):
    """one PLMS step in one launch (include/sdod_hip.h: sdod_plms_update): CFG of eps (+ v -> eps with v_coef = (c_e, c_x)),
    e' = (coefs[0] e_t + coefs[1:] . old) / div, DDIM update of x in place with ddim = schedule.coef(index), and -- with
    stage = (x_dst, temb_row, temb_dst) -- the next evaluation's inputs.  Returns e_t (fp32 [n, c, ...])."""
    lib = _lib.hip()
    _req(eps_nhwc, torch.float16, 'eps'); _req(x, torch.float32, 'x')
    n2, c = eps_nhwc.shape[0], eps_nhwc.shape[-1]
    n = n2 // 2
    hw = eps_nhwc.numel() // (n2 * c)
    assert x.numel() == n * c * hw and len(old) <= 3 and len(coefs) == len(old) + 1
    e_out = torch.empty_like(x)
    a = _lib.PlmsUpdateArgs()
    a.eps_nhwc, a.e_out, a.x = _p(eps_nhwc), _p(e_out), _p(x)
    olds = list(old) + [None] * (3 - len(old)); cs = list(coefs) + [0.0] * (4 - len(coefs))
    for t in old:
        _req(t, torch.float32, 'old')
    a.old1, a.old2, a.old3 = _p(olds[0]), _p(olds[1]), _p(olds[2])
    a.n, a.c, a.hw, a.uncond_first, a.mode = n, c, hw, 1, mode
    a.guidance, a.c0, a.c1, a.c2, a.c3, a.div = guidance, cs[0], cs[1], cs[2], cs[3], div
    if v_coef is not None:
        a.v_pred, a.vc0, a.vc1 = 1, v_coef[0], v_coef[1]
    a.sqrt_one_minus_at, a.sqrt_at, a.sqrt_a_prev, a.dir_coef = ddim['sqrt_one_minus_at'], ddim['sqrt_at'], ddim['sqrt_a_prev'], ddim['dir_coef']
    if stage is not None:
        x_dst, temb_row, temb_dst = stage
        _req(x_dst, torch.float32, 'x_dst'); _req(temb_row, torch.float16, 'temb_row'); _req(temb_dst, torch.float16, 'temb_dst')
        a.x_stage, a.stage_reps = _p(x_dst), x_dst.numel() // x.numel()
        assert a.stage_reps * x.numel() == x_dst.numel() and temb_dst.numel() % temb_row.numel() == 0
        a.temb_row, a.temb_dst, a.temb_width, a.temb_reps = _p(temb_row), _p(temb_dst), temb_row.numel(), temb_dst.numel() // temb_row.numel()
    check(lib.sdod_plms_update(ctypes.byref(a), _stream()))
    return e_out


def dpm_step(eps_nhwc, x, y_prev, coef, guidance, mode=0, uncond_first=True, stage=None):
    """the reference driver's per-step arithmetic in one launch (include/sdod_hip.h: sdod_dpm_step): CFG of eps, the
    DPM-Solver++(2M) update of x / y_prev in place with coef = DpmSolver.coef(step), and -- with stage = (x_dst, temb_row,
    temb_dst) -- the next evaluation's inputs"""
    lib = _lib.hip()
    _req(eps_nhwc, torch.float16, 'eps'); _req(x, torch.float32, 'x'); _req(y_prev, torch.float32, 'y_prev')
    n2, c = eps_nhwc.shape[0], eps_nhwc.shape[-1]
    n = n2 // 2
    hw = eps_nhwc.numel() // (n2 * c)
    assert x.numel() == n * c * hw == y_prev.numel()
    a = _lib.DpmStepArgs()
    a.eps_nhwc, a.x, a.y_prev = _p(eps_nhwc), _p(x), _p(y_prev)
    a.n, a.c, a.hw, a.uncond_first, a.mode, a.order = n, c, hw, 1 if uncond_first else 0, mode, coef['order']
    a.guidance = guidance
    a.sigma_s, a.alpha_s, a.sigma_ratio, a.c_prev, a.c_cur = coef['sigma_s'], coef['alpha_s'], coef['sigma_ratio'], coef['c_prev'], coef['c_cur']
    if stage is not None:
        x_dst, temb_row, temb_dst = stage
        _req(x_dst, torch.float32, 'x_dst'); _req(temb_row, torch.float16, 'temb_row'); _req(temb_dst, torch.float16, 'temb_dst')
        a.x_stage, a.stage_reps = _p(x_dst), x_dst.numel() // x.numel()
        a.temb_row, a.temb_dst, a.temb_width, a.temb_reps = _p(temb_row), _p(temb_dst), temb_row.numel(), temb_dst.numel() // temb_row.numel()
    check(lib.sdod_dpm_step(ctypes.byref(a), _stream()))


def stage_unet_inputs(x, x_dst, temb_row, temb_dst):
    """x fp32 [n,...] -> x_dst [reps*n,...] (repeated back to back); temb_row fp16 [w] -> every row of temb_dst [b, w]"""
    lib = _lib.hip()
    _req(x, torch.float32, 'x'); _req(x_dst, torch.float32, 'x_dst'); _req(temb_row, torch.float16, 'temb_row'); _req(temb_dst, torch.float16, 'temb_dst')
    reps = x_dst.numel() // x.numel()
    assert reps * x.numel() == x_dst.numel() and temb_dst.numel() % temb_row.numel() == 0
    check(lib.sdod_stage_unet_inputs(_p(x), _p(x_dst), x.numel(), reps, _p(temb_row), _p(temb_dst), temb_row.numel(),
                                     temb_dst.numel() // temb_row.numel(), _stream()))


def randn(shape, seed, stream_id, device, return_words=False):
    """N(0,1) fp32 tensor from the in-tree Philox4x32-10 generator: a pure function of (seed, stream_id, element index)"""
    lib = _lib.hip()
    out = torch.empty(shape, dtype=torch.float32, device=device)
    words = torch.empty(out.numel(), dtype=torch.int32, device=device) if return_words else None
    with torch.cuda.device(out.device):
        check(lib.sdod_randn_f32(_p(out), _p(words), out.numel(), int(seed) & (2 ** 64 - 1), int(stream_id) & (2 ** 64 - 1), _stream()))
    return (out, words) if return_words else out


def dpm_update(x, eps, y_prev, order, sigma_s, alpha_s, sigma_ratio, c_prev, c_cur):
    lib = _lib.hip()
    for t in (x, eps, y_prev):
        _req(t, torch.float32)
    check(lib.sdod_dpm_update(_p(x), _p(eps), _p(y_prev), x.numel(), order, sigma_s, alpha_s, sigma_ratio, c_prev, c_cur,
                              _stream()))


def ddim_step(x, e, sqrt_one_minus_at, sqrt_at, sqrt_a_prev, dir_coef):
    lib = _lib.hip()
    _req(x, torch.float32); _req(e, torch.float32)
    check(lib.sdod_ddim_step_f32(_p(x), _p(e), x.numel(), sqrt_one_minus_at, sqrt_at, sqrt_a_prev, dir_coef, _stream()))


def lincomb4(es, coefs, div):
    lib = _lib.hip()
    es = list(es) + [None] * (4 - len(es)); coefs = list(coefs) + [0.0] * (4 - len(coefs))
    out = torch.empty_like(es[0])
    check(lib.sdod_lincomb4_f32(_p(out), _p(es[0]), _p(es[1]), _p(es[2]), _p(es[3]), coefs[0], coefs[1], coefs[2], coefs[3],
                                div, out.numel(), _stream()))
    return out


def image_to_u8(img_nhwc, a=0.5, b=0.5, mode=1):
    lib = _lib.hip()
    _req(img_nhwc, torch.float16, 'img')
    out = torch.empty(img_nhwc.shape, dtype=torch.uint8, device=img_nhwc.device)
    check(lib.sdod_image_to_u8(_p(img_nhwc), _p(out), img_nhwc.numel(), a, b, mode, _stream()))
    return out


def device_info():
    lib = _lib.hip()
    cu = ctypes.c_int(); mem = ctypes.c_size_t(); arch = ctypes.create_string_buffer(64)
    check(lib.sdod_hip_device_info(ctypes.byref(cu), ctypes.byref(mem), arch, 64))
    return {'cu_count': cu.value, 'hbm_bytes': mem.value, 'arch': arch.value.decode()}
