"""ctypes wrapper of the graph-level C ABI (include/sdod_engine.h): the MI355X stand-in for the reference's
QnnGraph objects (context.cpp:105 loads unet / text_encoder / vae_decoder / temb; :201-221 wires their I/O)."""
import contextlib
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check

UNET, VAE_DECODER, TEXT_ENCODER, TEMB = 0, 1, 2, 3


class ModelConfig(ctypes.Structure):
    """mirror of `struct sdod_model_config`"""
    _fields_ = [(n, ctypes.c_int) for n in (
        'latent_channels', 'latent_h', 'latent_w', 'model_channels', 'context_dim', 'context_len', 'num_heads',
        'head_dim', 'vocab_size', 'text_layers', 'text_heads', 'vae_channels', 'linear_proj', 'text_arch', 'weight_quant')]


ENGINE_SYMBOLS = [
    'sdod_model_config_sd14', 'sdod_model_config_sd21', 'sdod_graph_create', 'sdod_graph_destroy', 'sdod_graph_num_params', 'sdod_graph_param_info',
    'sdod_graph_set_param', 'sdod_graph_load_file', 'sdod_graph_finalize', 'sdod_graph_io', 'sdod_graph_execute', 'sdod_graph_check',
    'sdod_graph_stats', 'sdod_graph_tune_info', 'sdod_graph_num_ops', 'sdod_graph_op_info', 'sdod_graph_op_detail', 'sdod_graph_profile',
]


def _engine():
    lib = _lib.hip()
    if not getattr(lib, '_sdod_engine_typed', False):
        P, I = ctypes.c_void_p, ctypes.c_int
        lib.sdod_model_config_sd14.argtypes = [ctypes.POINTER(ModelConfig)]
        lib.sdod_model_config_sd14.restype = None
        lib.sdod_model_config_sd21.argtypes = [ctypes.POINTER(ModelConfig)]
        lib.sdod_model_config_sd21.restype = None
        lib.sdod_graph_create.argtypes = [ctypes.POINTER(P), I, ctypes.POINTER(ModelConfig), I]
        lib.sdod_graph_destroy.argtypes = [P]
        lib.sdod_graph_num_params.argtypes = [P]
        lib.sdod_graph_param_info.argtypes = [P, I, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(I), ctypes.POINTER(ctypes.c_int64)]
        lib.sdod_graph_set_param.argtypes = [P, ctypes.c_char_p, P, I, ctypes.POINTER(ctypes.c_int64), I]
        lib.sdod_graph_load_file.argtypes = [P, ctypes.c_char_p, ctypes.c_char_p]
        lib.sdod_graph_finalize.argtypes = [P]
        lib.sdod_graph_io.argtypes = [P, I, I, ctypes.POINTER(P), ctypes.POINTER(ctypes.c_size_t)]
        lib.sdod_graph_execute.argtypes = [P, P, I]
        lib.sdod_graph_check.argtypes = [P]
        lib.sdod_graph_stats.argtypes = [P, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(I),
                                         ctypes.POINTER(ctypes.c_double)]
        lib.sdod_graph_tune_info.argtypes = [P, ctypes.POINTER(I), ctypes.POINTER(I), ctypes.c_char_p, I]
        lib.sdod_graph_num_ops.argtypes = [P]
        lib.sdod_graph_op_info.argtypes = [P, I, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        lib.sdod_graph_profile.argtypes = [P, P, I, P, I]
        lib.sdod_graph_op_detail.argtypes = [P, I, ctypes.POINTER(ctypes.c_char_p)]
        lib._sdod_engine_typed = True
    return lib


def sd14_config(latent_h=64, latent_w=64):
    cfg = ModelConfig()
    _engine().sdod_model_config_sd14(ctypes.byref(cfg))
    cfg.latent_h, cfg.latent_w = latent_h, latent_w
    return cfg


def sd21_config(latent_h=96, latent_w=96):
    """SD v2.1-768 shapes (BASELINE config 5): UNet with 64-wide heads / context 1024, OpenCLIP ViT-H/14 text tower"""
    cfg = ModelConfig()
    _engine().sdod_model_config_sd21(ctypes.byref(cfg))
    cfg.latent_h, cfg.latent_w = latent_h, latent_w
    return cfg


class _DevView:
    """exposes raw device memory through __cuda_array_interface__ so torch can alias it without a copy"""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': typestr, 'data': (int(ptr), False), 'version': 2}


def device_view(ptr, shape, dtype, device):
    typestr = {torch.float16: '<f2', torch.float32: '<f4', torch.int32: '<i4', torch.uint8: '|u1'}[dtype]
    return torch.as_tensor(_DevView(ptr, shape, typestr), device=device)


class Graph:
    """One compiled graph: parameters in, I/O slots as torch views, execute() on the current stream.
    Without a GPU only the parameter table can be read (what the checkpoint converter needs); everything that touches
    device memory raises."""

    def __init__(self, kind, cfg, batch, device='cuda:0'):
        self._lib = _engine()
        self._h = ctypes.c_void_p()
        self.kind, self.cfg, self.batch = kind, cfg, batch
        self.device = torch.device(device)
        with self._on_device():
            check(self._lib.sdod_graph_create(ctypes.byref(self._h), kind, ctypes.byref(cfg), batch))
        self.finalized = False

    def _on_device(self):
        return torch.cuda.device(self.device) if torch.cuda.is_available() else contextlib.nullcontext()

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            self._lib.sdod_graph_destroy(h)
            self._h = None

    def param_table(self):
        out = []
        n = self._lib.sdod_graph_num_params(self._h)
        name = ctypes.c_char_p(); nd = ctypes.c_int(); shape = (ctypes.c_int64 * 4)()
        for i in range(n):
            check(self._lib.sdod_graph_param_info(self._h, i, ctypes.byref(name), ctypes.byref(nd), shape))
            out.append((name.value.decode(), tuple(shape[:nd.value])))
        return out

    def set_param(self, name, tensor):
        if hasattr(tensor, 'payload'):     # weights.QuantU8: the library dequantises with the reference's arithmetic
            buf = tensor.payload()
            shape = (ctypes.c_int64 * max(len(tensor.shape), 1))(*tensor.shape)
            check(self._lib.sdod_graph_set_param(self._h, name.encode(), ctypes.cast(ctypes.c_char_p(buf), ctypes.c_void_p), 2, shape,
                                                 len(tensor.shape)))
            return
        t = tensor.detach().cpu().contiguous()
        if t.dtype not in (torch.float32, torch.float16):
            t = t.float()
        shape = (ctypes.c_int64 * max(t.dim(), 1))(*t.shape)
        check(self._lib.sdod_graph_set_param(self._h, name.encode(), ctypes.c_void_p(t.data_ptr()),
                                             1 if t.dtype == torch.float32 else 0, shape, t.dim()))

    def load_state_dict(self, sd, prefix=''):
        """sd: mapping of (prefix+)ldm/HF names to tensors in canonical layout; every graph parameter must be present."""
        with torch.cuda.device(self.device):
            for name, _ in self.param_table():
                key = prefix + name
                if key not in sd:
                    raise KeyError(f'missing parameter {key}')
                self.set_param(name, sd[key])

    def load_file(self, path, prefix=''):
        with self._on_device():
            check(self._lib.sdod_graph_load_file(self._h, path.encode(), prefix.encode()))

    def finalize(self):
        with torch.cuda.device(self.device):
            check(self._lib.sdod_graph_finalize(self._h))
        self.finalized = True
        return self

    def _io(self, is_out, idx):
        p = ctypes.c_void_p(); n = ctypes.c_size_t()
        check(self._lib.sdod_graph_io(self._h, 1 if is_out else 0, idx, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def io_tensor(self, is_out, idx, shape, dtype):
        ptr, nbytes = self._io(is_out, idx)
        numel = int(np.prod(shape))
        assert numel * torch.empty((), dtype=dtype).element_size() == nbytes, (shape, dtype, nbytes)
        return device_view(ptr, shape, dtype, self.device)

    def execute(self, use_hip_graph=False, static_unchanged=False):
        """static_unchanged: the static inputs (UNet: text context) are the same as in the previous execute()"""
        flags = (1 if use_hip_graph else 0) | (2 if static_unchanged else 0)
        check(self._lib.sdod_graph_execute(self._h, ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream), flags))

    def check(self):
        """raises once a launch of this device has failed in a way it could not report itself (a GroupNorm grid barrier that
        timed out): call it after synchronising; execute() makes the same check on entry"""
        check(self._lib.sdod_graph_check(self._h))

    def op_table(self):
        """[(label, flops, bytes)] for every launch of the graph"""
        out = []
        lab = ctypes.c_char_p(); fl = ctypes.c_double(); by = ctypes.c_double()
        for i in range(self._lib.sdod_graph_num_ops(self._h)):
            check(self._lib.sdod_graph_op_info(self._h, i, ctypes.byref(lab), ctypes.byref(fl), ctypes.byref(by)))
            out.append((lab.value.decode(), fl.value, by.value))
        return out

    def op_details(self):
        out = []
        det = ctypes.c_char_p()
        for i in range(self._lib.sdod_graph_num_ops(self._h)):
            check(self._lib.sdod_graph_op_detail(self._h, i, ctypes.byref(det)))
            out.append(det.value.decode())
        return out

    def profile(self, iters=3):
        """per-launch durations in ms (HIP events on the current stream, eager); same order as op_table()"""
        n = self._lib.sdod_graph_num_ops(self._h)
        ms = (ctypes.c_float * n)()
        check(self._lib.sdod_graph_profile(self._h, ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream), iters, ms, n))
        return list(ms)

    def tune_source(self):
        """where this graph's GEMM tiles came from: table file(s), shapes found there, shapes timed in this process"""
        a = ctypes.c_int(); b = ctypes.c_int(); buf = ctypes.create_string_buffer(1024)
        check(self._lib.sdod_graph_tune_info(self._h, ctypes.byref(a), ctypes.byref(b), buf, 1024))
        import os
        return {'table': ' + '.join(os.path.basename(p) for p in buf.value.decode().split(' + ') if p), 'shapes_from_table': a.value,
                'shapes_tuned_in_process': b.value}

    def stats(self):
        w = ctypes.c_size_t(); a = ctypes.c_size_t(); n = ctypes.c_int(); f = ctypes.c_double()
        check(self._lib.sdod_graph_stats(self._h, ctypes.byref(w), ctypes.byref(a), ctypes.byref(n), ctypes.byref(f)))
        return {'weight_bytes': w.value, 'arena_bytes': a.value, 'launches': n.value, 'flops': f.value}


class UNet(Graph):
    def __init__(self, cfg, batch, device='cuda:0'):
        super().__init__(UNET, cfg, batch, device)

    def finalize(self):
        super().finalize()
        c, b = self.cfg, self.batch
        self.x = self.io_tensor(False, 0, (b, c.latent_channels, c.latent_h, c.latent_w), torch.float32)
        self.temb_width = self._io(False, 1)[1] // (2 * b)       # projected time conditioning (TEMB graph output width)
        self.temb = self.io_tensor(False, 1, (b, self.temb_width), torch.float16)
        self.ctx = self.io_tensor(False, 2, (b, c.context_len, c.context_dim), torch.float16)
        self.eps = self.io_tensor(True, 0, (b, c.latent_h, c.latent_w, c.latent_channels), torch.float16)
        return self


class Temb(Graph):
    def __init__(self, cfg, batch, device='cuda:0'):
        super().__init__(TEMB, cfg, batch, device)

    def finalize(self):
        super().finalize()
        self.t = self.io_tensor(False, 0, (self.batch,), torch.float32)
        self.width = self._io(True, 0)[1] // (2 * self.batch)
        self.out = self.io_tensor(True, 0, (self.batch, self.width), torch.float16)
        return self


class TextEncoder(Graph):
    def __init__(self, cfg, batch, device='cuda:0'):
        super().__init__(TEXT_ENCODER, cfg, batch, device)

    def finalize(self):
        super().finalize()
        c = self.cfg
        self.ids = self.io_tensor(False, 0, (self.batch, c.context_len), torch.int32)
        self.out = self.io_tensor(True, 0, (self.batch, c.context_len, c.context_dim), torch.float16)
        return self


class VaeDecoder(Graph):
    def __init__(self, cfg, batch, device='cuda:0'):
        super().__init__(VAE_DECODER, cfg, batch, device)

    def finalize(self):
        super().finalize()
        c, b = self.cfg, self.batch
        self.z = self.io_tensor(False, 0, (b, c.latent_channels, c.latent_h, c.latent_w), torch.float32)
        self.img = self.io_tensor(True, 0, (b, 8 * c.latent_h, 8 * c.latent_w, 3), torch.float16)
        return self
