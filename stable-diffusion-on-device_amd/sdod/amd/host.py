"""ctypes bindings of the host-side C ABI (include/sdod_host.h, include/libsdod.h): tokenizer, DPM-Solver++ tables
and the reference's generation-driver API.  One C++ implementation serves both this Python loop and the C driver."""
import ctypes

import numpy as np

from . import _lib

HOST_SYMBOLS = [
    'sdod_tokenizer_create', 'sdod_tokenizer_destroy', 'sdod_tokenizer_encode', 'sdod_tokenizer_special', 'sdod_dpm_create',
    'sdod_dpm_destroy', 'sdod_dpm_prepare', 'sdod_dpm_table', 'sdod_dpm_coef', 'sdod_dpm_update_host', 'sdod_context_set_seed', 'sdod_context_set_initial_latent',
]
LIBSDOD_SYMBOLS = [
    'libsdod_setup', 'libsdod_set_steps', 'libsdod_set_log_level', 'libsdod_ref_context', 'libsdod_release',
    'libsdod_generate_image', 'libsdod_get_error_description', 'libsdod_get_last_error_extra_info',
]


def _host():
    lib = _lib.load('libsdod.so')
    if not getattr(lib, '_sdod_host_typed', False):
        P, U, I, F = ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_float
        PP = ctypes.POINTER
        lib.sdod_tokenizer_create.argtypes = [PP(P), ctypes.c_char_p]
        lib.sdod_tokenizer_destroy.argtypes = [P]
        lib.sdod_tokenizer_encode.argtypes = [P, ctypes.c_char_p, P, U]
        lib.sdod_tokenizer_special.argtypes = [P, PP(U), PP(U), PP(U)]
        lib.sdod_dpm_create.argtypes = [PP(P), U, F, F]
        lib.sdod_dpm_destroy.argtypes = [P]
        lib.sdod_dpm_prepare.argtypes = [P, U]
        lib.sdod_dpm_table.argtypes = [P, I, P, U, PP(U)]
        lib.sdod_dpm_coef.argtypes = [P, U, PP(I), PP(F), PP(F), PP(F), PP(F), PP(F)]
        lib.sdod_dpm_update_host.argtypes = [P, U, P, P, P, U]
        lib.sdod_context_set_seed.argtypes = [P, U]
        lib.sdod_context_set_initial_latent.argtypes = [P, P, ctypes.c_size_t]
        lib.libsdod_setup.argtypes = [PP(P), ctypes.c_char_p, U, U, U, U, U, I]
        lib.libsdod_set_steps.argtypes = [P, U]
        lib.libsdod_set_log_level.argtypes = [P, U]
        lib.libsdod_ref_context.argtypes = [P]
        lib.libsdod_release.argtypes = [P]
        lib.libsdod_generate_image.argtypes = [P, ctypes.c_char_p, F, PP(PP(ctypes.c_ubyte)), PP(U)]
        lib.libsdod_get_error_description.argtypes = [I]
        lib.libsdod_get_error_description.restype = ctypes.c_char_p
        lib.libsdod_get_last_error_extra_info.argtypes = [I, P]
        lib.libsdod_get_last_error_extra_info.restype = ctypes.c_char_p
        lib.sdod_hip_last_error.restype = ctypes.c_char_p
        lib._sdod_host_typed = True
    return lib


def _check(lib, rc):
    if rc != 0:
        msg = lib.sdod_hip_last_error()
        raise _lib.SdodError(rc, msg.decode() if msg else 'unknown error')


class Tokenizer:
    """CLIP BPE tokenizer (C++: csrc/libsdod/tokenizer.cpp); reads the reference's ctokenizer.txt format."""

    def __init__(self, path):
        self._lib = _host()
        self._h = ctypes.c_void_p()
        _check(self._lib, self._lib.sdod_tokenizer_create(ctypes.byref(self._h), str(path).encode()))
        s, e, v = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
        _check(self._lib, self._lib.sdod_tokenizer_special(self._h, ctypes.byref(s), ctypes.byref(e), ctypes.byref(v)))
        self.start_token, self.end_token, self.vocab_size = s.value, e.value, v.value

    def __del__(self):
        if getattr(self, '_h', None):
            self._lib.sdod_tokenizer_destroy(self._h)
            self._h = None

    def encode(self, text, context_len=77):
        out = np.zeros(context_len, np.uint16)
        _check(self._lib, self._lib.sdod_tokenizer_encode(self._h, text.encode('utf-8'), out.ctypes.data, context_len))
        return out


class DpmSolver:
    """DPM-Solver++(2M) host tables (C++: csrc/libsdod/dpm_solver.cpp), bit-identical to the reference's."""
    TABLES = {'ts': 0, 'log_alphas': 1, 'lambdas': 2, 'sigmas': 3, 'alphas': 4, 'phis': 5, 'i2rs': 6, 'model_ts': 7,
              'all_t': 8, 'all_log_alpha': 9}

    def __init__(self, timesteps=1000, lin_start=0.00085, lin_end=0.0120):
        self._lib = _host()
        self._h = ctypes.c_void_p()
        _check(self._lib, self._lib.sdod_dpm_create(ctypes.byref(self._h), timesteps, lin_start, lin_end))
        self.steps = 0

    def __del__(self):
        if getattr(self, '_h', None):
            self._lib.sdod_dpm_destroy(self._h)
            self._h = None

    def prepare(self, steps):
        _check(self._lib, self._lib.sdod_dpm_prepare(self._h, steps))
        self.steps = steps
        return self.table('model_ts')

    def table(self, name):
        n = ctypes.c_uint()
        _check(self._lib, self._lib.sdod_dpm_table(self._h, self.TABLES[name], None, 0, ctypes.byref(n)))
        out = np.zeros(n.value, np.float32)
        _check(self._lib, self._lib.sdod_dpm_table(self._h, self.TABLES[name], out.ctypes.data, n.value, ctypes.byref(n)))
        return out

    def coef(self, step):
        o = ctypes.c_int()
        f = [ctypes.c_float() for _ in range(5)]
        _check(self._lib, self._lib.sdod_dpm_coef(self._h, step, ctypes.byref(o), *[ctypes.byref(v) for v in f]))
        return dict(order=o.value, sigma_s=f[0].value, alpha_s=f[1].value, sigma_ratio=f[2].value, c_prev=f[3].value,
                    c_cur=f[4].value)

    def update_host(self, step, x, eps, y_prev):
        _check(self._lib, self._lib.sdod_dpm_update_host(self._h, step, x.ctypes.data, eps.ctypes.data, y_prev.ctypes.data, x.size))


class LibSdod:
    """The reference's C API, called exactly as simple_app.cpp does (csrc/libsdod/test/simple_app.cpp:7-36)."""

    def __init__(self, models_dir, latent_channels=4, latent_spatial=64, upscale_factor=8, steps=20, log_level=1, use_htp=1):
        self.lib = _host()
        self.ctx = ctypes.c_void_p()
        self.status = self.lib.libsdod_setup(ctypes.byref(self.ctx), str(models_dir).encode(), latent_channels, latent_spatial,
                                             upscale_factor, steps, log_level, use_htp)
        self.image_side = latent_spatial * upscale_factor

    def error(self, code=None):
        code = self.status if code is None else code
        d = self.lib.libsdod_get_error_description(code)
        e = self.lib.libsdod_get_last_error_extra_info(code, self.ctx)
        return (d.decode() if d else None, e.decode() if e else None)

    def set_seed(self, seed):
        return self.lib.sdod_context_set_seed(self.ctx, seed)

    def set_initial_latent(self, x):
        x = np.ascontiguousarray(x, np.float32)
        return self.lib.sdod_context_set_initial_latent(self.ctx, x.ctypes.data, x.size)

    def set_steps(self, steps):
        return self.lib.libsdod_set_steps(self.ctx, steps)

    def generate(self, prompt, guidance=7.5):
        buf = ctypes.POINTER(ctypes.c_ubyte)()
        n = ctypes.c_uint(0)
        rc = self.lib.libsdod_generate_image(self.ctx, prompt.encode('utf-8'), guidance, ctypes.byref(buf), ctypes.byref(n))
        if rc != 0:
            return rc, None
        img = np.ctypeslib.as_array(buf, shape=(n.value,)).copy().reshape(self.image_side, self.image_side, 3)
        ctypes.CDLL(None).free(buf)
        return 0, img

    def release(self):
        if self.ctx:
            rc = self.lib.libsdod_release(self.ctx)
            return rc
        return 0
