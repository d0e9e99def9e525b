"""ctypes loader for the in-tree HIP libraries.  Fails loudly: there is no CPU fallback."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.normpath(os.path.join(_HERE, '..', '..', 'lib'))

_cache = {}


def load(name):
    """Return the CDLL for lib/<name>; raises ImportError with build instructions if missing."""
    if name in _cache:
        return _cache[name]
    # SDOD_LIBSDOD=<file in lib/>: developer builds of the same library (sanitizer build: libsdod_asan.so)
    path = os.path.join(LIB_DIR, os.environ.get('SDOD_LIBSDOD', name) if name == 'libsdod.so' else name)
    if not os.path.exists(path):
        raise ImportError(
            f'{path} not found: build the MI355X extension first '
            f'(python -c "import __graft_entry__ as g; g.build()" or make -C stable-diffusion-on-device_amd). '
            f'There is no CPU fallback for the HIP path.')
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    _cache[name] = lib
    return lib


class SdodError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f'[libsdod status {code}] {msg}')
        self.code = code


def hip():
    lib = load('libsdod.so')
    if not getattr(lib, '_sdod_typed', False):
        _declare(lib)
        lib._sdod_typed = True
    return lib


c_void_p, c_int, c_float, c_size_t, c_char_p = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_char_p


class PlmsUpdateArgs(ctypes.Structure):
    """mirror of `struct sdod_plms_update_args` (include/sdod_hip.h)"""
    _fields_ = [('eps_nhwc', ctypes.c_void_p), ('e_out', ctypes.c_void_p), ('x', ctypes.c_void_p), ('old1', ctypes.c_void_p),
                ('old2', ctypes.c_void_p), ('old3', ctypes.c_void_p), ('x_stage', ctypes.c_void_p), ('temb_row', ctypes.c_void_p),
                ('temb_dst', ctypes.c_void_p)] + \
               [(k, ctypes.c_int) for k in ('n', 'c', 'hw', 'uncond_first', 'mode', 'v_pred', 'stage_reps', 'temb_width', 'temb_reps')] + \
               [(k, ctypes.c_float) for k in ('guidance', 'vc0', 'vc1', 'c0', 'c1', 'c2', 'c3', 'div', 'sqrt_one_minus_at', 'sqrt_at',
                                              'sqrt_a_prev', 'dir_coef')]


class DpmStepArgs(ctypes.Structure):
    """mirror of `struct sdod_dpm_step_args` (include/sdod_hip.h)"""
    _fields_ = [('eps_nhwc', ctypes.c_void_p), ('e_out', ctypes.c_void_p), ('x', ctypes.c_void_p), ('y_prev', ctypes.c_void_p),
                ('x_stage', ctypes.c_void_p), ('temb_row', ctypes.c_void_p), ('temb_dst', ctypes.c_void_p)] + \
               [(k, ctypes.c_int) for k in ('n', 'c', 'hw', 'uncond_first', 'mode', 'order', 'stage_reps', 'temb_width', 'temb_reps')] + \
               [(k, ctypes.c_float) for k in ('guidance', 'sigma_s', 'alpha_s', 'sigma_ratio', 'c_prev', 'c_cur')]


class GemmDesc(ctypes.Structure):
    """mirror of `struct sdod_gemm_desc` (include/sdod_hip.h)"""
    _fields_ = [
        ('a', c_void_p), ('a2', c_void_p), ('w', c_void_p), ('bias', c_void_p), ('row_bias', c_void_p),
        ('residual', c_void_p), ('out', c_void_p), ('workspace', c_void_p), ('workspace_bytes', c_size_t),
        ('M', c_int), ('N', c_int), ('K', c_int),
        ('lda', c_int), ('ldw', c_int), ('ldo', c_int), ('ldr', c_int),
        ('a_mode', c_int),
        ('n_img', c_int), ('h_in', c_int), ('w_in', c_int), ('c0', c_int), ('c1', c_int),
        ('stride', c_int), ('upsample', c_int), ('ksize', c_int), ('rows_per_img', c_int), ('ld_row_bias', c_int),
        ('act', c_int), ('alpha', c_float), ('bias_on_m', c_int), ('split_k', c_int), ('tile', c_int),
        ('geglu', c_int), ('k_tail', c_int), ('t0', c_void_p), ('t1', c_void_p), ('tc0', c_int), ('tc1', c_int),
        ('bias2', c_void_p), ('ln', c_int), ('ln_s', c_void_p), ('ln_eps', c_float), ('phase', c_int),
        ('wq', c_int), ('w_scale', c_void_p), ('w_off', c_void_p), ('fix_counters', c_void_p), ('xcd_panels', c_int),
        ('w_img_stride', c_int), ('vec_img_stride', c_int), ('softmax_cols', c_int),
    ]


def _declare(lib):
    P = c_void_p
    sig = {
        'sdod_gemm_f16': (c_int, [ctypes.POINTER(GemmDesc), P]),
        'sdod_gemm_fixup': (c_int, [P]),
        'sdod_gemm_fixup_counters': (c_size_t, []),
        'sdod_gemm_workspace_bytes': (c_size_t, [ctypes.POINTER(GemmDesc)]),
        'sdod_gemm_plan': (c_int, [ctypes.POINTER(GemmDesc), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
        'sdod_gemm_halo_ok': (c_int, [ctypes.POINTER(GemmDesc), c_int]),
        'sdod_gemm_panel_ok': (c_int, [ctypes.POINTER(GemmDesc), c_int]),
        'sdod_gemm_xcd_panels': (c_int, [ctypes.POINTER(GemmDesc)]),
        'sdod_gemm_time': (c_int, [ctypes.POINTER(GemmDesc), P, c_int, ctypes.POINTER(c_float)]),
        'sdod_gemm_time_cold': (c_int, [ctypes.POINTER(GemmDesc), P, c_int, P, c_size_t, ctypes.POINTER(c_float), ctypes.POINTER(c_float)]),
        'sdod_gemm_num_tiles': (c_int, []),
        'sdod_gemm_tile_info': (c_int, [c_int, ctypes.POINTER(c_int)]),
        'sdod_gemm_tile_shape': (c_int, [c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
        'sdod_gemm_reduce_info': (c_int, [ctypes.POINTER(GemmDesc), P]),
        'sdod_group_norm_reduce_ok': (c_int, [c_int, c_int, c_int, c_int]),
        'sdod_group_norm_reduce_nhwc': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, P]),
        'sdod_group_norm_launches': (c_int, [c_int, c_int, c_int, c_int]),
        'sdod_group_norm_workspace_bytes': (c_size_t, [c_int, c_int]),
        'sdod_group_norm_layout': (c_int, [c_int, c_int] + [ctypes.POINTER(c_size_t)] * 5),
        'sdod_group_norm_status': (c_int, []),
        'sdod_group_norm_clear_error': (c_int, []),
        'sdod_group_norm_path': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
        'sdod_group_norm_nchw_workspace_bytes': (c_size_t, [c_int, c_int]),
        'sdod_group_norm_nchw': (c_int, [P, P, P, P, c_int, c_int, ctypes.c_longlong, c_int, c_float, c_int, c_int, P, P]),
        'sdod_group_norm_nhwc': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int, P, P]),
        'sdod_layer_norm_f16': (c_int, [P, P, P, P, c_int, c_int, c_float, P]),
        'sdod_ln_fold_f16': (c_int, [P, c_int, c_int, c_int, P, P, P, P, P, P]),
        'sdod_compose_linear_f16': (c_int, [P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, P, P, P, P]),
        'sdod_attention_f16': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, P]),
        'sdod_xattn_fold_f16': (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, P, c_int, c_int, c_int, c_float, P, P, P, P, P]),
        'sdod_softmax_rows_f16': (c_int, [P, P, c_int, c_int, P]),
        'sdod_geglu_f16': (c_int, [P, P, c_int, c_int, P]),
        'sdod_act_f16': (c_int, [P, P, c_size_t, c_int, P]),
        'sdod_add_f16': (c_int, [P, P, P, c_size_t, P]),
        'sdod_concat_channels_f16': (c_int, [P, P, P, c_size_t, c_int, c_int, P]),
        'sdod_im2col3x3_small_f16': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
        'sdod_latent_im2col_f16': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_float, P]),
        'sdod_conv_in_f16': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, P]),
        'sdod_plms_update': (c_int, [ctypes.POINTER(PlmsUpdateArgs), P]),
        'sdod_dpm_step': (c_int, [ctypes.POINTER(DpmStepArgs), P]),
        'sdod_nchw_f32_to_nhwc_f16': (c_int, [P, P, c_int, c_int, c_int, c_float, P]),
        'sdod_nhwc_f16_to_nchw_f32': (c_int, [P, P, c_int, c_int, c_int, P]),
        'sdod_latent_prep_f16': (c_int, [P, P, P, P, c_int, c_int, c_int, c_float, P]),
        'sdod_embedding_f16': (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
        'sdod_timestep_features_f16': (c_int, [P, P, c_int, c_int, P]),
        'sdod_stage_unet_inputs': (c_int, [P, P, c_size_t, c_int, P, P, c_size_t, c_int, P]),
        'sdod_randn_f32': (c_int, [P, P, c_size_t, ctypes.c_uint64, ctypes.c_uint64, P]),
        'sdod_cfg_combine': (c_int, [P, P, c_int, c_int, c_int, c_float, c_int, c_int, P]),
        'sdod_dpm_update': (c_int, [P, P, P, c_size_t, c_int, c_float, c_float, c_float, c_float, c_float, P]),
        'sdod_ddim_step_f32': (c_int, [P, P, c_size_t, c_float, c_float, c_float, c_float, P]),
        'sdod_lincomb4_f32': (c_int, [P, P, P, P, P, c_float, c_float, c_float, c_float, c_float, c_size_t, P]),
        'sdod_image_to_u8': (c_int, [P, P, c_size_t, c_float, c_float, c_int, P]),
        'sdod_hip_last_error': (c_char_p, []),
        'sdod_hip_device_info': (c_int, [ctypes.POINTER(c_int), ctypes.POINTER(c_size_t), c_char_p, c_int]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args


HIP_SYMBOLS = [
    'sdod_gemm_f16', 'sdod_gemm_fixup', 'sdod_gemm_fixup_counters', 'sdod_gemm_workspace_bytes', 'sdod_gemm_plan', 'sdod_gemm_halo_ok', 'sdod_gemm_panel_ok', 'sdod_gemm_xcd_panels', 'sdod_gemm_num_tiles', 'sdod_gemm_tile_shape', 'sdod_gemm_tile_info', 'sdod_gemm_time', 'sdod_gemm_time_cold', 'sdod_group_norm_workspace_bytes', 'sdod_group_norm_layout', 'sdod_group_norm_status', 'sdod_group_norm_clear_error', 'sdod_l2_prefetch', 'sdod_group_norm_launches', 'sdod_group_norm_path', 'sdod_group_norm_nhwc', 'sdod_group_norm_nchw', 'sdod_group_norm_nchw_workspace_bytes', 'sdod_gemm_reduce_info', 'sdod_group_norm_reduce_ok', 'sdod_group_norm_reduce_nhwc',
    'sdod_layer_norm_f16', 'sdod_ln_fold_f16', 'sdod_compose_linear_f16', 'sdod_attention_f16', 'sdod_xattn_fold_f16', 'sdod_softmax_rows_f16', 'sdod_geglu_f16', 'sdod_act_f16',
    'sdod_add_f16', 'sdod_concat_channels_f16', 'sdod_im2col3x3_small_f16', 'sdod_latent_im2col_f16', 'sdod_conv_in_f16', 'sdod_plms_update', 'sdod_dpm_step', 'sdod_nchw_f32_to_nhwc_f16',
    'sdod_nhwc_f16_to_nchw_f32', 'sdod_latent_prep_f16', 'sdod_embedding_f16', 'sdod_timestep_features_f16', 'sdod_cfg_combine', 'sdod_stage_unet_inputs', 'sdod_randn_f32',
    'sdod_dpm_update', 'sdod_ddim_step_f32', 'sdod_lincomb4_f32', 'sdod_image_to_u8', 'sdod_hip_last_error',
    'sdod_hip_device_info',
]


def check(code):
    if code != 0:
        msg = hip().sdod_hip_last_error()
        raise SdodError(code, msg.decode() if msg else 'unknown error')
