"""oracle/pipeline_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

CPU fp32 restatement of the two sampler loops of the path:
  * plms_sample: config 1's CPU reference, the public ldm `PLMSSampler` (eta 0) + txt2img's CFG batch [uncond; cond]
    (ldm is not under /root/reference -> "parity unpinned");
  * dpm_sample: the reference driver's loop, context.cpp:342-382, with the solver arithmetic taken from
    oracle/sdod_oracle.c (which is pinned bit-for-bit to the reference's dpm_solver.cpp).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import ctypes

import numpy as np
import torch

from .sd_torch import timestep_embedding


def _alphas_cumprod():
    betas = torch.linspace(0.00085 ** 0.5, 0.0120 ** 0.5, 1000, dtype=torch.float64) ** 2
    return np.cumprod((1.0 - betas).numpy(), axis=0)


@torch.no_grad()
def guided_eps(unet, x, t, ctx_uc, ctx_c, scale):
    n = x.shape[0]
    x_in = torch.cat([x] * 2)
    t_in = torch.cat([t] * 2)
    c_in = torch.cat([ctx_uc.expand(n, -1, -1), ctx_c.expand(n, -1, -1)])
    e_u, e_c = unet(x_in, t_in, c_in).chunk(2)
    return e_u, e_c


@torch.no_grad()
def plms_sample(unet, ctx_uc, ctx_c, x_T, steps=20, scale=7.5, trace=None, parameterization='eps'):
    """parameterization='v' (SD 2.1-768): the guided model output is v and eps = sqrt(abar_t) v + sqrt(1 - abar_t) x
    (public ldm v2 `predict_eps_from_z_and_v`, applied in PLMSSampler.get_model_output)"""
    ac = torch.from_numpy(_alphas_cumprod()).to(torch.float32)
    c = 1000 // steps
    ddim_timesteps = np.asarray(list(range(0, 1000, c))) + 1
    alphas = ac[ddim_timesteps]
    alphas_prev = np.asarray([ac[0]] + ac[ddim_timesteps[:-1]].tolist())
    sqrt_one_minus_alphas = np.sqrt(1. - alphas)
    b = x_T.shape[0]

    def model_out(x, t):
        e_u, e_c = guided_eps(unet, x, t, ctx_uc, ctx_c, scale)
        out = e_u + scale * (e_c - e_u)
        if parameterization == 'v':
            a = ac[t].reshape(-1, 1, 1, 1)
            out = a.sqrt() * out + (1. - a).sqrt() * x
        return out

    def x_prev_of(x, e_t, index):
        a_t = torch.full((b, 1, 1, 1), float(alphas[index]))
        a_prev = torch.full((b, 1, 1, 1), float(alphas_prev[index]))
        s1m = torch.full((b, 1, 1, 1), float(sqrt_one_minus_alphas[index]))
        pred_x0 = (x - s1m * e_t) / a_t.sqrt()
        dir_xt = (1. - a_prev).sqrt() * e_t
        return a_prev.sqrt() * pred_x0 + dir_xt

    img = x_T.clone()
    time_range = np.flip(ddim_timesteps)
    old_eps = []
    for i, step in enumerate(time_range):
        index = steps - i - 1
        ts = torch.full((b,), int(step), dtype=torch.long)
        ts_next = torch.full((b,), int(time_range[min(i + 1, len(time_range) - 1)]), dtype=torch.long)
        e_t = model_out(img, ts)
        if len(old_eps) == 0:
            x_prev = x_prev_of(img, e_t, index)
            e_t_next = model_out(x_prev, ts_next)
            e_t_prime = (e_t + e_t_next) / 2
        elif len(old_eps) == 1:
            e_t_prime = (3 * e_t - old_eps[-1]) / 2
        elif len(old_eps) == 2:
            e_t_prime = (23 * e_t - 16 * old_eps[-1] + 5 * old_eps[-2]) / 12
        else:
            e_t_prime = (55 * e_t - 59 * old_eps[-1] + 37 * old_eps[-2] - 9 * old_eps[-3]) / 24
        img = x_prev_of(img, e_t_prime, index)
        old_eps.append(e_t)
        old_eps = old_eps[-3:]
        if trace is not None:
            trace.append((int(step), index))
    return img


@torch.no_grad()
def dpm_sample(unet, oracle_lib, ctx_uc, ctx_c, x_T, steps=20, guidance=7.5):
    """context.cpp:342-382 for one image: e = g*e_cond + (1-g)*e_uncond, then DPMSolver::update"""
    h = oracle_lib.oracle_dpm_create(1000, 0.00085, 0.0120)
    oracle_lib.oracle_dpm_prepare(h, steps)
    mts = np.zeros(steps + 1, np.float32)
    oracle_lib.oracle_dpm_table(h, 7, mts.ctypes.data)
    x = x_T.clone()
    xh = np.ascontiguousarray(x.numpy().reshape(-1))
    for s in range(steps):
        t = torch.full((x.shape[0],), float(mts[s]))
        e_u, e_c = guided_eps(unet, torch.from_numpy(xh.reshape(x.shape)), t, ctx_uc, ctx_c, guidance)
        ec = np.ascontiguousarray(e_c.numpy().reshape(-1)); eu = np.ascontiguousarray(e_u.numpy().reshape(-1))
        e = np.zeros_like(ec)
        oracle_lib.oracle_cfg_combine(e.ctypes.data, ec.ctypes.data, eu.ctypes.data, guidance, e.size)
        oracle_lib.oracle_dpm_update(h, s, xh.ctypes.data, e.ctypes.data, xh.size)
    oracle_lib.oracle_dpm_destroy(h)
    return torch.from_numpy(xh.reshape(x.shape).copy())


@torch.no_grad()
def decode_u8(vae, z, mode=1, oracle_lib=None):
    """mode 1: ldm txt2img, 255*clamp((dec+1)/2, 0, 1) -> uint8 (truncation), HWC.  mode 0: the reference driver's
    uint8(clamp(255*f, 0, 255)) with f = (dec+1)/2 (context.cpp:392-395 via oracle_to_uint8)."""
    dec = vae(z)
    f = (dec + 1.0) / 2.0
    hwc = f.permute(0, 2, 3, 1).contiguous()
    if mode == 1:
        return (255. * torch.clamp(hwc, 0.0, 1.0)).numpy().astype(np.uint8)
    flat = np.ascontiguousarray(hwc.numpy().reshape(-1))
    out = np.zeros(flat.size, np.uint8)
    oracle_lib.oracle_to_uint8(out.ctypes.data, flat.ctypes.data, flat.size)
    return out.reshape(hwc.shape)


def unet_time_embedding(unet, t):
    with torch.no_grad():
        return unet.time_embed(timestep_embedding(t, unet.model_ch))
