"""Sampler schedules on the host (numpy), applied on the GPU by the kernels of include/sdod_hip.h.

PLMS follows the public ldm `PLMSSampler` with eta = 0 (config 1's CPU reference, `scripts/txt2img.py --plms`; the ldm
repository is not part of /root/reference).  Index arithmetic (timestep sequence, alpha-bar lookups)
is integer/float64 work done here and must match the oracle exactly; the per-element arithmetic runs on device in IEEE
fp32 in ldm's operation order.  DPM-Solver++(2M) uses the C++ solver tables (host.DpmSolver)."""
import numpy as np


def scaled_linear_alphas_cumprod(n=1000, linear_start=0.00085, linear_end=0.0120):
    """ldm make_beta_schedule('linear'): betas = linspace(sqrt(s), sqrt(e), n, float64)**2; cumprod(1-betas) -> float32"""
    betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, n, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas, axis=0).astype(np.float32)


class PlmsSchedule:
    def __init__(self, steps=20, n_train=1000):
        c = n_train // steps
        self.timesteps = np.asarray(list(range(0, n_train, c))) + 1          # ldm make_ddim_timesteps('uniform') + 1
        ac = scaled_linear_alphas_cumprod(n_train)
        self.alphas = ac[self.timesteps]                                     # float32
        self.alphas_prev = np.asarray([ac[0]] + ac[self.timesteps[:-1]].tolist(), dtype=np.float32)
        self.sqrt_one_minus_alphas = np.sqrt(np.float32(1.0) - self.alphas).astype(np.float32)
        self.steps = len(self.timesteps)
        self.time_range = np.flip(self.timesteps)                           # descending: 951, 901, ..., 1
        # v-prediction models (SD 2.1-768): eps = sqrt(abar_t) v + sqrt(1 - abar_t) x  (ldm predict_eps_from_z_and_v)
        self.sqrt_alphas = np.sqrt(self.alphas).astype(np.float32)

    def v_to_eps_coef(self, index):
        """(coefficient of v, coefficient of x) at ddim index `index`"""
        return float(self.sqrt_alphas[index]), float(self.sqrt_one_minus_alphas[index])

    def coef(self, index):
        """fp32 scalars of get_x_prev_and_pred_x0 at ddim index `index` (sigma_t = 0)"""
        a_t = np.float32(self.alphas[index]); a_prev = np.float32(self.alphas_prev[index])
        return dict(sqrt_one_minus_at=float(self.sqrt_one_minus_alphas[index]), sqrt_at=float(np.sqrt(a_t)),
                    sqrt_a_prev=float(np.sqrt(a_prev)), dir_coef=float(np.sqrt(np.float32(1.0) - a_prev)))


# Adams-Bashforth combinations of the eps history (newest first): (coefficients, divisor)
PLMS_ORDERS = {1: ((3.0, -1.0), 2.0), 2: ((23.0, -16.0, 5.0), 12.0), 3: ((55.0, -59.0, 37.0, -9.0), 24.0)}
