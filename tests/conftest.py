import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'stable-diffusion-on-device_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the CPU oracle runs on this process's share of the host: os.cpu_count() reports every core of the machine (256 on a
    # GPU box whose per-GPU share is 16), and an oversubscribed intra-op pool makes the fp32 oracle several times slower
    try:
        import torch
        try:
            n = len(os.sched_getaffinity(0))
        except AttributeError:
            n = os.cpu_count() or 1
        torch.set_num_threads(max(1, min(16, n)))
    except ImportError:
        pass


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='session')
def oracle_lib():
    """ctypes handle of oracle/_build/libsdod_oracle.so (the CPU checker), built on demand with gcc."""
    import ctypes
    import subprocess
    so = os.path.join(ROOT, 'oracle', '_build', 'libsdod_oracle.so')
    src = os.path.join(ROOT, 'oracle', 'sdod_oracle.c')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'oracle', '_build', 'libsdod_oracle.so')])
    lib = ctypes.CDLL(so)
    V, U, F, I = ctypes.c_void_p, ctypes.c_uint, ctypes.c_float, ctypes.c_int
    lib.oracle_dpm_create.restype = V
    lib.oracle_dpm_create.argtypes = [U, F, F]
    lib.oracle_dpm_destroy.argtypes = [V]
    lib.oracle_dpm_prepare.argtypes = [V, U]
    lib.oracle_dpm_table.restype = U
    lib.oracle_dpm_table.argtypes = [V, I, V]
    lib.oracle_dpm_update.argtypes = [V, U, V, V, U]
    lib.oracle_timestep_features.argtypes = [F, U, V]
    lib.oracle_cfg_combine.argtypes = [V, V, V, F, U]
    lib.oracle_dequant_u8.argtypes = [V, V, ctypes.c_int32, F, U, I, I, F]
    lib.oracle_to_uint8.argtypes = [V, V, U]
    return lib
