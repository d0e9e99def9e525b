"""The multi-GPU launch path on the one-GPU box:
  * RCCL itself: a fresh process initialises the `nccl` backend (world size 1) and runs the path's one collective,
    `broadcast_conditioning`, plus the latency mode's `all_gather_into_tensor`, on DEVICE tensors -- the library, its
    communicator set-up and both call sites have then executed on MI355X at least once;
  * `python bench.py --gpus 2` from a bare shell (no torchrun): bench.py spawns its own ranks.  Two ranks cannot share one
    device under RCCL, so this rehearsal puts both on cuda:0 over gloo (SDOD_BENCH_SHARE_DEVICE / SDOD_DIST_BACKEND);
    the 1/2/4/8-GPU curve itself is the driver's to measure on a whole node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RCCL_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'stable-diffusion-on-device_amd'))
import torch, torch.distributed as dist
from sdod.amd.pipeline import broadcast_conditioning
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
ctx = torch.arange(2 * 77 * 768, dtype=torch.float32, device='cuda').reshape(2, 77, 768).half()
want = ctx.clone()
out = broadcast_conditioning(ctx, 0)
mine = torch.randn(1, 64, 64, 4, device='cuda').half()
both = torch.empty_like(mine)
dist.all_gather_into_tensor(both, mine)
t = torch.tensor([1.5], dtype=torch.float64, device='cuda')
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert dist.get_backend() == 'nccl' and torch.equal(out, want) and torch.equal(both, mine) and float(t) == 1.5
print('RCCL_OK', torch.cuda.nccl.version())
dist.destroy_process_group()
'''


def test_rccl_backend_runs_the_conditioning_broadcast_on_device():
    # HSA_ENABLE_IPC_MODE_LEGACY=0: the pool's host driver only supports dmabuf IPC (RCCL's transport set-up fails with
    # `hipIpcGetMemHandle: invalid argument` otherwise); the image exports it, the test pins it
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29700 + os.getpid() % 200), HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', RCCL_WORKER, ROOT], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and 'RCCL_OK' in r.stdout


def test_bench_gpus_2_from_a_bare_shell():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    env.update(SDOD_BENCH_SHARE_DEVICE='1', SDOD_DIST_BACKEND='gloo')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                        '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=1200)
    print(r.stderr[-3000:])
    assert r.returncode == 0, r.stdout[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout                      # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['scaling'] == 'weak' and out['value'] > 0 and out['config']['global_batch'] == 2
    # the block the driver verifies an N-rank run with: every rank's own time, gathered by the collective itself
    rc = out['rccl']
    assert rc['world_size_seen'] == 2 and rc['ranks_gathered'] == 2 and len(rc['per_rank_ms']) == 2 and all(t > 0 for t in rc['per_rank_ms'])
    assert rc['backend'] == 'gloo' and rc['shared_device_rehearsal'] is True and rc['per_rank_device'] == [0, 0]
    assert abs(max(rc['per_rank_ms']) - out['ms_per_step']) < 1e-2
