"""world_size-2 gloo tests (CPU) of the N>1 path: the single collective (conditioning broadcast) and the image sharding."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'stable-diffusion-on-device_amd'))
    import torch.distributed as dist
    from sdod.amd.pipeline import broadcast_conditioning, initial_latent, shard_images
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ctx = torch.arange(2 * 77 * 768, dtype=torch.float32).reshape(2, 77, 768).half() if rank == 0 else torch.zeros(2, 77, 768, dtype=torch.float16)
    ctx = broadcast_conditioning(ctx, 0)
    mine = shard_images(5, rank, world)
    lat = [initial_latent(42, i, (4, 8, 8)) for i in mine]
    q.put((rank, float(ctx.float().sum()), mine, [float(t.sum()) for t in lat]))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    expect = float(torch.arange(2 * 77 * 768, dtype=torch.float32).half().float().sum())
    assert res[0][1] == expect and res[1][1] == expect           # rank 1 received rank 0's conditioning
    assert res[0][2] == [0, 1, 2] and res[1][2] == [3, 4]         # disjoint cover of the 5 images
    sys.path.insert(0, os.path.join(ROOT, 'stable-diffusion-on-device_amd'))
    from sdod.amd.pipeline import initial_latent
    for rank, _, idx, sums in res:                                # x_T depends on (seed, image index) only
        assert sums == [float(initial_latent(42, i, (4, 8, 8)).sum()) for i in idx]


def test_plms_schedule_indexing_exact():
    """scheduler indexing must be bit-exact (north_star): tau = arange(0,1000,50)+1 and the alpha-bar lookups"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, 'stable-diffusion-on-device_amd'))
    from sdod.amd.samplers import PLMS_ORDERS, PlmsSchedule, scaled_linear_alphas_cumprod
    s = PlmsSchedule(20)
    assert s.timesteps.tolist() == list(range(1, 1000, 50)) and s.time_range.tolist() == list(range(951, 0, -50))
    betas = torch.linspace(0.00085 ** 0.5, 0.0120 ** 0.5, 1000, dtype=torch.float64) ** 2
    ac = np.cumprod((1.0 - betas).numpy(), axis=0).astype(np.float32)
    assert np.array_equal(scaled_linear_alphas_cumprod(), ac)
    assert np.array_equal(s.alphas, ac[s.timesteps]) and s.alphas_prev[0] == ac[0] and np.array_equal(s.alphas_prev[1:], ac[s.timesteps[:-1]])
    c = s.coef(19)
    a_t = torch.full((1,), float(ac[951])); a_prev = torch.full((1,), float(ac[901]))
    assert c['sqrt_at'] == float(a_t.sqrt()) and c['sqrt_a_prev'] == float(a_prev.sqrt())
    assert c['dir_coef'] == float((1. - a_prev).sqrt()) and c['sqrt_one_minus_at'] == float(np.sqrt(1. - torch.from_numpy(ac)[951:952])[0])
    assert PLMS_ORDERS[3] == ((55.0, -59.0, 37.0, -9.0), 24.0)
    s50 = PlmsSchedule(50)
    assert s50.timesteps[0] == 1 and s50.timesteps[-1] == 981 and s50.steps == 50


def test_bench_launcher_spawns_one_rank_per_gpu(tmp_path):
    """`python bench.py --gpus N` from a bare shell (no torchrun): spawn_ranks starts N fresh processes with the rendezvous
    environment; exercised here with a stand-in worker that joins a gloo group and reports what it saw."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    worker = tmp_path / 'worker.py'
    worker.write_text(
        "import json, os, sys\n"
        "import torch, torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "t = torch.tensor([float(dist.get_rank() + 1)])\n"
        "dist.all_reduce(t)\n"
        "open(os.path.join(sys.argv[1], 'rank%d.json' % dist.get_rank()), 'w').write(json.dumps(\n"
        "    {k: os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR')} | {'sum': float(t)}))\n"
        "dist.destroy_process_group()\n")
    assert bench.spawn_ranks(2, [str(worker), str(tmp_path)]) == 0
    got = [json.loads((tmp_path / f'rank{r}.json').read_text()) for r in range(2)]
    assert [g['RANK'] for g in got] == ['0', '1'] and [g['LOCAL_RANK'] for g in got] == ['0', '1']
    assert all(g['WORLD_SIZE'] == '2' and g['MASTER_ADDR'] == '127.0.0.1' and g['sum'] == 3.0 for g in got)
    bad = tmp_path / 'bad.py'
    bad.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(3)\ntime.sleep(60)\n")
    assert bench.spawn_ranks(2, [str(bad)]) != 0          # a dead rank ends the job instead of hanging its peers
