#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X txt2img hot path.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): SD v1.4, 512x512, 20-step PLMS,
guidance 7.5: CLIP text encode (uncond + cond) -> 21 batched (uncond, cond) UNet evaluations with CFG + PLMS updates
-> VAE decode -> uint8 HWC image.  One "step" = one such image per GPU (weak scaling: every rank generates its own
image(s); rank 0 encodes the prompt and the conditioning is sent with ONE RCCL broadcast).  Synthetic seeded weights
(no checkpoint exists offline), fixed token ids, injected x_T; all inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0).  Extra blocks:
  roofline     -- for the dominant kernel family of the UNet launch list: algorithmic FLOPs (or bytes) per launch
                  divided by its average launch duration, measured in this process with HIP events on the launch stream
  cpu_baseline -- the oracle (PyTorch-CPU fp32 restatement, kind "port") timed on the host cores on a bounded sample
  parity       -- GPU fp16 vs CPU fp32 on the first guided UNet evaluation of this very run
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'stable-diffusion-on-device_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

METRIC = 'images/sec + per-UNet-step ms, SD v1.4 512x512 20-step PLMS, 1/2/4/8 MI355X'
PEAK_TFLOPS_F16 = 2500.0   # dense fp16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
PMC_SUMMARY = 'r03_pmc_summary.json'   # committed summary of the rocprofv3 --pmc passes of this round's build
# "a photograph of an astronaut riding a horse" needs the CLIP vocabulary, which is absent offline: fixed ids
IDS_COND = [49406, 320, 1125, 539, 550, 18376, 6765, 320, 4558] + [49407] * 68
IDS_UNCOND = [49406] + [49407] * 76


T0 = time.time()


def log(msg):
    """progress on stderr (the JSON line on stdout stays alone); also keeps long runs visibly alive"""
    print(f'[bench {time.time() - T0:7.1f}s] {msg}', file=sys.stderr, flush=True)


def host_threads():
    """CPU share of this process: affinity mask, capped at 16 (the GPU box's per-GPU share; os.cpu_count() reports
    every core of the host and would oversubscribe a cgroup-limited container)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def kernel_symbol(label):
    """launch-list label -> (kernel symbol as rocprofv3 prints it, fragment of its mangled name or None).  GEMM labels carry
    the tile id; its template arguments come from the library (sdod_gemm_tile_info), so the table cannot go stale."""
    import ctypes
    from sdod.amd import _lib
    if label.startswith('gemm_t'):
        info = (ctypes.c_int * 7)()
        if _lib.hip().sdod_gemm_tile_info(int(label[6:]), info) == 0:
            bm, bn, wm, wn, st, spec, ksub = list(info)
            if st == 0:
                return f'gemm_kernel<{bm}, {bn}, {wm}, {wn}>', f'gemm_kernelILi{bm}ELi{bn}ELi{wm}ELi{wn}EE'
            if spec == 2:    # halo-patch 3x3 convolution
                return f'conv_halo_kernel<{bm}, {bn}, {wm}, {wn}, {st}, false>', f'conv_halo_kernelILi{bm}ELi{bn}ELi{wm}ELi{wn}ELi{st}ELb0EE'
            if spec == 3:    # A-panel kernel (short-K, wide-N Linears)
                return f'gemm_apanel_kernel<{bm}, {bn}, {wm}, {wn}, {st}>', f'gemm_apanel_kernelILi{bm}ELi{bn}ELi{wm}ELi{wn}ELi{st}EE'
            return (f'gemm_glds_kernel<{bm}, {bn}, {wm}, {wn}, {st}, {"true" if spec else "false"}, false, {ksub}>',
                    f'gemm_glds_kernelILi{bm}ELi{bn}ELi{wm}ELi{wn}ELi{st}ELb{spec}ELb0ELi{ksub}EE')
    plain = {'gn_group': 'gn_group_kernel<', 'gn_group_red': 'gn_group_kernel<', 'gn_grid': 'gn_grid_kernel<', 'gn_small': 'gn_small_kernel<', 'gn_stats_apply': 'gn_stats_kernel<', 'splitk_reduce': 'splitk_reduce_',
             'attn_d40': 'attn_kernel<40,', 'attn_d64': 'attn_kernel<64,', 'attn_d80': 'attn_kernel<80,', 'attn_d160': 'attn_kernel<160,',
             'layer_norm': 'layer_norm_kernel', 'softmax_rows': 'softmax_rows_kernel<'}
    return plain.get(label, label), None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3, help='timed images per GPU')
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--images-per-gpu', type=int, default=1)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--overlap-decode', action='store_true', help='experiment: decode image i on a side stream while image i+1 is sampled (measured slower)')
    ap.add_argument('--no-hip-graph', action='store_true')
    ap.add_argument('--sampler', default='plms', choices=['plms', 'dpm'], help="plms: the headline workload (config 3); dpm: the reference "
                    "driver's DPM-Solver++(2M), e.g. --sampler dpm --sampler-steps 50 --images-per-gpu 2 = one rank's share of config 4")
    ap.add_argument('--sampler-steps', type=int, default=20)
    ap.add_argument('--cfg-split', action='store_true', help='latency mode: ranks 2i / 2i+1 compute the uncond / cond half of ONE image '
                    '(exchange per UNet evaluation); even --gpus only; value counts one image per PAIR')
    ap.add_argument('--per-step-launches', action='store_true', help='drive the sampler loop from Python (one graph replay per UNet '
                    'evaluation + small launches) instead of replaying the whole trajectory as one device graph')
    return ap.parse_args()


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` from a bare shell: start N fresh worker processes (one rank per GPU) with the rendezvous
    environment torch.distributed.run would set, BEFORE this process touches the GPU (it never does); rank 0's JSON line
    goes to the inherited stdout.  Returns the exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        # HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver of this pool only supports dmabuf IPC; with the legacy mode RCCL's
        # intra-node transport set-up (and any CUDA-tensor sharing across processes) fails with `hipIpcGetMemHandle: invalid
        # argument`.  The image exports it already; it is repeated here so that a bare `python bench.py --gpus N` from a shell
        # that lost the variable still gives every rank the setting RCCL needs (an existing value is kept).
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env))
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs):
                break              # a rank that died leaves its peers waiting in a collective
            time.sleep(0.2)
        rc = max(abs(p.poll() or 0) for p in procs if p.poll() is not None)
    finally:
        for p in procs:            # end exactly our own children, by handle
            if p.poll() is None:
                p.terminate()
                rc = rc or 1
        for p in procs:
            p.wait()
    return rc


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: the launcher must start exactly --gpus ranks'
    # functional rehearsal of the N>1 path on a one-GPU box: SDOD_BENCH_SHARE_DEVICE=1 puts every rank on cuda:0 and
    # SDOD_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one device); never used for reported numbers
    share = os.environ.get('SDOD_BENCH_SHARE_DEVICE') == '1'
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('SDOD_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)

    from sdod.amd import engine as E, ops, weights as Wt
    from sdod.amd.pipeline import Txt2Img, broadcast_conditioning, device_latent, initial_latent

    t_setup = time.time()
    torch.set_num_threads(host_threads())
    log(f'rank {rank}/{world}: generating synthetic weights ({host_threads()} host threads)')
    cfg = E.sd14_config(64, 64)
    n = args.images_per_gpu
    tables = {'unet': E.UNet(cfg, 2).param_table(), 'temb': E.Temb(cfg, 1).param_table(),
              'vae': E.VaeDecoder(cfg, 1).param_table(), 'text': E.TextEncoder(cfg, 1).param_table()}
    sds = {k: Wt.synthetic_state_dict(t, seed=1234 + i) for i, (k, t) in enumerate(tables.items())}
    log('uploading weights / building graphs')
    pipe = Txt2Img(state_dicts=sds, images_per_gpu=n, latent_hw=64, device=f'cuda:{dev_index}',
                   use_hip_graph=not args.no_hip_graph, cfg_split=args.cfg_split)
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and (args.sampler, args.sampler_steps) == ('plms', 20)
    if not want_cpu:
        sds = None
    setup_s = time.time() - t_setup

    img_owner = rank // 2 if args.cfg_split else rank       # latency mode: both ranks of a pair work on the same images
    x_T = torch.cat([initial_latent(42, img_owner * n + i) for i in range(n)]).to(device)
    ids_u, ids_c = np.asarray(IDS_UNCOND), np.asarray(IDS_COND)

    n_images_per_step = n * (world // 2 if args.cfg_split else world)
    step_no = [0]

    def one_image():
        # a fresh x_T per timed image, drawn on the device (Philox keyed by (seed 42, global image index): SURVEY 7.2)
        k = step_no[0]; step_no[0] += 1
        x_T = torch.cat([device_latent(42, k * n_images_per_step + img_owner * n + i, device=device) for i in range(n)])
        if rank == 0:
            ctx2 = pipe.encode_tokens(ids_u, ids_c)
        else:
            ctx2 = torch.empty(2, cfg.context_len, cfg.context_dim, dtype=torch.float16, device=device)
        ctx2 = broadcast_conditioning(ctx2, 0)
        if args.per_step_launches or args.no_hip_graph:
            return pipe.generate(ctx2, x_T, steps=args.sampler_steps, guidance=7.5, sampler=args.sampler)
        if args.overlap_decode:
            # experiment: the VAE decode of this image on a side stream under the sampling of the next one (every image is
            # complete when the timed region's closing synchronize returns).  Measured SLOWER than the serial form on MI355X
            # (8.97 vs 9.29 images/s on one box): the decode's big grids take CUs from the latency-bound UNet chain
            return pipe.generate_pipelined(ctx2, x_T, steps=args.sampler_steps, guidance=7.5, sampler=args.sampler)[0]
        return pipe.generate_graphed(ctx2, x_T, steps=args.sampler_steps, guidance=7.5, sampler=args.sampler)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if not (args.per_step_launches or args.no_hip_graph):
        one_image()            # builds (captures) the trajectory graph: part of setup, like tile tuning, also when --warmup 0
        barrier()
        setup_s = time.time() - t_setup
    log(f'setup done in {setup_s:.1f}s; warm-up x{args.warmup}')
    for _ in range(args.warmup):
        img = one_image()
    barrier()
    log(f'timing {args.steps} image(s) per GPU')
    t0 = time.perf_counter()
    for _ in range(args.steps):
        img = one_image()
    barrier()
    elapsed = time.perf_counter() - t0
    rccl = None
    if dist is not None:
        # what the driver needs to confirm the collective really spanned N ranks: every rank's own time and device, gathered
        # with the collective itself (a line with world_size_seen != n_gpus, or fewer per-rank entries, is not an N-GPU run)
        mine = torch.tensor([elapsed, float(dev_index), float(rank)], dtype=torch.float64, device=device)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        rows = [g.cpu().tolist() for g in gathered]
        ver = None
        try:
            ver = '.'.join(str(v) for v in torch.cuda.nccl.version())   # RCCL reports through the NCCL API on ROCm
        except Exception:
            pass
        rccl = {'backend': dist.get_backend(), 'world_size_seen': dist.get_world_size(), 'ranks_gathered': len(rows),
                'nccl_version': ver, 'per_rank_ms': [round(1e3 * r[0] / args.steps, 3) for r in rows],
                'per_rank_device': [int(r[1]) for r in rows], 'shared_device_rehearsal': share,
                'hsa_enable_ipc_mode_legacy': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'),
                'collectives_per_image': 'one broadcast of the text conditioning [2, 77, 768] fp16 (236,544 bytes) from rank 0'}
        assert rccl['world_size_seen'] == world and sorted(int(r[2]) for r in rows) == list(range(world))
    assert img.shape == (n, 512, 512, 3) and img.dtype == torch.uint8
    log(f'timed region: {elapsed:.3f}s for {args.steps} step(s)')

    # ---- per-UNet-step time (batched cond+uncond evaluation + CFG + sampler update), HIP-graph replay, same stream
    ctx2 = pipe.encode_tokens(ids_u, ids_c) if rank == 0 else torch.zeros(2, 77, 768, dtype=torch.float16, device=device)
    pipe._set_context(ctx2)
    temb = pipe.time_embeddings(np.asarray([951.0], np.float32))
    x = x_T.clone()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # one step of the sampler loop as sample_plms runs it from its second step on: UNet replay, then ONE launch for guidance +
    # multistep combination + DDIM update + staging of the next evaluation's inputs (ops.plms_update)
    from sdod.amd.samplers import PlmsSchedule, PLMS_ORDERS
    sch = PlmsSchedule(20)
    coefs, div = PLMS_ORDERS[1]
    old1 = torch.zeros_like(x)

    def one_step():
        pipe.unet.execute(pipe.use_hip_graph, static_unchanged=not pipe._ctx_fresh)
        pipe._ctx_fresh = False
        eps = pipe._exchange_halves(pipe.unet.eps) if pipe.cfg_split else pipe.unet.eps
        return ops.plms_update(eps, x, [old1], coefs, div, sch.coef(10), 7.5, mode=1, stage=(pipe.unet.x, temb[0], pipe.unet.temb))

    ops.stage_unet_inputs(x, pipe.unet.x, temb[0], pipe.unet.temb)
    for _ in range(2):
        one_step()
    reps = 10
    ev0.record()
    for _ in range(reps):
        one_step()
    ev1.record()
    torch.cuda.synchronize()
    unet_step_ms = ev0.elapsed_time(ev1) / reps

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel family: per-launch HIP events over the UNet launch list (eager, same stream)
        log(f'per-UNet-step {unet_step_ms:.3f} ms; profiling the launch list')
        table = pipe.unet.op_table()
        ms = pipe.unet.profile(iters=5)
        # Durations are the kernels' own begin-to-end times (Graph::profile launches every entry through
        # hipExtLaunchKernelGGL with a start / stop event pair: what a profiler reports per dispatch).  One family per KERNEL
        # SYMBOL over everything ONE IMAGE launches -- 21 UNet evaluations + the VAE decode + the text encoder -- i.e. the
        # population behind rocprofv3's per-symbol average for this very command (profiles/*_kernel_stats.csv).
        fam = {}
        per_image = [(pipe.unet, table, ms, 21)]
        for g in (pipe.vae, pipe.text):
            if g is not None:
                per_image.append((g, g.op_table(), g.profile(iters=3), 1))
        for _, tab, tms, reps in per_image:
            for (label, fl, by), t in zip(tab, tms):
                f = fam.setdefault(label, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
                f['ms'] += t * reps; f['flops'] += fl * reps; f['bytes'] += by * reps; f['launches'] += reps
        # per-shape table (label, shape, launches per evaluation, us each, TF/s, GB/s): printed on stderr and carried in
        # the JSON line, so BENCH, the rocprofv3 CSV and the PMC passes can be compared shape by shape
        details = pipe.unet.op_details()
        shapes = {}
        for (label, fl, by), det, t in zip(table, details, ms):
            r = shapes.setdefault((label, det), [0, 0.0, 0.0, 0.0])
            r[0] += 1; r[1] += t; r[2] += fl; r[3] += by
        shape_rows = [[lab, det, r[0], round(1e3 * r[1] / r[0], 2), round(r[2] / (r[1] * 1e-3) / 1e12, 1) if r[2] else None,
                       round(r[3] / (r[1] * 1e-3) / 1e9, 1) if r[3] else None]
                      for (lab, det), r in sorted(shapes.items(), key=lambda kv: -kv[1][1])]
        log('UNet launch list by shape (kernel begin-to-end, per evaluation):\n' + '\n'.join(
            f'  {lab:16s} {det:44s} x{n:<3d} {us:8.2f} us  {str(tf):>7s} TF/s  {str(gb):>8s} GB/s' for lab, det, n, us, tf, gb in shape_rows))
        dom = max(fam, key=lambda k: fam[k]['ms'])
        d = fam[dom]
        total_ms = sum(ms)
        image_kernel_ms = sum(f['ms'] for f in fam.values())
        if d['flops'] > 0:
            ach = d['flops'] / (d['ms'] * 1e-3) / 1e12
            roof = dict(bound='mfma', achieved=round(ach, 2), peak=PEAK_TFLOPS_F16, unit='TFLOP/s', frac=round(ach / PEAK_TFLOPS_F16, 4),
                        traffic=None)
        else:
            ach = d['bytes'] / (d['ms'] * 1e-3) / 1e9
            roof = dict(bound='hbm', achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit='GB/s', frac=round(ach / PEAK_HBM_GBS, 4), traffic=None)
        # HBM bytes per launch of that kernel: rocprofv3 PMC passes cannot run inside this process, so the figure comes from
        # the committed summary of separate `--pmc` passes over the SAME launch list (the shipped tune table fixes the tile
        # picks, tools/run_profile.sh + tools/pmc_summary.py apply the guide's gfx950 corrections); `traffic_source` says so,
        # and the fields stay null when no committed measurement matches the dominant symbol
        try:
            sym, mangled = kernel_symbol(dom)
            pmc_file = os.path.join('profiles', PMC_SUMMARY)
            for k in json.load(open(os.path.join(ROOT, pmc_file)))['kernels']:
                name = k['kernel']
                if (mangled and mangled in name) or (not mangled and sym in name):
                    roof['traffic'] = k['hbm_fetch_bytes_per_launch'] + k['hbm_write_bytes_per_launch']
                    roof['mfma_util_pmc'] = k['mfma_util']
                    roof['traffic_source'] = f'offline: {pmc_file} (separate rocprofv3 --pmc passes, not measured in this run)'
                    break
        except (OSError, ValueError, KeyError, TypeError):
            pass
        roof.update(kernel=dom, kernel_symbol=kernel_symbol(dom)[0],
                    flops_per_launch=round(d['flops'] / d['launches']), algorithmic_bytes_per_launch=round(d['bytes'] / d['launches']),
                    launches_per_image=d['launches'], avg_launch_us=round(1e3 * d['ms'] / d['launches'], 2),
                    share_of_image_kernel_time=round(d['ms'] / image_kernel_ms, 3), image_kernel_ms=round(image_kernel_ms, 2),
                    unet_eval_kernel_ms=round(total_ms, 3), timing='kernel begin-to-end (hipExtLaunchKernelGGL start/stop events), per image: 21 UNet evaluations + VAE decode + text encoder',
                    launches_per_unet_eval_total=len(table), tune_table=pipe.unet.tune_source(),
                    unet_eval_tflops=round(pipe.unet.stats()['flops'] / (unet_step_ms * 1e-3) / 1e12, 1),
                    families={k: dict(ms=round(v['ms'], 3), launches=v['launches'],
                                      tflops=round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 1) if v['flops'] else None,
                                      gbs=round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1) if v['bytes'] else None)
                              for k, v in sorted(fam.items(), key=lambda kv: -kv[1]['ms'])},
                    shapes=shape_rows)

        value = args.steps * n * (world // 2 if args.cfg_split else world) / elapsed
        out = {
            'metric': METRIC, 'value': round(value, 4), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
            'config': {'workload': ('SD v1.4 txt2img 512x512, 20-step PLMS (21 UNet evals, batch 2 = cond+uncond per image), '
                                    'CLIP encode + VAE decode + uint8, guidance 7.5') if (args.sampler, args.sampler_steps) == ('plms', 20) else
                                   f'SD v1.4 txt2img 512x512, {args.sampler_steps}-step {args.sampler.upper()}, CLIP encode + VAE decode + uint8, guidance 7.5',
                       'images_per_gpu': n, 'global_batch': n * (world // 2 if args.cfg_split else world), 'parallelism': (f'dp{world // 2} x cfg-split pairs (1 broadcast + 1 all-gather per UNet evaluation)' if args.cfg_split
                                       else f'dp{world} (image shards, 1 RCCL broadcast)'),
                       'hip_graph': not args.no_hip_graph,
                       'trajectory_graph': not (args.per_step_launches or args.no_hip_graph),
                       'decode_overlaps_next_sampling': bool(args.overlap_decode) and not (args.per_step_launches or args.no_hip_graph)},
            'unet_step_ms': round(unet_step_ms, 3),
            'roofline': roof,
            'setup_s': round(setup_s, 1),
        }
        if rccl is not None:
            out['rccl'] = rccl

        if want_cpu:
            z_gpu = pipe.sample_plms(ctx2, x_T, steps=20, guidance=7.5)
            out.update(cpu_baseline_and_parity(pipe, sds, x_T, ctx2, e_gpu=pipe._eps(x_T, temb[0], 7.5, 1),
                                               img_gpu=pipe.decode(z_gpu, mode=1), z_gpu=z_gpu))
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline_and_parity(pipe, sds, x_T, ctx2, e_gpu, img_gpu, z_gpu):
    """The oracle (CPU fp32 restatement, kind "port") on the host cores, as BASELINE.md section 3 prescribes: one warm-up
    UNet evaluation, ONE timed full image (CLIP encode of both prompts + 20-step PLMS = 21 batch-2 UNet evaluations + VAE
    decode + uint8) and 3 timed single UNet evaluations; also the in-run parity block (first guided eps, final latent,
    uint8 image of that very image)."""
    from oracle import pipeline_oracle as PO, sd_torch as S
    threads = host_threads()
    torch.set_num_threads(threads)
    log(f'cpu baseline: building the oracle ({threads} threads)')
    with torch.device('meta'):
        unet, vae, clip = S.UNetModel(), S.AutoencoderKLDecode(), S.ClipTextModel()
    unet.load_state_dict({**sds['unet'], **sds['temb']}, assign=True)
    vae.load_state_dict(sds['vae'], assign=True)
    clip.load_state_dict(sds['text'], assign=True)
    unet.eval(); vae.eval(); clip.eval()
    c16 = ctx2.float().cpu()                      # the oracle consumes the SAME fp16-rounded conditioning the GPU used
    x = x_T[:1].float().cpu()
    t = torch.tensor([951, 951])
    evals = [0]
    unet.register_forward_hook(lambda *a: (evals.__setitem__(0, evals[0] + 1), log(f'cpu baseline: UNet evaluation {evals[0]} done'))[0])
    with torch.no_grad():
        x2 = torch.cat([x, x]); c2 = torch.cat([c16[0:1], c16[1:2]])
        t0 = time.perf_counter(); e = unet(x2, t, c2); warm = time.perf_counter() - t0
        times = []
        for _ in range(3):
            t0 = time.perf_counter(); e = unet(x2, t, c2); times.append(time.perf_counter() - t0)
        t_unet = float(np.mean(times))
        log(f'cpu baseline: UNet evaluation {t_unet:.2f}s (first {warm:.2f}s); timing one full image')
        t0 = time.perf_counter()
        clip(torch.from_numpy(np.stack([IDS_UNCOND, IDS_COND])))
        t_clip = time.perf_counter() - t0
        z_ref = PO.plms_sample(unet, c16[0:1], c16[1:2], x, steps=20, scale=7.5)
        t_loop = time.perf_counter() - t0 - t_clip
        img_ref = PO.decode_u8(vae, z_ref, mode=1)
        t_img = time.perf_counter() - t0
        log(f'cpu baseline: full image {t_img:.1f}s (CLIP {t_clip:.2f}s, sampler {t_loop:.1f}s, decode {t_img - t_loop - t_clip:.1f}s)')
    e_u, e_c = e.chunk(2)
    e_ref = e_u + 7.5 * (e_c - e_u)
    rel = float(((e_gpu[:1].float().cpu()) - e_ref).norm() / e_ref.norm())
    zg = z_gpu[:1].float().cpu()
    rel_z = float((zg.double() - z_ref.double()).norm() / z_ref.double().norm())
    d = np.abs(img_gpu[:1].cpu().numpy().astype(np.int32) - img_ref.astype(np.int32))
    return {
        'cpu_baseline': {'value': round(1.0 / t_img, 6), 'unit': 'images/s', 'cores': threads, 'kind': 'port',
                         'unet_step_ms': round(1e3 * t_unet, 1), 'image_s': round(t_img, 2),
                         'sample': f'PyTorch-CPU fp32 oracle, same weights/inputs: 1 warm-up UNet evaluation, 3 timed UNet evaluations '
                                   f'(batch 2, 4x64x64; mean {t_unet:.2f}s), ONE timed full image = CLIP encode x2 ({t_clip:.2f}s) + 20-step PLMS, '
                                   f'21 UNet evaluations ({t_loop:.1f}s) + VAE decode 64x64->512x512 + uint8 = {t_img:.1f}s'},
        'parity': {'guided_eps_rel_l2_gpu_f16_vs_cpu_f32': round(rel, 6), 'tolerance': 1e-2,
                   'final_latent_rel_l2': round(rel_z, 6), 'final_latent_tolerance': 2e-2,
                   'uint8_image_max_abs_diff': int(d.max()), 'uint8_within_2_lsb': round(float((d <= 2).mean()), 6)},
    }


if __name__ == '__main__':
    main()
