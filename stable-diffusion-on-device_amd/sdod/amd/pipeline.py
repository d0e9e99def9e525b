"""txt2img on MI355X: Python host loop over the engine graphs (north_star: "Python host code on PyTorch-ROCm drives
the PLMS/DPM sampler loop while the UNet denoising step ... and the VAE decoder run as hand-written CDNA4 HIP kernels").

Mirrors the reference's Context (context.cpp:49-403) in structure: setup -> graphs + cached unconditional embedding +
cached time embeddings; generate -> tokenise, text-encode, sampler loop over the batched (uncond, cond) UNet
evaluation with CFG, decode, uint8.  Multi-GPU: one process per GPU, images sharded over ranks, the text conditioning
computed on rank 0 and sent with ONE RCCL broadcast."""
import numpy as np
import torch

from . import engine as E
from . import ops
from .host import DpmSolver
from .samplers import PLMS_ORDERS, PlmsSchedule


class Txt2Img:
    def __init__(self, state_dicts=None, models_dir=None, images_per_gpu=1, latent_hw=64, device='cuda:0', use_hip_graph=True,
                 tokenizer=None, with_text_encoder=True, model='sd14', with_vae=True, cfg_split=False, weight_quant=None):
        """state_dicts: {'unet': sd, 'temb': sd, 'text': sd, 'vae': sd} in ldm/HF naming (canonical layouts; values may be
        weights.QuantU8 for an int8-weight checkpoint), or models_dir with the .sdodw containers libsdod_setup uses.
        model='sd21': SD v2.1-768 (BASELINE config 5): UNet with 64-wide heads / context 1024, v-prediction, OpenCLIP
        ViT-H/14 text tower (open_clip key names, penultimate block + ln_final, prompts padded with id 0 after EOT)."""
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.n = images_per_gpu
        self.model = model
        self.v_prediction = model == 'sd21'
        self.cfg = E.sd21_config(latent_hw, latent_hw) if model == 'sd21' else E.sd14_config(latent_hw, latent_hw)
        # int8 weight streaming (BASELINE config 5): when the UNet checkpoint holds affine-uint8 tensors (weights.QuantU8, the
        # reference's QNN encoding) they stay uint8 in HBM and the GEMMs expand them on the fly; weight_quant=False keeps the
        # round-1 behaviour (dequantise once at load, fp16 in HBM)
        if weight_quant is None:
            weight_quant = state_dicts is not None and any(hasattr(v, 'payload') and len(v.shape) >= 2 for v in state_dicts['unet'].values())
        # weight_quant='auto' (or 2): stream the codes only where that is not slower than fp16 (sdod_model_config.weight_quant = 2)
        self.cfg.weight_quant = 2 if weight_quant in ('auto', 2) else 1 if weight_quant else 0
        # latency mode: the two halves of the classifier-free-guidance batch run on TWO GPUs -- even rank =
        # unconditional, odd rank = conditional -- and exchange their [n,H,W,4] fp16 predictions once per UNet evaluation
        # (32 KB per image over xGMI); everything after the exchange is computed redundantly, so both ranks hold the image
        self.cfg_split = bool(cfg_split)
        self._pair = None
        if self.cfg_split:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() % 2 == 0):
                raise RuntimeError('cfg_split needs torch.distributed initialised with an even world size')
            rank = dist.get_rank()
            self._half = rank % 2
            for a in range(0, dist.get_world_size(), 2):           # every rank creates every pair group, in the same order
                grp = dist.new_group([a, a + 1])
                if a == rank - self._half:
                    self._pair = grp
            self._pair_staged = dist.get_backend() == 'gloo'         # gloo gathers on the host (rehearsals); RCCL on device
        self.use_hip_graph = use_hip_graph
        self.tokenizer = tokenizer
        self.unet = E.UNet(self.cfg, self.n if cfg_split else 2 * self.n, device)
        self.vae = E.VaeDecoder(self.cfg, 1, device) if with_vae else None
        self.text = E.TextEncoder(self.cfg, 2, device) if with_text_encoder else None
        self._temb_graphs = {}
        self._sd = state_dicts
        self._dir = models_dir
        for g, key, stem in ((self.unet, 'unet', 'unet'), (self.vae, 'vae', 'vae_decoder'), (self.text, 'text', 'text_encoder')):
            if g is None:
                continue
            self._load(g, key, stem)
            g.finalize()
        self._temb_cache = {}
        self._ctx_fresh = True

    def _load(self, g, key, stem):
        if self._sd is not None:
            g.load_state_dict(self._sd[key])
        else:
            g.load_file(f'{self._dir}/{stem}.sdodw')

    # ------------------------------------------------------------------ conditioning
    def encode_tokens(self, ids_uncond, ids_cond):
        """ids: int arrays [77]; returns fp16 [2, 77, 768] = (uncond, cond), computed on this GPU"""
        ids = torch.from_numpy(np.stack([np.asarray(ids_uncond), np.asarray(ids_cond)]).astype(np.int32))
        self.text.ids.copy_(ids)
        self.text.execute(self.use_hip_graph)
        return self.text.out.clone()

    def encode_prompt(self, prompt, negative=''):
        return self.encode_tokens(self._ids(negative), self._ids(prompt))

    def _ids(self, text):
        """token ids [77]: SOT, tokens, EOT, padding.  CLIP (SD1.x) pads with EOT, as the reference's tokenizer does
        (tokenizer.cpp:274-275); open_clip's tokenizer (SD2.x) pads with 0 -- the padded positions are part of the context"""
        ids = np.array(self.tokenizer.encode(text), dtype=np.int64)
        if self.model == 'sd21':
            eot = int(ids.max())                       # EOT is the largest id of the vocabulary
            first = int(np.argmax(ids == eot))
            ids[first + 1:] = 0
        return ids

    def time_embeddings(self, times):
        """[len(times), E] fp16: time-MLP output projected for every ResBlock, for model times `times` (cached per
        schedule, as context.cpp:257-278 does)"""
        key = tuple(float(t) for t in times)
        if key not in self._temb_cache:
            g = self._temb_graphs.get(len(key))
            if g is None:
                g = E.Temb(self.cfg, len(key), self.device)
                self._load(g, 'temb', 'temb')
                g.finalize()
                self._temb_graphs[len(key)] = g
            g.t.copy_(torch.tensor(key, dtype=torch.float32))
            g.execute()
            self._temb_cache[key] = g.out.clone()
        return self._temb_cache[key]

    # ------------------------------------------------------------------ one guided eps evaluation
    def _set_context(self, ctx2):
        n = self.n
        if self.cfg_split:
            self.unet.ctx.copy_(ctx2[self._half:self._half + 1].expand(n, -1, -1))
        else:
            self.unet.ctx[:n].copy_(ctx2[0:1].expand(n, -1, -1))
            self.unet.ctx[n:].copy_(ctx2[1:2].expand(n, -1, -1))
        self._ctx_fresh = True     # the next UNet execute must redo the cross-attention K/V projections

    def _exchange_halves(self, mine):
        """[n,H,W,4] fp16 of this rank -> [2n,H,W,4] = (uncond rows ; cond rows), identical on both ranks of the pair"""
        import torch.distributed as dist
        both = torch.empty((2 * self.n,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
        if self._pair_staged:
            host = torch.empty(both.shape, dtype=mine.dtype)
            dist.all_gather_into_tensor(host, mine.cpu().contiguous(), group=self._pair)
            both.copy_(host)
        else:
            dist.all_gather_into_tensor(both, mine.contiguous(), group=self._pair)
        return both

    def _eps(self, x, temb_row, guidance, mode, v_coef=None):
        """x: fp32 [n,4,H,W]; returns guided eps fp32 [n,4,H,W].  Batch rows: [uncond x n ; cond x n] (ldm order).
        v_coef = (sqrt(abar_t), sqrt(1 - abar_t)) for a v-prediction model: the guided output is v, eps follows from it."""
        # one in-tree launch stages the graph inputs: x repeated for the (uncond, cond) halves, the time row for every batch row
        ops.stage_unet_inputs(x, self.unet.x, temb_row, self.unet.temb)
        self.unet.execute(self.use_hip_graph, static_unchanged=not self._ctx_fresh)
        self._ctx_fresh = False
        eps = self._exchange_halves(self.unet.eps) if self.cfg_split else self.unet.eps
        out = ops.cfg_combine(eps, guidance, uncond_first=True, mode=mode)
        if v_coef is not None:
            out = ops.lincomb4([out, x], [v_coef[0], v_coef[1]], 1.0)
        return out

    # ------------------------------------------------------------------ samplers
    def sample_plms(self, ctx2, x_T, steps=20, guidance=7.5, trace=None):
        sch = PlmsSchedule(steps)
        temb = self.time_embeddings(sch.timesteps.astype(np.float32))     # row k <-> timestep index k
        self._set_context(ctx2)
        x = x_T.to(self.device, torch.float32).clone()
        old = []
        staged = False   # the previous step's fused update already staged this evaluation's inputs
        n_steps = len(sch.time_range)
        for i, step in enumerate(sch.time_range):
            index = sch.steps - i - 1
            vc = sch.v_to_eps_coef(index) if self.v_prediction else None
            if len(old) > 0:
                # steps 2..: stage (unless done) -> UNet -> ONE launch for guidance, the multistep combination, the DDIM update
                # and the staging of the next evaluation's inputs (ops.plms_update = the four separate launches, bit for bit)
                if not staged:
                    ops.stage_unet_inputs(x, self.unet.x, temb[index], self.unet.temb)
                self.unet.execute(self.use_hip_graph, static_unchanged=not self._ctx_fresh)
                self._ctx_fresh = False
                eps = self._exchange_halves(self.unet.eps) if self.cfg_split else self.unet.eps
                k = min(len(old), 3)
                coefs, div = PLMS_ORDERS[k]
                nxt_stage = (self.unet.x, temb[index - 1], self.unet.temb) if i + 1 < n_steps else None
                e_t = ops.plms_update(eps, x, old[::-1][:k], coefs, div, sch.coef(index), guidance, mode=1, v_coef=vc, stage=nxt_stage)
                staged = nxt_stage is not None
                old.append(e_t)
                old = old[-3:]
                if trace is not None:
                    trace.append((int(step), index))
                continue
            # first step (pseudo improved Euler, two evaluations): the separate launches
            e_t = self._eps(x, temb[index], guidance, mode=1, v_coef=vc)
            x_pred = x.clone()
            ops.ddim_step(x_pred, e_t, **sch.coef(index))
            nxt = max(index - 1, 0)
            e_next = self._eps(x_pred, temb[nxt], guidance, mode=1, v_coef=sch.v_to_eps_coef(nxt) if self.v_prediction else None)
            e_prime = ops.lincomb4([e_t, e_next], [1.0, 1.0], 2.0)
            ops.ddim_step(x, e_prime, **sch.coef(index))
            old.append(e_t)
            old = old[-3:]
            if trace is not None:
                trace.append((int(step), index))
        return x

    def sample_dpm(self, ctx2, x_T, steps=20, guidance=7.5):
        """the reference driver's sampler: DPM-Solver++(2M), CFG as g*e_c + (1-g)*e_u (context.cpp:342-382)"""
        solver = DpmSolver()
        model_ts = solver.prepare(steps)
        temb = self.time_embeddings(model_ts[:steps])
        self._set_context(ctx2)
        x = x_T.to(self.device, torch.float32).clone()
        y_prev = torch.zeros_like(x)
        # one staging launch in front of the loop; every step is then the UNet replay + ONE launch (guidance, solver update and the
        # staging of the next step's inputs: ops.dpm_step = cfg_combine + dpm_update + stage_unet_inputs, bit for bit)
        ops.stage_unet_inputs(x, self.unet.x, temb[0], self.unet.temb)
        for s in range(steps):
            self.unet.execute(self.use_hip_graph, static_unchanged=not self._ctx_fresh)
            self._ctx_fresh = False
            eps = self._exchange_halves(self.unet.eps) if self.cfg_split else self.unet.eps
            ops.dpm_step(eps, x, y_prev, solver.coef(s), guidance, mode=0,
                         stage=(self.unet.x, temb[s + 1], self.unet.temb) if s + 1 < steps else None)
        return x

    # ------------------------------------------------------------------ decode
    def decode(self, latents, mode=1):
        """latents fp32 [n,4,H,W] -> uint8 [n, 8H, 8W, 3] (mode 1 = ldm's 255*clamp((x+1)/2,0,1); mode 0 = reference driver)"""
        outs = []
        for i in range(latents.shape[0]):
            self.vae.z.copy_(latents[i:i + 1])
            self.vae.execute(self.use_hip_graph)
            outs.append(ops.image_to_u8(self.vae.img, 0.5, 0.5, mode))
        return torch.cat(outs, 0)

    def generate(self, ctx2, x_T, steps=20, guidance=7.5, sampler='plms'):
        z = self.sample_plms(ctx2, x_T, steps, guidance) if sampler == 'plms' else self.sample_dpm(ctx2, x_T, steps, guidance)
        return self.decode(z, mode=1 if sampler == 'plms' else 0)

    def generate_graphed(self, ctx2, x_T, steps=20, guidance=7.5, sampler='plms'):
        """generate() with the WHOLE trajectory -- context upload, every UNet evaluation, CFG, sampler updates, VAE decode,
        uint8 -- replayed as ONE device graph: the host enqueues a single launch per image instead of ~9 small launches per
        step, so the GPU never waits for Python between steps (2-3 ms per image at 20 steps).  The sequence is static for a
        given (sampler, steps, guidance, batch): it is captured once from the ordinary eager code path (so it is the same
        kernels on the same buffers, bit for bit) and cached; ctx2 / x_T are copied into the graph's static inputs."""
        if self.cfg_split:                 # a collective per evaluation cannot live inside one captured graph
            return self.generate(ctx2, x_T, steps, guidance, sampler)
        key = (sampler, int(steps), float(guidance), tuple(x_T.shape))
        cache = self.__dict__.setdefault('_traj', {})
        if key not in cache:
            s_ctx = torch.empty_like(ctx2, device=self.device)
            s_x = torch.empty(tuple(x_T.shape), dtype=torch.float32, device=self.device)
            s_ctx.copy_(ctx2); s_x.copy_(x_T)
            keep = self.use_hip_graph
            self.use_hip_graph = False          # inside a capture the UNet runs its launch list, not its own graph
            try:
                self.generate(s_ctx, s_x, steps, guidance, sampler)          # warm-up: kernel attributes, time embeddings, tuning
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                # thread_local: another thread of the process (a collective backend's watchdog) may touch the runtime meanwhile
                with torch.cuda.graph(g, capture_error_mode='thread_local'):
                    out = self.generate(s_ctx, s_x, steps, guidance, sampler)
            finally:
                self.use_hip_graph = keep
            cache[key] = (g, s_ctx, s_x, out)
        g, s_ctx, s_x, out = cache[key]
        s_ctx.copy_(ctx2); s_x.copy_(x_T)
        g.replay()
        return out

    def generate_pipelined(self, ctx2, x_T, steps=20, guidance=7.5, sampler='plms'):
        """generate_graphed() as TWO device graphs -- sampling (context upload, every UNet evaluation, CFG, sampler updates) on
        the current stream and decoding (VAE + uint8) on a side stream -- so that the decode of image i runs while image i+1 is
        being sampled.  EXPERIMENT, not the default: the guided UNet chain is latency-bound (batch 1 takes 82 % of the time of
        batch 2, tools/two_chain_probe.py), but on MI355X the decode's big grids take more from that chain than the overlap
        gives back (bench.py --overlap-decode: 8.97 vs 9.29 images/s serial, same box).  Returns (uint8 images, event): the
        images are valid once the event has completed and until the decode of the NEXT call starts."""
        if self.cfg_split:
            out = self.generate(ctx2, x_T, steps, guidance, sampler)
            ev = torch.cuda.Event(); ev.record()
            return out, ev
        key = ('pipelined', sampler, int(steps), float(guidance), tuple(x_T.shape))
        cache = self.__dict__.setdefault('_traj', {})
        mode = 1 if sampler == 'plms' else 0
        if key not in cache:
            s_ctx = torch.empty_like(ctx2, device=self.device)
            s_x = torch.empty(tuple(x_T.shape), dtype=torch.float32, device=self.device)
            s_ctx.copy_(ctx2); s_x.copy_(x_T)
            sample = self.sample_plms if sampler == 'plms' else self.sample_dpm
            side = torch.cuda.Stream(device=self.device)
            keep = self.use_hip_graph
            self.use_hip_graph = False          # inside a capture the graphs run their launch lists
            try:
                z = sample(s_ctx, s_x, steps, guidance)                       # warm-up: kernel attributes, time embeddings, tuning
                self.decode(z, mode=mode)
                torch.cuda.synchronize(self.device)
                g_s = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_s, capture_error_mode='thread_local'):
                    z_s = sample(s_ctx, s_x, steps, guidance)
                z_in = torch.empty_like(z_s)
                g_d = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_d, capture_error_mode='thread_local'):
                    out = self.decode(z_in, mode=mode)
            finally:
                self.use_hip_graph = keep
            cache[key] = dict(g_s=g_s, g_d=g_d, s_ctx=s_ctx, s_x=s_x, z_s=z_s, z_in=z_in, out=out, side=side,
                              copied=None, decoded=None)
        c = cache[key]
        main = torch.cuda.current_stream(self.device)
        if c['copied'] is not None:
            main.wait_event(c['copied'])        # the previous latent has left z_s
        c['s_ctx'].copy_(ctx2); c['s_x'].copy_(x_T)
        c['g_s'].replay()
        sampled = torch.cuda.Event(); sampled.record(main)
        with torch.cuda.stream(c['side']):
            c['side'].wait_event(sampled)
            c['z_in'].copy_(c['z_s'])
            c['copied'] = torch.cuda.Event(); c['copied'].record(c['side'])
            c['g_d'].replay()
            c['decoded'] = torch.cuda.Event(); c['decoded'].record(c['side'])
        return c['out'], c['decoded']


def broadcast_conditioning(ctx2, src=0):
    """the one collective of the path: CLIP output [2,77,768] fp16 (236,544 B) from rank `src` to every rank over RCCL"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.broadcast(ctx2, src=src)
    return ctx2


def shard_images(total, rank, world):
    """block distribution of image indices over ranks"""
    per = (total + world - 1) // world
    lo = min(total, rank * per)
    return list(range(lo, min(total, lo + per)))


def initial_latent(seed, image_index, shape=(4, 64, 64)):
    """x_T for image `image_index`: CPU generator seeded with (seed, index) so any sharding yields the same images"""
    g = torch.Generator().manual_seed(int(seed) * 1000003 + int(image_index))
    return torch.randn((1,) + tuple(shape), generator=g)


def device_latent(seed, image_index, shape=(4, 64, 64), device='cuda:0'):
    """x_T drawn ON the device for throughput runs (SURVEY 7.2 "RNG"; the reference draws on the host, context.cpp:333-334):
    in-tree Philox4x32-10 + Box-Muller, stream = image index, so the latent depends on (seed, image index) only -- any
    sharding of the images over ranks, and any batch slot, yields the same image.  Parity runs inject x_T instead."""
    return ops.randn((1,) + tuple(shape), seed, image_index, torch.device(device))
