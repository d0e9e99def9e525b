"""oracle/sd_torch.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

PyTorch-CPU fp32 restatement of the arithmetic the reference executes as opaque QNN graphs
(unet / text_encoder / vae_decoder / temb, csrc/libsdod/src/context.cpp:105, :214-218, :276-278,
:327, :352, :387) and of config 1's CPU reference (`scripts/txt2img.py --plms` of the CompVis `ldm`
fork cited at README.md:23,60-65).  That third-party repo is NOT under /root/reference and is
unpinned (no commit / lockfile); no SD checkpoint exists offline.  So this file restates the PUBLIC
SD v1.x architecture; its structural known answers are the parameter totals
(UNet 859.52 M, VAE decoder 49.49 M, CLIP text 123.06 M) and, for CLIP, agreement with the local
`transformers.CLIPTextModel` class on the same weights.  => "parity unpinned" by reference tests at
this boundary; everything else in oracle/ is pinned to the reference itself.

Module trees and state-dict key names follow CompVis ldm (`model.diffusion_model.*`,
`first_stage_model.*`) and HF CLIP (`cond_stage_model.transformer.*`) so a real sd-v1-4.ckpt would load.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------ UNet
class GroupNorm32(nn.GroupNorm):
    def forward(self, x):
        return super().forward(x.float()).type(x.dtype)


def timestep_embedding(timesteps, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class ResBlock(nn.Module):
    def __init__(self, cin, cout, emb_ch):
        super().__init__()
        self.in_layers = nn.Sequential(GroupNorm32(32, cin), nn.SiLU(), nn.Conv2d(cin, cout, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_ch, cout))
        self.out_layers = nn.Sequential(GroupNorm32(32, cout), nn.SiLU(), nn.Dropout(0.0), nn.Conv2d(cout, cout, 3, padding=1))
        self.skip_connection = nn.Identity() if cin == cout else nn.Conv2d(cin, cout, 1)

    def forward(self, x, emb, context=None):
        h = self.in_layers(x)
        h = h + self.emb_layers(emb)[:, :, None, None]
        h = self.out_layers(h)
        return self.skip_connection(x) + h


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim, heads, dim_head):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(0.0))

    def forward(self, x, context=None):
        context = x if context is None else context
        b, n, _ = x.shape
        h = self.heads
        q, k, v = self.to_q(x), self.to_k(context), self.to_v(context)
        q, k, v = (t.reshape(b, t.shape[1], h, -1).transpose(1, 2) for t in (q, k, v))
        sim = torch.einsum('bhid,bhjd->bhij', q, k) * self.scale
        attn = sim.softmax(dim=-1)
        out = torch.einsum('bhij,bhjd->bhid', attn, v)
        out = out.transpose(1, 2).reshape(b, n, -1)
        return self.to_out(out)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        x, gate = self.proj(x).chunk(2, dim=-1)
        return x * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.Sequential(GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim))

    def forward(self, x):
        return self.net(x)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, d_head, context_dim):
        super().__init__()
        self.attn1 = CrossAttention(dim, dim, heads, d_head)
        self.ff = FeedForward(dim)
        self.attn2 = CrossAttention(dim, context_dim, heads, d_head)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)

    def forward(self, x, context):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context) + x
        return self.ff(self.norm3(x)) + x


class SpatialTransformer(nn.Module):
    """use_linear=False: SD1.x (1x1 conv projections); True: SD2.x `use_linear_in_transformer` (Linear on the token view)"""

    def __init__(self, ch, heads, d_head, context_dim, use_linear=False):
        super().__init__()
        self.use_linear = use_linear
        self.norm = nn.GroupNorm(32, ch, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(ch, heads * d_head) if use_linear else nn.Conv2d(ch, heads * d_head, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(heads * d_head, heads, d_head, context_dim)])
        self.proj_out = nn.Linear(heads * d_head, ch) if use_linear else nn.Conv2d(heads * d_head, ch, 1)

    def forward(self, x, emb=None, context=None):
        b, c, h, w = x.shape
        x_in = x
        x = self.norm(x)
        if not self.use_linear:
            x = self.proj_in(x)
        x = x.permute(0, 2, 3, 1).reshape(b, h * w, -1)
        if self.use_linear:
            x = self.proj_in(x)
        for blk in self.transformer_blocks:
            x = blk(x, context)
        if self.use_linear:
            x = self.proj_out(x)
        x = x.reshape(b, h, w, -1).permute(0, 3, 1, 2)
        if not self.use_linear:
            x = self.proj_out(x)
        return x + x_in


class Downsample(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.op = nn.Conv2d(ch, ch, 3, stride=2, padding=1)

    def forward(self, x, emb=None, context=None):
        return self.op(x)


class Upsample(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x, emb=None, context=None):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode='nearest'))


class TimestepSeq(nn.Sequential):
    def forward(self, x, emb=None, context=None):
        for layer in self:
            x = layer(x, emb, context) if not isinstance(layer, nn.Conv2d) else layer(x)
        return x


class UNetModel(nn.Module):
    """ldm UNetModel(in 4, out 4, model_channels 320, attention at ds 1/2/4, 2 res blocks, mult 1-2-4-4,
    8 heads, transformer depth 1, context 768)."""

    def __init__(self, in_ch=4, out_ch=4, model_ch=320, mult=(1, 2, 4, 4), num_res=2, attn_ds=(1, 2, 4), heads=8,
                 context_dim=768, head_dim=None, use_linear=False):
        """SD2.x (public ldm v2 config): head_dim=64 (heads = ch // 64), context_dim=1024, use_linear=True"""
        super().__init__()
        st = lambda ch: SpatialTransformer(ch, ch // head_dim if head_dim else heads, head_dim if head_dim else ch // heads,
                                           context_dim, use_linear)
        emb_ch = model_ch * 4
        self.model_ch = model_ch
        self.time_embed = nn.Sequential(nn.Linear(model_ch, emb_ch), nn.SiLU(), nn.Linear(emb_ch, emb_ch))
        self.input_blocks = nn.ModuleList([TimestepSeq(nn.Conv2d(in_ch, model_ch, 3, padding=1))])
        chans = [model_ch]
        ch, ds = model_ch, 1
        for level, m in enumerate(mult):
            for _ in range(num_res):
                layers = [ResBlock(ch, m * model_ch, emb_ch)]
                ch = m * model_ch
                if ds in attn_ds:
                    layers.append(st(ch))
                self.input_blocks.append(TimestepSeq(*layers))
                chans.append(ch)
            if level != len(mult) - 1:
                self.input_blocks.append(TimestepSeq(Downsample(ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepSeq(ResBlock(ch, ch, emb_ch), st(ch), ResBlock(ch, ch, emb_ch))
        self.output_blocks = nn.ModuleList()
        for level, m in list(enumerate(mult))[::-1]:
            for i in range(num_res + 1):
                layers = [ResBlock(ch + chans.pop(), m * model_ch, emb_ch)]
                ch = m * model_ch
                if ds in attn_ds:
                    layers.append(st(ch))
                if level and i == num_res:
                    layers.append(Upsample(ch))
                    ds //= 2
                self.output_blocks.append(TimestepSeq(*layers))
        self.out = nn.Sequential(GroupNorm32(32, ch), nn.SiLU(), nn.Conv2d(model_ch, out_ch, 3, padding=1))

    def forward(self, x, timesteps, context):
        emb = self.time_embed(timestep_embedding(timesteps, self.model_ch))
        hs = []
        h = x
        for m in self.input_blocks:
            h = m(h, emb, context)
            hs.append(h)
        h = self.middle_block(h, emb, context)
        for m in self.output_blocks:
            h = torch.cat([h, hs.pop()], dim=1)
            h = m(h, emb, context)
        return self.out(h)


# ------------------------------------------------------------------------------------------ VAE decoder
def _norm(ch):
    return nn.GroupNorm(32, ch, eps=1e-6, affine=True)


class VaeResnetBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1 = _norm(cin)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = _norm(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.nin_shortcut = nn.Conv2d(cin, cout, 1)
        self.cin, self.cout = cin, cout

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        if self.cin != self.cout:
            x = self.nin_shortcut(x)
        return x + h


class VaeAttnBlock(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.norm = _norm(ch)
        self.q = nn.Conv2d(ch, ch, 1)
        self.k = nn.Conv2d(ch, ch, 1)
        self.v = nn.Conv2d(ch, ch, 1)
        self.proj_out = nn.Conv2d(ch, ch, 1)

    def forward(self, x):
        h_ = self.norm(x)
        q, k, v = self.q(h_), self.k(h_), self.v(h_)
        b, c, h, w = q.shape
        q = q.reshape(b, c, h * w).permute(0, 2, 1)
        k = k.reshape(b, c, h * w)
        w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
        w_ = F.softmax(w_, dim=2)
        v = v.reshape(b, c, h * w)
        h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, h, w)
        return x + self.proj_out(h_)


class VaeUpsample(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode='nearest'))


class VaeDecoder(nn.Module):
    def __init__(self, ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=4):
        super().__init__()
        self.num_levels = len(ch_mult)
        block_in = ch * ch_mult[-1]
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = VaeResnetBlock(block_in, block_in)
        self.mid.attn_1 = VaeAttnBlock(block_in)
        self.mid.block_2 = VaeResnetBlock(block_in, block_in)
        ups = []
        for i_level in reversed(range(self.num_levels)):
            block_out = ch * ch_mult[i_level]
            up = nn.Module()
            up.block = nn.ModuleList()
            for _ in range(num_res_blocks + 1):
                up.block.append(VaeResnetBlock(block_in, block_out))
                block_in = block_out
            if i_level != 0:
                up.upsample = VaeUpsample(block_in)
            ups.insert(0, up)
        self.up = nn.ModuleList(ups)
        self.norm_out = _norm(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, 3, padding=1)

    def forward(self, z):
        h = self.conv_in(z)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        for i_level in reversed(range(self.num_levels)):
            for blk in self.up[i_level].block:
                h = blk(h)
            if i_level != 0:
                h = self.up[i_level].upsample(h)
        return self.conv_out(F.silu(self.norm_out(h)))


class AutoencoderKLDecode(nn.Module):
    """`first_stage_model.{post_quant_conv, decoder}`; decode(z) = decoder(post_quant_conv(z / 0.18215))."""
    scale_factor = 0.18215

    def __init__(self):
        super().__init__()
        self.post_quant_conv = nn.Conv2d(4, 4, 1)
        self.decoder = VaeDecoder()

    def forward(self, z):
        return self.decoder(self.post_quant_conv(z * (1.0 / self.scale_factor)))


# -------------------------------------------------------------------------------------- CLIP text model
class ClipLayer(nn.Module):
    def __init__(self, d, heads, inter):
        super().__init__()
        self.self_attn = nn.Module()
        for n in ('q_proj', 'k_proj', 'v_proj', 'out_proj'):
            setattr(self.self_attn, n, nn.Linear(d, d))
        self.layer_norm1 = nn.LayerNorm(d)
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(d, inter)
        self.mlp.fc2 = nn.Linear(inter, d)
        self.layer_norm2 = nn.LayerNorm(d)
        self.heads = heads

    def forward(self, x, mask):
        b, n, d = x.shape
        h = self.heads
        r = x
        y = self.layer_norm1(x)
        q, k, v = self.self_attn.q_proj(y), self.self_attn.k_proj(y), self.self_attn.v_proj(y)
        q, k, v = (t.reshape(b, n, h, d // h).transpose(1, 2) for t in (q, k, v))
        att = (q @ k.transpose(-1, -2)) * (d // h) ** -0.5 + mask
        y = (att.softmax(-1) @ v).transpose(1, 2).reshape(b, n, d)
        x = r + self.self_attn.out_proj(y)
        y = self.mlp.fc1(self.layer_norm2(x))
        y = y * torch.sigmoid(1.702 * y)  # quick_gelu
        return x + self.mlp.fc2(y)


class ClipTextModel(nn.Module):
    """CLIP ViT-L/14 text transformer; keys as HF `CLIPTextModel` (text_model.*); output last_hidden_state."""

    def __init__(self, vocab=49408, d=768, layers=12, heads=12, inter=3072, max_pos=77):
        super().__init__()
        tm = nn.Module()
        tm.embeddings = nn.Module()
        tm.embeddings.token_embedding = nn.Embedding(vocab, d)
        tm.embeddings.position_embedding = nn.Embedding(max_pos, d)
        tm.encoder = nn.Module()
        tm.encoder.layers = nn.ModuleList([ClipLayer(d, heads, inter) for _ in range(layers)])
        tm.final_layer_norm = nn.LayerNorm(d)
        self.text_model = tm

    def forward(self, ids):
        tm = self.text_model
        n = ids.shape[1]
        x = tm.embeddings.token_embedding(ids) + tm.embeddings.position_embedding(torch.arange(n))[None]
        mask = torch.full((n, n), float('-inf')).triu(1)
        for layer in tm.encoder.layers:
            x = layer(x, mask)
        return tm.final_layer_norm(x)


class OpenClipBlock(nn.Module):
    """open_clip ResidualAttentionBlock (text tower): pre-LN, nn.MultiheadAttention (fused in_proj), erf-GELU MLP"""

    def __init__(self, d, heads):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d)
        self.attn = nn.MultiheadAttention(d, heads, batch_first=True)
        self.ln_2 = nn.LayerNorm(d)
        self.mlp = nn.Sequential()
        self.mlp.add_module('c_fc', nn.Linear(d, 4 * d))
        self.mlp.add_module('gelu', nn.GELU())
        self.mlp.add_module('c_proj', nn.Linear(4 * d, d))

    def forward(self, x, mask):
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False, attn_mask=mask)[0]
        return x + self.mlp(self.ln_2(x))


class OpenClipTextModel(nn.Module):
    """OpenCLIP ViT-H/14 text tower as SD2.x uses it (public ldm v2 `FrozenOpenCLIPEmbedder`, layer='penultimate'): token +
    positional embedding, the first `run_layers` of the checkpoint's 24 causal blocks, then ln_final; no text projection.
    Keys as open_clip: token_embedding.weight, positional_embedding, transformer.resblocks.N.*, ln_final.*"""

    def __init__(self, vocab=49408, d=1024, layers=24, heads=16, max_pos=77, run_layers=23):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, d)
        self.positional_embedding = nn.Parameter(torch.empty(max_pos, d))
        self.transformer = nn.Module()
        self.transformer.resblocks = nn.ModuleList([OpenClipBlock(d, heads) for _ in range(layers)])
        self.ln_final = nn.LayerNorm(d)
        self.run_layers = run_layers

    def forward(self, ids):
        n = ids.shape[1]
        x = self.token_embedding(ids) + self.positional_embedding[:n]
        mask = torch.full((n, n), float('-inf')).triu(1)
        for blk in self.transformer.resblocks[:self.run_layers]:
            x = blk(x, mask)
        return self.ln_final(x)


# ------------------------------------------------------------------------------------ synthetic weights
def synthetic_init_(module, seed=1234):
    """Deterministic seeded parameters: conv/linear weight ~ N(0, 1/fan_in), norm weight
    1 + 0.1 N(0,1), every bias 0.1 N(0,1), embeddings 0.02 N(0,1).  One generator, state-dict order."""
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if 'embedding' in name:
                p.copy_(0.02 * torch.randn(p.shape, generator=gen))
            elif p.dim() >= 2:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=gen) * fan_in ** -0.5)
            elif name.endswith('weight'):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=gen))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=gen))
    return module


def count_params(module):
    return sum(p.numel() for p in module.parameters())


def build(cls, *args, seed=1234, **kwargs):
    """Construct on the meta device (skips nn's default init of ~1e9 parameters), then fill with the seeded
    synthetic weights."""
    with torch.device('meta'):
        m = cls(*args, **kwargs)
    m = m.to_empty(device='cpu')
    return synthetic_init_(m, seed).eval()
