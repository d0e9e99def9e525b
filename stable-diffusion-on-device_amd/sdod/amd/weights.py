"""Weight handling for the engine graphs: the `.sdodw` container that libsdod_setup loads from models_dir (the MI355X
stand-in for the serialized QNN blobs of context.cpp:105-115) and the deterministic synthetic initialisation used when
no checkpoint is available (there is none offline)."""
import struct

import numpy as np
import torch

MAGIC = b'SDODW001'


class QuantU8:
    """per-tensor affine uint8 weight, the reference's QNN weight format (`quantize=8`, todlc.py:108): real = (q + offset)
    * scale with q unsigned and offset <= 0 (qnn_context.cpp:1018-1033)"""

    def __init__(self, q, scale, offset):
        assert q.dtype == torch.uint8
        self.q, self.scale, self.offset = q.contiguous(), float(scale), int(offset)

    @property
    def shape(self):
        return self.q.shape

    def dequantize(self):
        """the reference's arithmetic: (q + offset) * scale in double, rounded to float"""
        return ((self.q.to(torch.float64) + self.offset) * float(np.float32(self.scale))).to(torch.float32)

    def payload(self):
        return struct.pack('<fi', self.scale, self.offset) + self.q.numpy().tobytes()


def quantize_u8(t):
    """min/max affine quantisation of one tensor to uint8 (what a QNN converter run with 8-bit weights produces)"""
    t = t.detach().float()
    lo, hi = min(float(t.min()), 0.0), max(float(t.max()), 0.0)          # the range always contains 0 (offset <= 0)
    scale = float(np.float32((hi - lo) / 255.0)) if hi > lo else 1.0
    offset = int(round(lo / scale))
    q = torch.clamp(torch.round(t / scale) - offset, 0, 255).to(torch.uint8)
    return QuantU8(q, scale, offset)


def quantize_state_dict(sd, min_ndim=2):
    """conv / linear weights (ndim >= 2, embeddings included) -> QuantU8; biases and norm parameters stay as they are"""
    return {k: (quantize_u8(v) if v.dim() >= min_ndim else v) for k, v in sd.items()}


def save(path, tensors):
    """tensors: mapping name -> torch tensor (fp32 or fp16, canonical PyTorch layout).  Layout of the file:
    MAGIC, u64 count, per tensor {u32 name_len, name, u32 dtype (0 f16, 1 f32), u32 ndim, u64 dims[ndim], u64 offset,
    u64 nbytes}, then 64-byte aligned payloads (read by Graph::load_file, csrc/engine.hip)."""
    items = []
    header = 16
    for name, t in tensors.items():
        if not isinstance(t, QuantU8):
            t = t.detach().cpu().contiguous()
            if t.dtype not in (torch.float16, torch.float32):
                t = t.float()
        items.append((name.encode(), t))
        header += 4 + len(name.encode()) + 8 + 8 * len(t.shape) + 16
    pos = (header + 63) // 64 * 64
    with open(path, 'wb') as f:
        f.write(MAGIC)
        f.write(struct.pack('<Q', len(items)))
        offsets = []
        for nb, t in items:
            quant = isinstance(t, QuantU8)
            nbytes = 8 + t.q.numel() if quant else t.numel() * t.element_size()
            f.write(struct.pack('<I', len(nb))); f.write(nb)
            f.write(struct.pack('<II', 2 if quant else (1 if t.dtype == torch.float32 else 0), len(t.shape)))
            for d in t.shape:
                f.write(struct.pack('<Q', d))
            f.write(struct.pack('<QQ', pos, nbytes))
            offsets.append(pos)
            pos = (pos + nbytes + 63) // 64 * 64
        for (nb, t), off in zip(items, offsets):
            f.seek(off)
            f.write(t.payload() if isinstance(t, QuantU8) else t.numpy().tobytes())


def load(path):
    """Inverse of save(); returns {name: tensor or QuantU8}."""
    out = {}
    with open(path, 'rb') as f:
        data = f.read()
    assert data[:8] == MAGIC, 'bad magic'
    count = struct.unpack_from('<Q', data, 8)[0]
    pos = 16
    for _ in range(count):
        nl = struct.unpack_from('<I', data, pos)[0]; pos += 4
        name = data[pos:pos + nl].decode(); pos += nl
        dt, nd = struct.unpack_from('<II', data, pos); pos += 8
        dims = struct.unpack_from('<' + 'Q' * nd, data, pos); pos += 8 * nd
        off, nb = struct.unpack_from('<QQ', data, pos); pos += 16
        if dt == 2:
            scale, offset = struct.unpack_from('<fi', data, off)
            q = np.frombuffer(data, dtype=np.uint8, count=nb - 8, offset=off + 8)
            out[name] = QuantU8(torch.from_numpy(q.copy()).reshape(dims), scale, offset)
            continue
        arr = np.frombuffer(data, dtype=np.float32 if dt == 1 else np.float16, count=nb // (4 if dt == 1 else 2), offset=off)
        out[name] = torch.from_numpy(arr.copy()).reshape(dims)
    return out


def synthetic_state_dict(table, seed=1234, dtype=torch.float32):
    """Deterministic seeded parameters for a graph's parameter table [(name, shape), ...]:
    conv / linear weight ~ N(0, 1/fan_in), norm weight 1 + 0.1 N(0,1), every bias 0.1 N(0,1), embeddings 0.02 N(0,1).
    One generator, table order.  The CPU oracle loads the same tensors, so both sides compute on identical weights."""
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in table:
        shape = tuple(int(d) for d in shape)
        x = torch.randn(shape, generator=gen)
        if 'embedding' in name:
            x.mul_(0.02)
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            x.mul_(fan_in ** -0.5)
        elif name.endswith('weight'):
            x.mul_(0.1).add_(1.0)
        else:
            x.mul_(0.1)
        sd[name] = x.to(dtype)
    return sd
