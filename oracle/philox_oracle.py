"""oracle/philox_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

CPU restatement (numpy) of the counter-based generator behind `sdod_randn_f32` (csrc/elementwise.hip): Philox4x32-10
(Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11 -- the published algorithm; pinned below
by the known-answer vectors distributed with the authors' Random123 library) followed by Box-Muller.

The reference draws x_T on the host with std::mt19937 + std::normal_distribution (context.cpp:16, :333-334), a stream that
is implementation-defined and not reproducible on a GPU (SURVEY 7.2 "RNG"): parity runs inject x_T; throughput runs use
this generator, keyed by (seed, image index) so that any sharding of the images over ranks yields the same latents.
Only tests/ may import this module."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter: uint32 [..., 4]; key: uint32 [..., 2] (broadcastable) -> uint32 [..., 4]"""
    c = [np.asarray(counter[..., i], np.uint32).copy() for i in range(4)]
    k0 = np.asarray(key[..., 0], np.uint32).copy()
    k1 = np.asarray(key[..., 1], np.uint32).copy()
    with np.errstate(over='ignore'):
        for r in range(10):
            p0 = M0 * c[0].astype(np.uint64)
            p1 = M1 * c[2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return np.stack(c, -1)


def randn(count, seed, stream):
    """the first `count` normals of stream (seed, stream): element 4j+q comes from counter (j, stream), word q;
    u = ((w >> 8) + 0.5) / 2^24; words (0,1) and (2,3) are Box-Muller pairs -> (r cos t, r sin t)"""
    nblk = (count + 3) // 4
    j = np.arange(nblk, dtype=np.uint64)
    ctr = np.stack([(j & MASK).astype(np.uint32), (j >> np.uint64(32)).astype(np.uint32),
                    np.full(nblk, stream & 0xFFFFFFFF, np.uint32), np.full(nblk, (stream >> 32) & 0xFFFFFFFF, np.uint32)], -1)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], np.uint32)
    w = philox4x32_10(ctr, key[None, :])
    u = ((w >> np.uint32(8)).astype(np.float64) + 0.5) / 16777216.0
    out = np.empty((nblk, 4), np.float64)
    for a in (0, 2):
        r = np.sqrt(-2.0 * np.log(u[:, a]))
        t = 2.0 * np.pi * u[:, a + 1]
        out[:, a] = r * np.cos(t)
        out[:, a + 1] = r * np.sin(t)
    return w, out.reshape(-1)[:count]
