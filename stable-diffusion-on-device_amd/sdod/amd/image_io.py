"""Output formats either side of the path.

  * `output.bin`: what the reference's sample app writes and its viewer reads -- the raw image bytes, HWC, RGB, uint8,
    nothing else (simple_app.cpp:31-33; show_output.py:5-6 reshapes 512*512*3 bytes to [512, 512, 3]).
  * PNG: 8-bit RGB, non-interlaced, written with zlib only (no imaging library is assumed on the host).
"""
import struct
import zlib

import numpy as np


def _as_hwc_u8(img):
    a = img.detach().cpu().numpy() if hasattr(img, 'detach') else np.asarray(img)
    if a.ndim == 4 and a.shape[0] == 1:
        a = a[0]
    if a.ndim != 3 or a.shape[2] != 3 or a.dtype != np.uint8:
        raise ValueError(f'expected uint8 [H, W, 3], got {a.dtype} {a.shape}')
    return np.ascontiguousarray(a)


def write_output_bin(path, img):
    a = _as_hwc_u8(img)
    with open(path, 'wb') as f:
        f.write(a.tobytes())
    return path


def read_output_bin(path, height=512, width=512):
    data = np.fromfile(path, dtype=np.uint8)
    if data.size != height * width * 3:
        raise ValueError(f'{path}: {data.size} bytes, expected {height * width * 3} for {height}x{width} RGB')
    return data.reshape(height, width, 3)


def _chunk(tag, payload):
    return struct.pack('>I', len(payload)) + tag + payload + struct.pack('>I', zlib.crc32(tag + payload) & 0xFFFFFFFF)


def write_png(path, img, level=6):
    a = _as_hwc_u8(img)
    h, w, _ = a.shape
    rows = np.empty((h, 1 + 3 * w), np.uint8)
    rows[:, 0] = 0                                   # filter type 0 (None) on every scanline
    rows[:, 1:] = a.reshape(h, 3 * w)
    with open(path, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n')
        f.write(_chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 2, 0, 0, 0)))   # 8-bit, colour type 2 (RGB)
        f.write(_chunk(b'IDAT', zlib.compress(rows.tobytes(), level)))
        f.write(_chunk(b'IEND', b''))
    return path


def read_png(path):
    """reader for 8-bit RGB non-interlaced PNGs (all five scanline filters) -- enough to verify what write_png wrote
    and to load reference renders saved by common tools"""
    with open(path, 'rb') as f:
        data = f.read()
    if data[:8] != b'\x89PNG\r\n\x1a\n':
        raise ValueError('not a PNG file')
    pos, idat, w = 8, b'', None
    while pos < len(data):
        n, tag = struct.unpack('>I4s', data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        crc, = struct.unpack('>I', data[pos + 8 + n:pos + 12 + n])
        if zlib.crc32(tag + body) & 0xFFFFFFFF != crc:
            raise ValueError(f'bad CRC in chunk {tag!r}')
        if tag == b'IHDR':
            w, h, depth, ctype, _, _, interlace = struct.unpack('>IIBBBBB', body)
            if (depth, ctype, interlace) != (8, 2, 0):
                raise ValueError('only 8-bit RGB non-interlaced PNGs are supported')
        elif tag == b'IDAT':
            idat += body
        elif tag == b'IEND':
            break
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    out = np.zeros((h, 3 * w), np.uint8)
    prev = np.zeros(3 * w, np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:
            cur = np.zeros(3 * w, np.int32)
            for x in range(3 * w):
                a = cur[x - 3] if x >= 3 else 0
                b = prev[x]
                c = prev[x - 3] if x >= 3 else 0
                if ft == 1:
                    p = a
                elif ft == 3:
                    p = (a + b) >> 1
                elif ft == 4:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                else:
                    raise ValueError(f'unknown PNG filter {ft}')
                cur[x] = (line[x] + p) & 255
        out[y] = cur
        prev = cur
    return out.reshape(h, w, 3)
