"""Test-side JPEG construction: files written by Pillow (libjpeg: baseline / progressive, 4:4:4 / 4:2:2 / 4:2:0, grey, restart
intervals, optimised tables) and a coefficient-level baseline writer of our own for the layouts Pillow cannot produce
(4:1:1, arbitrary component ids, 16-bit DQT, restart intervals of any length, hand-picked coefficients). Shares nothing with
the oracle or the product."""
from __future__ import annotations

import io
import struct

import numpy as np
from PIL import Image

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def pil_jpeg(img: np.ndarray, **kw) -> bytes:
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, "JPEG", **kw)
    return buf.getvalue()


def pil_decode(data: bytes) -> np.ndarray:
    im = Image.open(io.BytesIO(data))
    return np.asarray(im if im.mode == "L" else im.convert("RGB"))


def test_image(h: int, w: int, seed: int = 0, smooth: bool = True) -> np.ndarray:
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    if smooth:
        r = 128 + 100 * np.sin(xx / 17.0 + seed) * np.cos(yy / 23.0)
        g = 128 + 90 * np.cos(xx / 11.0) * np.sin(yy / 13.0 + seed)
        b = 128 + 110 * np.sin((xx + yy) / 29.0)
        img = np.stack([r, g, b], -1) + rng.normal(0, 3, (h, w, 3))
    else:
        img = rng.integers(0, 256, (h, w, 3)).astype(np.float64)
    return np.clip(img, 0, 255).astype(np.uint8)


def segments(data: bytes):
    """(marker, payload) for the header segments up to the first SOS."""
    pos, out = 2, []
    while pos + 4 <= len(data):
        assert data[pos] == 0xFF
        m = data[pos + 1]
        ln = struct.unpack(">H", data[pos + 2:pos + 4])[0]
        out.append((m, data[pos + 4:pos + 2 + ln]))
        if m == 0xDA:
            break
        pos += 2 + ln
    return out


_STD = None


def std_huffman():
    """The four Annex K tables {(class, id): (bits[16], vals)} as libjpeg writes them (non-optimised file)."""
    global _STD
    if _STD is None:
        tables = {}
        for m, p in segments(pil_jpeg(test_image(16, 16), quality=75, subsampling=2)):
            if m == 0xC4:
                pos = 0
                while pos < len(p):
                    tc, th = p[pos] >> 4, p[pos] & 15
                    bits = list(p[pos + 1:pos + 17])
                    n = sum(bits)
                    tables[(tc, th)] = (bits, list(p[pos + 17:pos + 17 + n]))
                    pos += 17 + n
        assert len(tables) == 4
        _STD = tables
    return _STD


def _codes(bits, vals):
    code, k, out = 0, 0, {}
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            out[vals[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return out


class _BitWriter:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value: int, length: int):
        if length == 0:
            return
        self.acc = (self.acc << length) | (value & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            byte = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(byte)
            if byte == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def _category(v: int) -> int:
    return int(abs(v)).bit_length()


def write_baseline(width, height, comps, qtables, coeffs, restart_interval=0, dqt16=False, tables=None, eoi=True) -> bytes:
    """comps: [(id, h, v, tq, td, ta)]; qtables: {tq: 64 natural-order values}; coeffs[c]: int array (blocks_y, blocks_x, 64)
    of QUANTISED coefficients in natural order over the padded MCU grid (blocks_x = mcus_x * h)."""
    tables = tables or std_huffman()
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    mx, my = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    out = bytearray(b"\xFF\xD8")
    for tq, q in sorted(qtables.items()):
        zz = [int(q[ZIGZAG[i]]) for i in range(64)]
        body = bytes([(16 if dqt16 else 0) | tq]) + (b"".join(struct.pack(">H", v) for v in zz) if dqt16 else bytes(zz))
        out += b"\xFF\xDB" + struct.pack(">H", 2 + len(body)) + body
    sof = struct.pack(">BHHB", 8, height, width, len(comps)) + b"".join(bytes([c[0], c[1] << 4 | c[2], c[3]]) for c in comps)
    out += b"\xFF\xC0" + struct.pack(">H", 2 + len(sof)) + sof
    for (tc, th), (bits, vals) in sorted(tables.items()):
        body = bytes([tc << 4 | th]) + bytes(bits) + bytes(vals)
        out += b"\xFF\xC4" + struct.pack(">H", 2 + len(body)) + body
    if restart_interval:
        out += b"\xFF\xDD" + struct.pack(">HH", 4, restart_interval)
    sos = bytes([len(comps)]) + b"".join(bytes([c[0], c[4] << 4 | c[5]]) for c in comps) + bytes([0, 63, 0])
    out += b"\xFF\xDA" + struct.pack(">H", 2 + len(sos)) + sos
    dc_codes = {th: _codes(*tables[(0, th)]) for (tc, th) in tables if tc == 0}
    ac_codes = {th: _codes(*tables[(1, th)]) for (tc, th) in tables if tc == 1}
    bw = _BitWriter()
    pred = [0] * len(comps)
    count = rst = 0
    for y in range(my):
        for x in range(mx):
            if restart_interval and count == restart_interval:
                bw.flush()
                out += bw.out + bytes([0xFF, 0xD0 + rst])
                bw = _BitWriter()
                rst = (rst + 1) & 7
                pred = [0] * len(comps)
                count = 0
            for ci, (_id, h, v, _tq, td, ta) in enumerate(comps):
                for vv in range(v):
                    for hh in range(h):
                        blk = coeffs[ci][y * v + vv, x * h + hh]
                        diff = int(blk[0]) - pred[ci]
                        pred[ci] = int(blk[0])
                        cat = _category(diff)
                        bw.put(*dc_codes[td][cat])
                        bw.put(diff if diff >= 0 else diff + (1 << cat) - 1, cat)
                        run = 0
                        last = max([k for k in range(1, 64) if blk[ZIGZAG[k]] != 0], default=0)
                        for k in range(1, last + 1):
                            val = int(blk[ZIGZAG[k]])
                            if val == 0:
                                run += 1
                                continue
                            while run > 15:
                                bw.put(*ac_codes[ta][0xF0])
                                run -= 16
                            cat = _category(val)
                            bw.put(*ac_codes[ta][run << 4 | cat])
                            bw.put(val if val >= 0 else val + (1 << cat) - 1, cat)
                            run = 0
                        if last < 63:
                            bw.put(*ac_codes[ta][0x00])
            count += 1
    bw.flush()
    out += bw.out
    if eoi:
        out += b"\xFF\xD9"
    return bytes(out)


def random_coefficients(rng, comps, width, height, density=0.2, dc_range=60, ac_range=25):
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    mx, my = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    out = []
    for (_id, h, v, *_rest) in comps:
        blk = np.zeros((my * v, mx * h, 64), np.int32)
        blk[..., 0] = rng.integers(-dc_range, dc_range + 1, blk.shape[:2])
        mask = rng.random(blk.shape) < density
        mask[..., 0] = False
        falloff = np.array([1.0 / (1 + (ZIGZAG.index(i)) / 6.0) for i in range(64)])
        ac = (rng.integers(-ac_range, ac_range + 1, blk.shape) * falloff).astype(np.int32)
        blk[mask] = ac[mask]
        out.append(blk)
    return out


FLAT_Q = {0: [4] * 64, 1: [6] * 64}
YCC = [(1, 1, 1, 0, 0, 0), (2, 1, 1, 1, 1, 1), (3, 1, 1, 1, 1, 1)]


def layout(luma_h: int, luma_v: int, ids=(1, 2, 3)):
    return [(ids[0], luma_h, luma_v, 0, 0, 0), (ids[1], 1, 1, 1, 1, 1), (ids[2], 1, 1, 1, 1, 1)]
