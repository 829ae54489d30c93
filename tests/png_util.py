"""Test-side PNG construction (container, packing, filtering, Adam7 split) straight from the PNG specification, with
Python's zlib for the compressed stream, plus a numpy model of what the reference's toNativeImage makes of the samples.
Nothing here is shared with the oracle or the product: it is the third leg the other two are checked against."""
from __future__ import annotations

import struct
import zlib

import numpy as np

SIGNATURE = bytes([137, 80, 78, 71, 13, 10, 26, 10])
GRAY, RGB, PALETTE, GRAY_ALPHA, RGBA = 0, 2, 3, 4, 6
CHANNELS = {GRAY: 1, RGB: 3, PALETTE: 1, GRAY_ALPHA: 2, RGBA: 4}
ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]  # x0 y0 dx dy
EMPTY_ZLIB = bytes([0x78, 0x9c, 0x03, 0x00, 0x00, 0x00, 0x00, 0x01])  # png.zig:2150


def chunk(ctype: bytes, data: bytes = b"") -> bytes:
    return struct.pack(">I", len(data)) + ctype + data + struct.pack(">I", zlib.crc32(ctype + data) & 0xffffffff)


def ihdr(width, height, bit_depth, color_type, interlace=0, compression=0, filter_method=0) -> bytes:
    return chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, bit_depth, color_type, compression, filter_method, interlace))


def pack_row(samples: np.ndarray, bit_depth: int) -> bytes:
    """samples: (w, channels) integer array of raw sample values at `bit_depth`."""
    flat = samples.reshape(-1).astype(np.uint32)
    if bit_depth == 8:
        return flat.astype(np.uint8).tobytes()
    if bit_depth == 16:
        return flat.astype(">u2").tobytes()
    per = 8 // bit_depth
    pad = (-len(flat)) % per
    flat = np.concatenate([flat, np.zeros(pad, np.uint32)])
    out = np.zeros(len(flat) // per, np.uint32)
    for k in range(per):
        out |= flat[k::per] << ((per - 1 - k) * bit_depth)
    return out.astype(np.uint8).tobytes()


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def filter_row(ftype: int, row: bytes, prev: bytes | None, bpp: int) -> bytes:
    out = bytearray(len(row))
    for i, v in enumerate(row):
        a = row[i - bpp] if i >= bpp else 0
        b = prev[i] if prev is not None else 0
        c = prev[i - bpp] if prev is not None and i >= bpp else 0
        pred = (0, a, b, (a + b) // 2, _paeth(a, b, c))[ftype]
        out[i] = (v - pred) & 0xff
    return bytes(out)


def scan_data(samples: np.ndarray, bit_depth: int, color_type: int, interlace: int, filters) -> bytes:
    """The filtered scanline stream (before deflate). samples: (h, w, channels). filters: int or callable(row_index) -> int."""
    h, w = samples.shape[:2]
    bpp = max(1, CHANNELS[color_type] * bit_depth // 8)
    pick = filters if callable(filters) else (lambda y: filters)
    out = bytearray()
    blocks = [samples] if not interlace else [samples[y0::dy, x0::dx] for x0, y0, dx, dy in ADAM7]
    counter = 0
    for block in blocks:
        if block.shape[0] == 0 or block.shape[1] == 0:
            continue
        prev = None
        for y in range(block.shape[0]):
            row = pack_row(block[y], bit_depth)
            f = pick(counter)
            counter += 1
            out.append(f)
            out += filter_row(f, row, prev, bpp)
            prev = row
    return bytes(out)


def make_png(samples: np.ndarray, bit_depth: int, color_type: int, interlace: int = 0, filters=0, palette=None, trns=None,
             pre_idat: bytes = b"", idat_split: int = 0, level: int = 6, iend: bool = True) -> bytes:
    if samples.ndim == 2:
        samples = samples[:, :, None]
    h, w = samples.shape[:2]
    z = zlib.compress(scan_data(samples, bit_depth, color_type, interlace, filters), level)
    out = SIGNATURE + ihdr(w, h, bit_depth, color_type, interlace) + pre_idat
    if palette is not None:
        out += chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    if idat_split:
        for i in range(0, len(z), idat_split):
            out += chunk(b"IDAT", z[i:i + idat_split])
    else:
        out += chunk(b"IDAT", z)
    return out + (chunk(b"IEND") if iend else b"")


def random_samples(rng, h, w, bit_depth, color_type, palette_len=None):
    hi = (palette_len if color_type == PALETTE else (1 << bit_depth))
    return rng.integers(0, hi, (h, w, CHANNELS[color_type]), dtype=np.uint32)


def native_model(samples: np.ndarray, bit_depth: int, color_type: int, interlace: int, palette=None, trns=None) -> np.ndarray:
    """What png.toNativeImage returns for these samples (png.zig:852-1146, 1855-2053), written with numpy from its rules:
    16-bit samples keep their high byte, sub-byte greys scale by 255 / (2^n - 1), grey tRNS compares the SCALED 8-bit value
    with the low byte (8-bit and below) or high byte (16-bit) of the key, grey + alpha is Rgba unless the image is interlaced
    and has no tRNS (then the alpha is dropped and the image is Image(u8))."""
    if samples.ndim == 2:
        samples = samples[:, :, None]
    s = samples.astype(np.uint32)
    to8 = (lambda v: v >> 8) if bit_depth == 16 else (lambda v: v)
    h, w = s.shape[:2]
    if color_type in (GRAY, GRAY_ALPHA):
        v = s[..., 0]
        g = (v * (255 // ((1 << bit_depth) - 1))) if bit_depth < 8 else to8(v)
        alpha = np.full((h, w), 255, np.uint32)
        if color_type == GRAY_ALPHA:
            alpha = to8(s[..., 1])
        if color_type == GRAY and trns is not None and len(trns) >= 2:
            key = trns[0] if bit_depth == 16 else trns[1]
            alpha = np.where(g == key, 0, alpha)
        rgba = (color_type == GRAY_ALPHA and not interlace) or trns is not None
        if not rgba:
            return g.astype(np.uint8)
        return np.stack([g, g, g, alpha], -1).astype(np.uint8)
    if color_type == RGB:
        rgb = to8(s)
        if trns is None:
            return rgb.astype(np.uint8)
        k = 0 if bit_depth == 16 else 1
        key = np.array([trns[k], trns[2 + k], trns[4 + k]], np.uint32)
        alpha = np.where((rgb == key).all(-1), 0, 255)
        return np.concatenate([rgb, alpha[..., None]], -1).astype(np.uint8)
    if color_type == RGBA:
        return to8(s).astype(np.uint8)
    pal = np.asarray(palette, np.uint8).reshape(-1, 3)
    idx = s[..., 0]
    rgb = pal[idx]
    if trns is None:
        return rgb
    t = np.full(256, 255, np.uint8)
    t[:len(trns)] = np.frombuffer(bytes(trns), np.uint8)
    return np.concatenate([rgb, t[idx][..., None]], -1)
