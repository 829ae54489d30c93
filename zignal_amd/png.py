"""zignal_amd.png — host-side mirror of the reference's PNG codec surface (src/codecs/png.zig) over zg_png_*.

The chunk layer, inflate / deflate and de-filtering run on the host inside libzignal_hip.so; unpacking to pixels, the
conversion to the requested Image(T), the row filters and their adaptive selection run on the MI355X. Errors of the
reference's error set surface as `CodecError` with `.name` == the Zig error name.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib as L
from .image import Image, torch

GRAYSCALE, RGB, PALETTE, GRAYSCALE_ALPHA, RGBA = 0, 2, 3, 4, 6  # png.ColorType (png.zig:82-106)
FILTER_ADAPTIVE, FILTER_NONE, FILTER_SUB, FILTER_UP, FILTER_AVERAGE, FILTER_PAETH = -1, 0, 1, 2, 3, 4  # FilterMode / FilterType

_KINDS = {"u8": (L.PIXEL_U8, L.CS_GRAY, 1), "rgb_u8": (L.PIXEL_RGB_U8, L.CS_RGB, 3), "rgba_u8": (L.PIXEL_RGBA_U8, L.CS_RGBA, 4)}
_KIND_OF_PIXEL = {L.PIXEL_U8: "u8", L.PIXEL_RGB_U8: "rgb_u8", L.PIXEL_RGBA_U8: "rgba_u8"}


def decode_limits(**overrides) -> L.ZgPngLimits:
    """png.DecodeLimits{...}: the defaults (png.zig:23-41) with the given fields replaced; 0 disables a limit."""
    lim = L.ZgPngLimits()
    L.lib().zg_png_default_limits(C.byref(lim))
    for k, v in overrides.items():
        if not hasattr(lim, k):
            raise TypeError(f"DecodeLimits has no field {k}")
        setattr(lim, k, v)
    return lim


@dataclass
class EncodeOptions:
    """png.EncodeOptions (png.zig:1296-1316)."""
    filter: int = FILTER_ADAPTIVE
    compression_level: int = -1
    gamma: Optional[float] = None
    srgb_intent: Optional[int] = None

    def _c(self) -> L.ZgPngEncodeOptions:
        return L.ZgPngEncodeOptions(self.filter, self.compression_level, 0 if self.gamma is None else 1,
                                    0.0 if self.gamma is None else float(self.gamma), -1 if self.srgb_intent is None else int(self.srgb_intent))


def _buf(data: bytes):
    data = bytes(data)
    return (C.c_uint8 * max(1, len(data))).from_buffer_copy(data if data else b"\0"), len(data)


def _lim(limits):
    return C.byref(limits) if limits is not None else None


def get_info(data: bytes, limits: Optional[L.ZgPngLimits] = None) -> L.ZgPngHeader:
    """png.getInfo (png.zig:308-410)."""
    buf, n = _buf(data)
    h = L.ZgPngHeader()
    L.check(L.lib().zg_png_info(buf, n, _lim(limits), C.byref(h)))
    return h


def decode(data: bytes, limits: Optional[L.ZgPngLimits] = None):
    """png.decode (png.zig:629-794), the chunk layer: (header, native kind "u8" / "rgb_u8" / "rgba_u8", truncated)."""
    buf, n = _buf(data)
    h, native, trunc = L.ZgPngHeader(), C.c_int(0), C.c_int(0)
    L.check(L.lib().zg_png_probe(buf, n, _lim(limits), C.byref(h), C.byref(native), C.byref(trunc)))
    return h, _KIND_OF_PIXEL[native.value], bool(trunc.value)


def scan_hash(data: bytes, limits: Optional[L.ZgPngLimits] = None):
    """(FNV-1a of the inflated, de-filtered scan data, truncated): the host half of a decode, no device involved."""
    buf, n = _buf(data)
    h, t = C.c_uint64(0), C.c_int(0)
    L.check(L.lib().zg_png_scan_hash(buf, n, _lim(limits), C.byref(h), C.byref(t)))
    return h.value, bool(t.value)


def load_from_bytes(data: bytes, kind: Optional[str] = None, limits: Optional[L.ZgPngLimits] = None, device: Optional[str] = "cuda",
                    return_truncated: bool = False):
    """png.loadFromBytes(T) (png.zig:1151-1186). kind = "u8" | "rgb_u8" | "rgba_u8" names T; None keeps the file's native
    type (png.toNativeImage). device=None decodes into host memory (the zg_png_decode_host entry point)."""
    header, native_kind, _ = decode(data, limits)
    pixel, space, ch = _KINDS[kind or native_kind]
    shape = (header.height, header.width) if ch == 1 else (header.height, header.width, ch)
    if device is None:
        out = Image(np.zeros(shape, np.uint8))
    else:
        out = Image(torch.zeros(shape, dtype=torch.uint8, device=device))
    buf, n = _buf(data)
    d, trunc = out._desc(), C.c_int(0)
    if out.on_device:
        L.check(L.lib().zg_png_decode(buf, n, _lim(limits), C.byref(d), space, C.byref(trunc), out._stream()))
    else:
        L.check(L.lib().zg_png_decode_host(buf, n, _lim(limits), C.byref(d), space, C.byref(trunc)))
    return (out, bool(trunc.value)) if return_truncated else out


def load(path: str, kind: Optional[str] = None, limits: Optional[L.ZgPngLimits] = None, device: Optional[str] = "cuda") -> Image:
    """png.load (png.zig:1188-1194)."""
    with open(path, "rb") as f:
        return load_from_bytes(f.read(), kind, limits, device)


def _space_of(image: Image, space: Optional[int]) -> int:
    if space is not None:
        return space
    return {1: L.CS_GRAY, 3: L.CS_RGB, 4: L.CS_RGBA}[1 if image.data.ndim == 2 else int(image.data.shape[2])]


def encode(image, options: Optional[EncodeOptions] = None, space: Optional[int] = None) -> bytes:
    """png.encode(T) (png.zig:1400-1425): u8 -> greyscale, Rgb -> RGB, Rgba -> RGBA, anything else goes through Rgb."""
    image = Image._wrap(image)
    out, n = C.c_void_p(), C.c_size_t(0)
    d = image._desc()
    opt = (options or EncodeOptions())._c()
    lib = L.lib()
    if image.on_device:
        L.check(lib.zg_png_encode(C.byref(d), _space_of(image, space), C.byref(opt), C.byref(out), C.byref(n), image._stream()))
    else:
        L.check(lib.zg_png_encode_host(C.byref(d), _space_of(image, space), C.byref(opt), C.byref(out), C.byref(n)))
    try:
        return C.string_at(out.value, n.value)
    finally:
        lib.zg_png_free(out)


def save(image, path: str, options: Optional[EncodeOptions] = None) -> None:
    """png.save (png.zig:1427-1439)."""
    with open(path, "wb") as f:
        f.write(encode(image, options))


def filter_scanlines(image, filter: int = FILTER_ADAPTIVE):
    """filterScanlines / filterScanlinesAdaptive (png.zig:1265-1294, :1661-1719) of a device Image(u8 / Rgb / Rgba):
    a (rows, 1 + cols * channels) uint8 device tensor, one filter byte then the filtered bytes per row."""
    image = Image._wrap(image)
    if not image.on_device:
        raise ValueError("filter_scanlines works on device images (encode() takes host images too)")
    ch = 1 if image.data.ndim == 2 else int(image.data.shape[2])
    out = torch.empty((image.rows, image.cols * ch + 1), dtype=torch.uint8, device=image.data.device)
    d = image._desc()
    L.check(L.lib().zg_png_filter(C.byref(d), int(filter), C.c_void_p(out.data_ptr()), image._stream()))
    return out


def compress_scanlines(scanlines, compression_level: int = -1) -> bytes:
    """The IDAT payload of encodeRaw (png.zig:1297-1306, :1372-1391): the zlib stream of filtered scanlines (bytes, or a
    uint8 array such as filter_scanlines(...).cpu().numpy()). Host only; large inputs are deflated on several threads."""
    raw = scanlines if isinstance(scanlines, (bytes, bytearray)) else np.ascontiguousarray(scanlines, dtype=np.uint8).tobytes()
    raw = bytes(raw)
    mem, n = C.c_void_p(), C.c_size_t()
    L.check(L.lib().zg_png_compress(raw, len(raw), int(compression_level), C.byref(mem), C.byref(n)))
    try:
        return C.string_at(mem.value, n.value)
    finally:
        L.lib().zg_png_free(mem)
