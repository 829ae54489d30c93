"""zignal_amd.jpeg — host-side mirror of the reference's JPEG decoding surface (src/codecs/jpeg.zig) over zg_jpeg_*.

Markers and Huffman decoding run on the host inside libzignal_hip.so; dequantisation, the IDCT, chroma upsampling, colour
conversion and the conversion to the requested Image(T) run on the MI355X. Errors of the reference's error set surface as
`CodecError` with `.name` == the Zig error name. Encoding is the mirror image: colour conversion, chroma averaging, the forward
DCT and quantisation on the device, the Huffman coder on the host; the file is byte for byte the reference's.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib as L
from .image import Image, torch

_KINDS = {"u8": (L.PIXEL_U8, L.CS_GRAY, 1), "rgb_u8": (L.PIXEL_RGB_U8, L.CS_RGB, 3), "rgba_u8": (L.PIXEL_RGBA_U8, L.CS_RGBA, 4)}
SUBSAMPLING = {0: "yuv444", 1: "yuv422", 2: "yuv420", -1: None}  # jpeg.Subsampling (jpeg.zig:260-283)


def decode_limits(**overrides) -> L.ZgJpegLimits:
    """jpeg.DecodeLimits{...}: the defaults (jpeg.zig:19-33) with the given fields replaced; 0 disables a limit."""
    lim = L.ZgJpegLimits()
    L.lib().zg_jpeg_default_limits(C.byref(lim))
    for k, v in overrides.items():
        if not hasattr(lim, k):
            raise TypeError(f"DecodeLimits has no field {k}")
        setattr(lim, k, v)
    return lim


def _buf(data: bytes):
    data = bytes(data)
    return (C.c_uint8 * max(1, len(data))).from_buffer_copy(data if data else b"\0"), len(data)


def get_info(data: bytes, limits: Optional[L.ZgJpegLimits] = None) -> L.ZgJpegHeader:
    """jpeg.getInfo (jpeg.zig:77-179)."""
    buf, n = _buf(data)
    h = L.ZgJpegHeader()
    L.check(L.lib().zg_jpeg_info(buf, n, C.byref(limits) if limits is not None else None, C.byref(h)))
    return h


def decode(data: bytes, limits: Optional[L.ZgJpegLimits] = None):
    """jpeg.decode (jpeg.zig:2035-2151), host only: (header, scan_limit_reached). Progressive scans are entropy-decoded."""
    buf, n = _buf(data)
    h, hit = L.ZgJpegHeader(), C.c_int(0)
    L.check(L.lib().zg_jpeg_probe(buf, n, C.byref(limits) if limits is not None else None, C.byref(h), C.byref(hit)))
    return h, bool(hit.value)


def coefficient_hash(data: bytes, limits: Optional[L.ZgJpegLimits] = None) -> int:
    """FNV-1a of the entropy-decoded coefficients (jpeg.decode + performBlockScan): the host half of a decode, no device."""
    buf, n = _buf(data)
    h = C.c_uint64(0)
    L.check(L.lib().zg_jpeg_coefficient_hash(buf, n, C.byref(limits) if limits is not None else None, C.byref(h)))
    return h.value


def load_from_bytes(data: bytes, kind: Optional[str] = None, limits: Optional[L.ZgJpegLimits] = None, device: Optional[str] = "cuda",
                    return_scan_limit_reached: bool = False):
    """jpeg.loadFromBytes(T) (jpeg.zig:2825-2851). kind = "u8" | "rgb_u8" | "rgba_u8" names T; None keeps the file's native
    type (u8 for one component, rgb_u8 otherwise). device=None decodes into host memory (zg_jpeg_decode_host)."""
    try:  # dimensions only (to allocate the image): every check and limit is the decode's own, below
        header = get_info(data, decode_limits(max_jpeg_bytes=2**62))
        rows, cols, comps = header.height, header.width, header.num_components
        # a few header bytes must not size a multi-gigabyte allocation the decode is going to refuse anyway: a frame
        # beyond the effective limits (the caller's, else DecodeLimits{}) gets a 1 x 1 stand-in and the decode below
        # reports error.ImageTooLarge in its own words, before it ever looks at the destination
        eff = limits if limits is not None else decode_limits()
        if ((eff.max_width and cols > eff.max_width) or (eff.max_height and rows > eff.max_height)
                or (eff.max_pixels and rows * cols > eff.max_pixels)):
            rows, cols = 1, 1
    except L.CodecError:
        rows, cols, comps = 1, 1, 3  # no readable frame header: let the decode report what is wrong, in its own words
    native = "u8" if comps == 1 else "rgb_u8"
    _pixel, space, ch = _KINDS[kind or native]
    rows, cols = max(rows, 1), max(cols, 1)
    shape = (rows, cols) if ch == 1 else (rows, cols, ch)
    out = Image(np.zeros(shape, np.uint8)) if device is None else Image(torch.zeros(shape, dtype=torch.uint8, device=device))
    buf, n = _buf(data)
    d, hit = out._desc(), C.c_int(0)
    lim = C.byref(limits) if limits is not None else None
    if out.on_device:
        L.check(L.lib().zg_jpeg_decode(buf, n, lim, C.byref(d), space, C.byref(hit), out._stream()))
    else:
        L.check(L.lib().zg_jpeg_decode_host(buf, n, lim, C.byref(d), space, C.byref(hit)))
    return (out, bool(hit.value)) if return_scan_limit_reached else out


def load(path: str, kind: Optional[str] = None, limits: Optional[L.ZgJpegLimits] = None, device: Optional[str] = "cuda") -> Image:
    """jpeg.load (jpeg.zig:2853-2858)."""
    with open(path, "rb") as f:
        return load_from_bytes(f.read(), kind, limits, device)


YUV444, YUV422, YUV420 = 0, 1, 2  # jpeg.Subsampling


@dataclass
class EncodeOptions:
    """jpeg.EncodeOptions (jpeg.zig:284-290)."""
    quality: int = 90
    subsampling: int = YUV420
    density_dpi: int = 72
    comment: Optional[bytes] = None

    def _c(self) -> L.ZgJpegEncodeOptions:
        return L.ZgJpegEncodeOptions(int(self.quality), int(self.subsampling), int(self.density_dpi), self.comment, len(self.comment) if self.comment else 0)


def encode(image, options: Optional[EncodeOptions] = None, space: Optional[int] = None) -> bytes:
    """jpeg.encode(T) (jpeg.zig:307-329): u8 -> one component, Rgb -> YCbCr, anything else goes through Rgb."""
    image = Image._wrap(image)
    if space is None:
        space = {1: L.CS_GRAY, 3: L.CS_RGB, 4: L.CS_RGBA}[1 if image.data.ndim == 2 else int(image.data.shape[2])]
    out, n = C.c_void_p(), C.c_size_t(0)
    d, opt, lib = image._desc(), (options or EncodeOptions())._c(), L.lib()
    if image.on_device:
        L.check(lib.zg_jpeg_encode(C.byref(d), space, C.byref(opt), C.byref(out), C.byref(n), image._stream()))
    else:
        L.check(lib.zg_jpeg_encode_host(C.byref(d), space, C.byref(opt), C.byref(out), C.byref(n)))
    try:
        return C.string_at(out.value, n.value)
    finally:
        lib.zg_jpeg_free(out)


def save(image, path: str, options: Optional[EncodeOptions] = None) -> None:
    """jpeg.save (jpeg.zig:293-303): EncodeOptions{ .subsampling = .yuv420 } unless told otherwise."""
    with open(path, "wb") as f:
        f.write(encode(image, options))


def encode_blocks(blocks, rows: int, cols: int, gray: bool = False, options: Optional[EncodeOptions] = None) -> bytes:
    """The host half of encode (jpeg.zig:771-817, :929-1043) around quantised coefficient blocks: an int16 array of
    (n_blocks, 64) in the device half's order (luma row-major on its padded block grid, then all Cb, then all Cr)."""
    blocks = np.ascontiguousarray(blocks, dtype=np.int16)
    opt = (options or EncodeOptions())._c()
    hm = 1 if gray or opt.subsampling == 0 else 2
    vm = 2 if not gray and opt.subsampling == 2 else 1
    mcus = -(-int(cols) // (8 * hm)) * -(-int(rows) // (8 * vm)) if rows > 0 and cols > 0 else 0
    if rows > 0 and cols > 0 and blocks.size != mcus * (hm * vm + (0 if gray else 2)) * 64:
        raise ValueError(f"encode_blocks: {blocks.size // 64} blocks given, the layout has {mcus * (hm * vm + (0 if gray else 2))}")
    out, n = C.c_void_p(), C.c_size_t(0)
    L.check(L.lib().zg_jpeg_encode_blocks(C.c_void_p(blocks.ctypes.data), int(rows), int(cols), int(bool(gray)), C.byref(opt), C.byref(out), C.byref(n)))
    try:
        return C.string_at(out.value, n.value)
    finally:
        L.lib().zg_jpeg_free(out)
