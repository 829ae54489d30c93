"""ctypes binding of oracle/liboracle.so (C restatement of the reference's CPU path).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg. Images are numpy arrays: (R,C) uint8/float32 or (R,C,3|4) uint8/float32; the row stride may
exceed cols (views), the channel axis must be contiguous.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

U8, F32, RGB_U8, RGBA_U8, RGB_F32, RGBA_F32 = range(6)
ZERO, REPLICATE, MIRROR, WRAP = range(4)
NEAREST, BILINEAR, BICUBIC, CATMULL_ROM, MITCHELL, LANCZOS = range(6)
SIMILARITY, AFFINE, PROJECTIVE = range(3)
CS_GRAY, CS_RGB, CS_RGBA, CS_OKLAB, CS_XYZ, CS_YCBCR, CS_HSL, CS_HSV, CS_LAB, CS_LCH, CS_LMS, CS_OKLCH, CS_XYB = range(13)


class ZoImage(C.Structure):
    _fields_ = [("data", C.c_void_p), ("stride", C.c_size_t), ("rows", C.c_uint32),
                ("cols", C.c_uint32), ("pixel", C.c_int32)]


class ZoMethod(C.Structure):
    _fields_ = [("kind", C.c_int32), ("b", C.c_float), ("c", C.c_float), ("lanczos_lut", C.c_void_p)]


def build(native: bool = False) -> str:
    target = "liboracle_native.so" if native else "liboracle.so"
    subprocess.run(["make", "-C", _HERE] + (["native"] if native else []), check=True,
                   stdout=subprocess.DEVNULL)
    return os.path.join(_HERE, target)


_libs: dict[str, C.CDLL] = {}


def lib(native: bool = False) -> C.CDLL:
    key = "native" if native else "default"
    if key in _libs:
        return _libs[key]
    path = os.path.join(_HERE, "liboracle_native.so" if native else "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
        path = build(native)
    l = C.CDLL(path)
    l.zo_resolve_index.restype = C.c_int64
    l.zo_resolve_index.argtypes = [C.c_int64, C.c_int64, C.c_int]
    for name in ("zo_expf", "zo_logf", "zo_cbrtf", "zo_sinf", "zo_cosf"):
        getattr(l, name).restype = C.c_float
        getattr(l, name).argtypes = [C.c_float]
    l.zo_powf.restype = C.c_float
    l.zo_powf.argtypes = [C.c_float, C.c_float]
    l.zo_clamp_u8_f32.restype = C.c_uint8
    l.zo_clamp_u8_f32.argtypes = [C.c_float]
    _libs[key] = l
    return l


def pixel_of(a: np.ndarray) -> int:
    if a.ndim == 2:
        if a.dtype == np.uint8:
            return U8
        if a.dtype == np.float32:
            return F32
    elif a.ndim == 3:
        key = (a.dtype.type, a.shape[2])
        table = {(np.uint8, 3): RGB_U8, (np.uint8, 4): RGBA_U8, (np.float32, 3): RGB_F32,
                 (np.float32, 4): RGBA_F32}
        if key in table:
            return table[key]
    raise TypeError(f"unsupported image array {a.dtype} {a.shape}")


def as_image(a: np.ndarray) -> ZoImage:
    """Describe a numpy array (possibly a row-strided view) as Image(T)."""
    pixel = pixel_of(a)
    psize = a.itemsize * (a.shape[2] if a.ndim == 3 else 1)
    rows, cols = a.shape[0], a.shape[1]
    if a.ndim == 3 and a.shape[2] > 1:
        assert a.strides[2] == a.itemsize, "channel axis must be contiguous"
    if cols > 0 and rows > 0:
        assert a.strides[1] == psize, "pixels of a row must be contiguous"
        assert a.strides[0] % psize == 0
        stride = a.strides[0] // psize if rows > 1 else max(cols, a.strides[0] // psize)
    else:
        stride = cols
    return ZoImage(a.ctypes.data, stride, rows, cols, pixel)


def method(kind: int, b: float = 0.0, c: float = 0.0) -> ZoMethod:
    return ZoMethod(kind, b, c, None)


def _f32p(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with status {rc}")


# ---- thin functional API (allocates the output like the reference's allocating variants) --------

def resolve_index(idx, length, border):
    r = lib().zo_resolve_index(idx, length, border)
    return None if r < 0 else r


def gaussian_kernel(sigma: float) -> np.ndarray:
    n = lib().zo_gaussian_kernel(C.c_float(sigma), None, 0)
    if n < 0:
        raise ValueError("invalid sigma")
    out = np.empty(n, np.float32)
    lib().zo_gaussian_kernel(C.c_float(sigma), out.ctypes.data_as(C.POINTER(C.c_float)), n)
    return out


def conv_separable(src, kx, ky, border, out=None, native=False):
    out = np.empty_like(src) if out is None else out
    kxa, kxp = _f32p(kx)
    kya, kyp = _f32p(ky)
    s, d = as_image(src), as_image(out)
    rc = lib(native).zo_conv_separable(C.byref(s), C.byref(d), kxp, len(kxa), kyp, len(kya), border)
    _check(rc, "conv_separable")
    return out


def gaussian_blur(src, sigma, out=None, native=False):
    out = np.empty_like(src) if out is None else out
    s, d = as_image(src), as_image(out)
    rc = lib(native).zo_gaussian_blur(C.byref(s), C.byref(d), C.c_float(sigma))
    if rc == 2:
        raise ValueError("InvalidSigma")
    _check(rc, "gaussian_blur")
    return out


def convolve(src, kernel, border, out=None):
    out = np.empty_like(src) if out is None else out
    k, kp = _f32p(kernel)
    s, d = as_image(src), as_image(out)
    rc = lib().zo_convolve(C.byref(s), C.byref(d), kp, k.shape[0], k.shape[1], border)
    _check(rc, "convolve")
    return out


def box_blur(src, radius, out=None):
    out = np.empty_like(src) if out is None else out
    s, d = as_image(src), as_image(out)
    _check(lib().zo_box_blur(C.byref(s), C.byref(d), radius), "box_blur")
    return out


def interpolate(img, x, y, m: ZoMethod, border):
    px = np.zeros(img.shape[2:] if img.ndim == 3 else (), img.dtype)
    buf = np.zeros(4, img.dtype)
    s = as_image(img)
    ok = lib().zo_interpolate(C.byref(s), C.c_float(x), C.c_float(y), C.byref(m), border,
                              C.c_void_p(buf.ctypes.data))
    if not ok:
        return None
    if img.ndim == 2:
        return buf[0].copy()
    px[...] = buf[: img.shape[2]]
    return px


def resize(src, out_shape_or_out, m: ZoMethod):
    if isinstance(out_shape_or_out, np.ndarray):
        out = out_shape_or_out
    else:
        out = np.empty(tuple(out_shape_or_out) + src.shape[2:], src.dtype)
    s, d = as_image(src), as_image(out)
    _check(lib().zo_resize(C.byref(s), C.byref(d), C.byref(m)), "resize")
    return out


def letterbox(src, out, m: ZoMethod):
    rect = (C.c_uint32 * 4)()
    s, d = as_image(src), as_image(out)
    _check(lib().zo_letterbox(C.byref(s), C.byref(d), C.byref(m), rect), "letterbox")
    return tuple(rect)


def project(kind, mat, x, y):
    m, mp = _f32p(mat)
    ox, oy = C.c_float(), C.c_float()
    lib().zo_project(kind, mp, C.c_float(x), C.c_float(y), C.byref(ox), C.byref(oy))
    return ox.value, oy.value


def warp(src, out_shape_or_out, kind, mat, m: ZoMethod):
    if isinstance(out_shape_or_out, np.ndarray):
        out = out_shape_or_out
    else:
        out = np.empty(tuple(out_shape_or_out) + src.shape[2:], src.dtype)
    ma, mp = _f32p(mat)
    s, d = as_image(src), as_image(out)
    _check(lib().zo_warp(C.byref(s), C.byref(d), kind, mp, C.byref(m)), "warp")
    return out


def cos_sin(angle: float):
    l = lib()
    return l.zo_cosf(C.c_float(angle)), l.zo_sinf(C.c_float(angle))


def rotate_bounds(rows, cols, angle):
    ca, sa = cos_sin(angle)
    r, c = C.c_uint32(), C.c_uint32()
    lib().zo_rotate_bounds(rows, cols, C.c_float(angle), C.c_float(ca), C.c_float(sa), C.byref(r), C.byref(c))
    return r.value, c.value


def rotate_into(src, out, angle, m: ZoMethod, border):
    ca, sa = cos_sin(angle)
    s, d = as_image(src), as_image(out)
    _check(lib().zo_rotate_into(C.byref(s), C.byref(d), C.c_float(angle), C.c_float(ca), C.c_float(sa),
                                C.byref(m), border), "rotate_into")
    return out


def rotate(src, angle, m: ZoMethod, border):
    r, c = rotate_bounds(src.shape[0], src.shape[1], angle)
    out = np.empty((r, c) + src.shape[2:], src.dtype)
    return rotate_into(src, out, angle, m, border)


def extract(src, out, rect, angle, m: ZoMethod, border):
    ca, sa = cos_sin(angle)
    ra, rp = _f32p(rect)
    s, d = as_image(src), as_image(out)
    _check(lib().zo_extract(C.byref(s), C.byref(d), rp, C.c_float(angle), C.c_float(ca), C.c_float(sa),
                            C.byref(m), border), "extract")
    return out


def crop(src, rect):
    ra, rp = _f32p(rect)
    r, c = C.c_uint32(), C.c_uint32()
    lib().zo_crop_dims(rp, C.byref(r), C.byref(c))
    out = np.empty((r.value, c.value) + src.shape[2:], src.dtype)
    s, d = as_image(src), as_image(out)
    _check(lib().zo_crop(C.byref(s), C.byref(d), rp), "crop")
    return out


def flip_left_right(img):
    s = as_image(img)
    _check(lib().zo_flip_left_right(C.byref(s)), "flip_left_right")
    return img


def flip_top_bottom(img):
    s = as_image(img)
    _check(lib().zo_flip_top_bottom(C.byref(s)), "flip_top_bottom")
    return img


def insert(self_img, source, rect, angle, m: ZoMethod, blend_mode=0):
    ca, sa = cos_sin(angle)
    ra, rp = _f32p(rect)
    s, d = as_image(self_img), as_image(source)
    _check(lib().zo_insert(C.byref(s), C.byref(d), rp, C.c_float(angle), C.c_float(ca), C.c_float(sa),
                           C.byref(m), blend_mode), "insert")
    return self_img


def threshold_otsu(src):
    out = np.empty_like(src)
    t = C.c_uint8(0)
    s, d = as_image(src), as_image(out)
    _check(lib().zo_threshold_otsu(C.byref(s), C.byref(d), C.byref(t)), "threshold_otsu")
    return out, t.value


def threshold_adaptive_mean(src, radius, c):
    out = np.empty_like(src)
    s, d = as_image(src), as_image(out)
    _check(lib().zo_threshold_adaptive_mean(C.byref(s), C.byref(d), C.c_uint32(radius), C.c_float(c)), "threshold_adaptive_mean")
    return out


MORPH_DILATE, MORPH_ERODE, MORPH_OPEN, MORPH_CLOSE = range(4)


def morph(src, kernel, iterations, op):
    out = np.empty_like(src)
    k = np.ascontiguousarray(kernel, np.uint8)
    s, d = as_image(src), as_image(out)
    _check(lib().zo_morph(C.byref(s), C.byref(d), k.ctypes.data_as(C.POINTER(C.c_uint8)), k.shape[0], k.shape[1], C.c_uint32(iterations), int(op)), "morph")
    return out


OS_PERCENTILE, OS_MIDPOINT, OS_ALPHA_TRIMMED = range(3)


def order_statistic_blur(src, radius, op, param=0.5, border=MIRROR):
    out = np.empty_like(src)
    s, d = as_image(src), as_image(out)
    _check(lib().zo_order_statistic_blur(C.byref(s), C.byref(d), C.c_uint32(radius), int(op), C.c_double(param), int(border)), "order_statistic_blur")
    return out


def autocontrast(img, cutoff=0.0):
    s = as_image(img)
    _check(lib().zo_autocontrast(C.byref(s), C.c_float(cutoff)), "autocontrast")
    return img


def equalize(img):
    s = as_image(img)
    _check(lib().zo_equalize(C.byref(s)), "equalize")
    return img


def sharpen(src, radius):
    out = np.empty_like(src)
    s, d = as_image(src), as_image(out)
    _check(lib().zo_sharpen(C.byref(s), C.byref(d), C.c_uint32(radius)), "sharpen")
    return out


def integral(src):
    ch = 1 if src.ndim == 2 else src.shape[2]
    planes = np.empty((ch,) + src.shape[:2], np.float32)
    s = as_image(src)
    _check(lib().zo_integral(C.byref(s), planes.ctypes.data_as(C.POINTER(C.c_float))), "integral")
    return planes


def fill(img, value):
    s = as_image(img)
    v = np.atleast_1d(np.asarray(value, img.dtype))
    lib().zo_fill.argtypes = [C.c_void_p, C.c_void_p]
    _check(lib().zo_fill(C.byref(s), v.ctypes.data), "fill")
    return img


def set_border(img, rect, value=None):
    """Image.setBorder (image.zig:200-230), in place; rect = (l, t, r, b)."""
    s = as_image(img)
    ch = 1 if img.ndim == 2 else img.shape[2]
    v = np.zeros(ch, img.dtype) if value is None else np.atleast_1d(np.asarray(value, img.dtype))
    lib().zo_set_border.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    _check(lib().zo_set_border(C.byref(s), (C.c_uint32 * 4)(*[int(x) for x in rect]), v.ctypes.data), "set_border")
    return img


def invert(img):
    s = as_image(img)
    _check(lib().zo_invert(C.byref(s)), "invert")
    return img


def blend_rgba_u8(base, overlay, mode: int):
    """Rgba(u8).blend(overlay, mode) (blending.zig:27-157)."""
    b = (C.c_uint8 * 4)(*[int(v) for v in base]); ov = (C.c_uint8 * 4)(*[int(v) for v in overlay])
    lib().zo_blend_rgba_u8(b, ov, int(mode))
    return tuple(b)


def srgb_to_linear_lut() -> np.ndarray:
    out = np.empty(256, np.float32)
    lib().zo_srgb_to_linear_lut(out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


_CS_LAYOUT = {(CS_GRAY, np.uint8): ((), np.uint8), (CS_GRAY, np.float32): ((), np.float32)}


def convert(src, src_space, dst_space, dst_dtype, dst_channels, out=None, srgb_lut=None):
    if out is None:
        shape = src.shape[:2] + ((dst_channels,) if dst_channels > 1 else ())
        out = np.empty(shape, dst_dtype)
    s, d = as_image(src), as_image(out)
    lut = None
    if srgb_lut is not None:
        srgb_lut = np.ascontiguousarray(srgb_lut, np.float32)
        lut = srgb_lut.ctypes.data_as(C.POINTER(C.c_float))
    _check(lib().zo_convert(C.byref(s), src_space, C.byref(d), dst_space, lut), "convert")
    return out


def color_to(values, from_space: int, to_space: int, dtype=np.float64) -> np.ndarray:
    """<Space>(T).to(target) on one colour (fields in declaration order), T = f64 (default) or f32."""
    ct, fn = (C.c_double, lib().zo_color_to_f64) if dtype == np.float64 else (C.c_float, lib().zo_color_to_f32)
    fn.argtypes = [C.c_int, C.POINTER(ct), C.c_int, C.POINTER(ct)]
    fn.restype = None
    vals = list(values) + [0] * (4 - len(values))
    a, o = (ct * 4)(*vals), (ct * 4)()
    fn(from_space, a, to_space, o)
    n = 1 if to_space == CS_GRAY else (4 if to_space == CS_RGBA else 3)
    return np.array(list(o)[:n], dtype)


def homography_from_4pts(from_pts, to_pts) -> np.ndarray:
    f = np.ascontiguousarray(from_pts, np.float64).reshape(8)
    t = np.ascontiguousarray(to_pts, np.float64).reshape(8)
    m = np.empty(9, np.float32)
    rc = lib().zo_homography_from_4pts(f.ctypes.data_as(C.POINTER(C.c_double)),
                                       t.ctypes.data_as(C.POINTER(C.c_double)),
                                       m.ctypes.data_as(C.POINTER(C.c_float)))
    _check(rc, "homography")
    return m.reshape(3, 3)


def splitmix64_bytes(seed: int, n: int) -> np.ndarray:
    """Seeded byte stream shared by oracle and GPU inputs."""
    m = (n + 7) // 8
    idx = (np.arange(1, m + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed))
    z = idx
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    return z.view(np.uint8)[:n].copy()


def synth_u8(seed: int, shape) -> np.ndarray:
    n = int(np.prod(shape))
    return splitmix64_bytes(seed, n).reshape(shape)


def synth_f32(seed: int, shape) -> np.ndarray:
    n = int(np.prod(shape))
    u = splitmix64_bytes(seed, 4 * n).view(np.uint32)
    return ((u >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).reshape(shape)


def sobel(src):
    out = np.empty(src.shape[:2], np.uint8)
    s, d = as_image(src), as_image(out)
    _check(lib().zo_sobel(C.byref(s), C.byref(d)), "sobel")
    return out


def canny(src, sigma, low, high):
    out = np.empty(src.shape[:2], np.uint8)
    s, d = as_image(src), as_image(out)
    _check(lib().zo_canny(C.byref(s), C.byref(d), C.c_float(sigma), C.c_float(low), C.c_float(high)), "canny")
    return out


def shen_castan(src, smooth=0.9, window_size=7, high_ratio=0.99, low_rel=0.5, hysteresis=True, use_nms=False):
    out = np.empty(src.shape[:2], np.uint8)
    s, d = as_image(src), as_image(out)
    _check(lib().zo_shen_castan(C.byref(s), C.byref(d), C.c_float(smooth), C.c_uint32(window_size), C.c_float(high_ratio), C.c_float(low_rel),
                                int(bool(hysteresis)), int(bool(use_nms))), "shen_castan")
    return out


def isef_plane(plane, smooth=0.9):
    """isefFilter2D (edges.zig:308-349) of a contiguous f32 plane; returns a new array."""
    out = np.ascontiguousarray(plane, dtype=np.float32).copy()
    fn = lib().zo_isef_plane
    fn.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float]
    _check(fn(out.ctypes.data, out.shape[0], out.shape[1], C.c_float(smooth)), "isef_plane")
    return out


def motion_blur_linear(src, angle, distance, cos_sin=None):
    out = np.empty_like(src)
    ca, sa = cos_sin if cos_sin is not None else (float(np.cos(np.float32(angle), dtype=np.float32)), float(np.sin(np.float32(angle), dtype=np.float32)))
    s, d = as_image(src), as_image(out)
    _check(lib().zo_motion_blur_linear(C.byref(s), C.byref(d), C.c_float(angle), C.c_float(ca), C.c_float(sa), C.c_uint32(distance)), "motion_blur_linear")
    return out


def motion_blur_radial(src, center_x, center_y, strength, spin):
    out = np.empty_like(src)
    s, d = as_image(src), as_image(out)
    _check(lib().zo_motion_blur_radial(C.byref(s), C.byref(d), C.c_float(center_x), C.c_float(center_y), C.c_float(strength), int(bool(spin))), "motion_blur_radial")
    return out


def pyramid(source, n_levels, scale_factor, blur_sigma):
    """ImagePyramid.build (pyramid.zig:31-102) composed from the oracle's own gaussian_blur and resize."""
    l = lib()
    l.zo_pyramid_scale.restype = C.c_float
    levels = [source]
    for i in range(1, n_levels):
        scale = l.zo_pyramid_scale(C.c_float(scale_factor), i)
        r, c, sig = C.c_uint32(), C.c_uint32(), C.c_float()
        stop = l.zo_pyramid_level(source.shape[0], source.shape[1], C.c_float(scale), C.c_float(blur_sigma), C.byref(r), C.byref(c), C.byref(sig))
        if stop:
            break
        base = gaussian_blur(source, sig.value) if sig.value > 0.5 else source
        levels.append(resize(base, (r.value, c.value), method(BILINEAR)))
    return levels


# ---- PNG (oracle/png.c; src/codecs/png.zig) -------------------------------------------------------

class PngError(Exception):
    """One of the reference's PNG errors; `.name` is the Zig error name (error.InvalidCrc -> "InvalidCrc")."""

    def __init__(self, name: str):
        super().__init__(name)
        self.name = name


class ZoPngHeader(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("bit_depth", C.c_uint8), ("color_type", C.c_uint8),
                ("compression_method", C.c_uint8), ("filter_method", C.c_uint8), ("interlace_method", C.c_uint8),
                ("has_gamma", C.c_uint8), ("has_srgb", C.c_uint8), ("srgb_intent", C.c_uint8), ("gamma", C.c_float)]


class ZoPngLimits(C.Structure):
    _fields_ = [("max_png_bytes", C.c_size_t), ("max_chunk_bytes", C.c_size_t), ("max_idat_bytes", C.c_size_t),
                ("max_chunks", C.c_size_t), ("max_width", C.c_uint32), ("max_height", C.c_uint32),
                ("max_pixels", C.c_uint64), ("max_decompressed_bytes", C.c_size_t)]


def png_limits(**overrides) -> ZoPngLimits:
    lim = ZoPngLimits()
    lib().zo_png_default_limits(C.byref(lim))
    for k, v in overrides.items():
        setattr(lim, k, v)
    return lim


def _png_check(rc: int):
    if rc != 0:
        fn = lib().zo_png_error_name
        fn.restype = C.c_char_p
        raise PngError(fn(rc).decode())


def _png_buf(data: bytes):
    return (C.c_uint8 * max(1, len(data))).from_buffer_copy(data if len(data) else b"\0")


def png_crc(data: bytes) -> int:
    fn = lib().zo_png_crc
    fn.restype = C.c_uint32
    fn.argtypes = [C.c_void_p, C.c_size_t]
    return fn(_png_buf(data), len(data))


def png_paeth(a: int, b: int, c: int) -> int:
    fn = lib().zo_png_paeth
    fn.restype = C.c_uint8
    return fn(a, b, c)


def png_info(data: bytes, limits: ZoPngLimits | None = None) -> ZoPngHeader:
    h = ZoPngHeader()
    fn = lib().zo_png_info
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    _png_check(fn(_png_buf(data), len(data), C.byref(limits) if limits else None, C.byref(h)))
    return h


def png_decode_chunks(data: bytes, limits: ZoPngLimits | None = None):
    """png.decode: (header, truncated, palette_len or -1, trns_len or -1)."""
    h, t, pl, tl = ZoPngHeader(), C.c_int(0), C.c_int(0), C.c_int(0)
    fn = lib().zo_png_decode_chunks
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p] + [C.c_void_p] * 4
    _png_check(fn(_png_buf(data), len(data), C.byref(limits) if limits else None, C.byref(h), C.byref(t), C.byref(pl), C.byref(tl)))
    return h, bool(t.value), pl.value, tl.value


def png_decode_native(data: bytes, limits: ZoPngLimits | None = None):
    """png.decode + png.toNativeImage: (pixels ndarray in the native type, truncated, header)."""
    h, native, t, px = ZoPngHeader(), C.c_int(0), C.c_int(0), C.c_void_p()
    fn = lib().zo_png_decode_native
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p] + [C.c_void_p] * 4
    _png_check(fn(_png_buf(data), len(data), C.byref(limits) if limits else None, C.byref(h), C.byref(native), C.byref(px), C.byref(t)))
    ch = {U8: 1, RGB_U8: 3, RGBA_U8: 4}[native.value]
    n = h.height * h.width * ch
    arr = np.frombuffer(C.string_at(px.value, n), np.uint8).copy()
    free = lib().zo_png_free
    free.argtypes = [C.c_void_p]
    free(px)
    return arr.reshape((h.height, h.width) if ch == 1 else (h.height, h.width, ch)), bool(t.value), h


def png_scan_hash(data: bytes, limits: ZoPngLimits | None = None):
    """(FNV-1a of the inflated, de-filtered scan data, truncated): the host half of a PNG decode."""
    h, t = C.c_uint64(0), C.c_int(0)
    fn = lib().zo_png_scan_hash
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    _png_check(fn(_png_buf(data), len(data), C.byref(limits) if limits else None, C.byref(h), C.byref(t)))
    return h.value, bool(t.value)


def png_load(data: bytes, kind: str, limits: ZoPngLimits | None = None) -> np.ndarray:
    """png.loadFromBytes(T): the native image, converted with Image.convert when T differs (png.zig:1151-1186)."""
    native, _, _ = png_decode_native(data, limits)
    spaces = {1: CS_GRAY, 3: CS_RGB, 4: CS_RGBA}
    ch = {"u8": 1, "rgb_u8": 3, "rgba_u8": 4}[kind]
    nch = 1 if native.ndim == 2 else native.shape[2]
    if nch == ch:
        return native
    return convert(native, spaces[nch], spaces[ch], np.uint8, ch)


PNG_ADAPTIVE = -1


def png_filter(img: np.ndarray, mode: int = PNG_ADAPTIVE) -> np.ndarray:
    """filterScanlines / filterScanlinesAdaptive on an 8-bit u8 / rgb / rgba image: rows x (1 + row bytes)."""
    img = np.ascontiguousarray(img, np.uint8)
    rows = img.shape[0]
    bpp = 1 if img.ndim == 2 else img.shape[2]
    rb = img.shape[1] * bpp
    out = np.empty((rows, rb + 1), np.uint8)
    fn = lib().zo_png_filter
    fn.argtypes = [C.c_void_p, C.c_uint32, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    _png_check(fn(img.ctypes.data, rows, rb, bpp, mode, out.ctypes.data))
    return out


def png_encode_stored(img: np.ndarray, mode: int = PNG_ADAPTIVE) -> bytes:
    im = as_image(np.ascontiguousarray(img))
    out, n = C.c_void_p(), C.c_size_t(0)
    fn = lib().zo_png_encode_stored
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    _png_check(fn(C.byref(im), mode, C.byref(out), C.byref(n)))
    data = C.string_at(out.value, n.value)
    free = lib().zo_png_free
    free.argtypes = [C.c_void_p]
    free(out)
    return data


# ---- JPEG (oracle/jpeg.c; src/codecs/jpeg.zig) ----------------------------------------------------

class JpegError(Exception):
    """One of the reference's JPEG errors; `.name` is the Zig error name."""

    def __init__(self, name: str):
        super().__init__(name)
        self.name = name


class ZoJpegHeader(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("precision", C.c_uint8), ("num_components", C.c_uint8),
                ("progressive", C.c_uint8), ("subsampling", C.c_int8)]


class ZoJpegLimits(C.Structure):
    _fields_ = [("max_jpeg_bytes", C.c_size_t), ("max_marker_bytes", C.c_size_t), ("max_width", C.c_uint32), ("max_height", C.c_uint32),
                ("max_pixels", C.c_uint64), ("max_blocks", C.c_size_t), ("max_scans", C.c_size_t)]


def jpeg_limits(**overrides) -> ZoJpegLimits:
    lim = ZoJpegLimits()
    lib().zo_jpeg_default_limits(C.byref(lim))
    for k, v in overrides.items():
        setattr(lim, k, v)
    return lim


def _jpeg_check(rc: int):
    if rc != 0:
        fn = lib().zo_jpeg_error_name
        fn.restype = C.c_char_p
        raise JpegError(fn(rc).decode())


def jpeg_info(data: bytes, limits: ZoJpegLimits | None = None) -> ZoJpegHeader:
    h = ZoJpegHeader()
    fn = lib().zo_jpeg_info
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    _jpeg_check(fn(_png_buf(data), len(data), C.byref(limits) if limits else None, C.byref(h)))
    return h


def jpeg_decode_state(data: bytes, limits: ZoJpegLimits | None = None):
    """jpeg.decode: (header, scan_limit_reached) without rendering."""
    h, lim_hit, native = ZoJpegHeader(), C.c_int(0), C.c_int(0)
    fn = lib().zo_jpeg_decode_native
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p] + [C.c_void_p] * 4
    _jpeg_check(fn(_png_buf(data), len(data), C.byref(limits) if limits else None, C.byref(h), C.byref(native), None, C.byref(lim_hit)))
    return h, bool(lim_hit.value)


def jpeg_decode_native(data: bytes, limits: ZoJpegLimits | None = None):
    """jpeg.decode + jpeg.toNativeImage: (pixels, header, scan_limit_reached); pixels are (h, w) u8 or (h, w, 3) u8."""
    h, lim_hit, native, px = ZoJpegHeader(), C.c_int(0), C.c_int(0), C.c_void_p()
    fn = lib().zo_jpeg_decode_native
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p] + [C.c_void_p] * 4
    _jpeg_check(fn(_png_buf(data), len(data), C.byref(limits) if limits else None, C.byref(h), C.byref(native), C.byref(px), C.byref(lim_hit)))
    ch = 1 if native.value == U8 else 3
    arr = np.frombuffer(C.string_at(px.value, h.height * h.width * ch), np.uint8).copy()
    free = lib().zo_jpeg_free
    free.argtypes = [C.c_void_p]
    free(px)
    return arr.reshape((h.height, h.width) if ch == 1 else (h.height, h.width, 3)), h, bool(lim_hit.value)


def jpeg_get_bits(data: bytes, counts) -> list:
    """BitReader.getBits for each count in turn; stops at the first failure (UnexpectedEndOfData)."""
    n = len(counts)
    arr, out = (C.c_int * n)(*counts), (C.c_uint32 * n)()
    fn = lib().zo_jpeg_get_bits
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]
    got = fn(_png_buf(data), len(data), arr, n, out)
    return list(out)[:got]


def jpeg_coefficient_hash(data: bytes, limits: ZoJpegLimits | None = None) -> int:
    """FNV-1a of the entropy-decoded coefficient blocks (decode + performBlockScan), before dequantisation."""
    h = C.c_uint64(0)
    fn = lib().zo_jpeg_coefficient_hash
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    _jpeg_check(fn(_png_buf(data), len(data), C.byref(limits) if limits else None, C.byref(h)))
    return h.value


def jpeg_load(data: bytes, kind: str, limits: ZoJpegLimits | None = None) -> np.ndarray:
    """jpeg.loadFromBytes(T) (jpeg.zig:2825-2851)."""
    native, _, _ = jpeg_decode_native(data, limits)
    spaces = {1: CS_GRAY, 3: CS_RGB, 4: CS_RGBA}
    ch = {"u8": 1, "rgb_u8": 3, "rgba_u8": 4}[kind]
    nch = 1 if native.ndim == 2 else 3
    if nch == ch:
        return native
    return convert(native, spaces[nch], spaces[ch], np.uint8, ch)


def jpeg_idct8x8(block) -> np.ndarray:
    b = np.ascontiguousarray(block, np.int32).reshape(64).copy()
    fn = lib().zo_jpeg_idct8x8
    fn.argtypes = [C.c_void_p]
    fn.restype = None
    fn(b.ctypes.data)
    return b.reshape(8, 8)


def jpeg_encode(img: np.ndarray, quality: int = 90, subsampling: int = 2, density_dpi: int = 72, comment: bytes | None = None) -> bytes:
    """jpeg.encode(T) (jpeg.zig:307-329): u8 -> greyscale, Rgb -> YCbCr, any other T is converted to Rgb first."""
    img = np.array(img, order="C", copy=True)  # a fresh copy: canonical strides even for 1-pixel-wide slices
    if not (img.dtype == np.uint8 and (img.ndim == 2 or img.shape[2] == 3)):
        src_space = {1: CS_GRAY, 3: CS_RGB, 4: CS_RGBA}[1 if img.ndim == 2 else img.shape[2]]
        img = convert(img, src_space, CS_RGB, np.uint8, 3)
    im = as_image(img)
    out, n = C.c_void_p(), C.c_size_t(0)
    fn = lib().zo_jpeg_encode
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p]
    rc = fn(C.byref(im), quality, subsampling, density_dpi, comment, len(comment) if comment else 0, C.byref(out), C.byref(n))
    if rc:
        raise JpegError({1: "InvalidImageDimensions", 2: "ImageTooLarge"}.get(rc, "OutOfMemory"))
    data = C.string_at(out.value, n.value)
    free = lib().zo_jpeg_free
    free.argtypes = [C.c_void_p]
    free(out)
    return data


def jpeg_fdct8x8(block) -> np.ndarray:
    b = np.ascontiguousarray(block, np.int32).reshape(64)
    out = np.empty(64, np.int32)
    fn = lib().zo_jpeg_fdct8x8
    fn.argtypes = [C.c_void_p, C.c_void_p]
    fn.restype = None
    fn(b.ctypes.data, out.ctypes.data)
    return out.reshape(8, 8)
