"""Host-side mirror of zignal's `Image(T)` call surface over the libzignal_hip C ABI.

Method names, argument meaning and error behaviour follow the reference's Python binding
(`zignal.Image`, reference bindings/python/src/image.zig:922-1180) and, underneath, the Zig methods
(reference src/image.zig:375-994): outputs are pre-allocated by the allocating variants exactly
where the reference allocates, `DimensionMismatch` / `InvalidArgument` are raised where the
reference returns `error.DimensionMismatch` / `error.InvalidSigma` / `error.InvalidScaleFactor`.

Two flavours share all code:
  * `Image(ndarray)`        host pixels (numpy); every call is synchronous: H2D -> kernel -> D2H
                            through the zg_<op>_host entry points.
  * `Image(torch_tensor)`   device pixels (a CUDA/HIP tensor used purely as device memory); calls are
                            asynchronous on torch's current stream through the zg_<op> entry points.

Pixels: (R, C) uint8 / float32, (R, C, 3|4) uint8 / float32. Views (row stride > cols) are allowed.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib as L

try:  # torch is plumbing (device memory + streams); the host flavour works without it
    import torch
except Exception:  # pragma: no cover
    torch = None


class Interpolation:
    """reference src/image/interpolation.zig:53-68 (a tagged union; Mitchell carries b, c)."""

    def __init__(self, kind: int, b: float = 0.0, c: float = 0.0):
        self.kind, self.b, self.c = kind, float(b), float(c)

    def _c(self) -> L.ZgMethod:
        return L.ZgMethod(self.kind, self.b, self.c, None)

    def __repr__(self):
        names = ["nearest", "bilinear", "bicubic", "catmull_rom", "mitchell", "lanczos"]
        return f"Interpolation.{names[self.kind]}"


Interpolation.nearest = Interpolation(L.INTERP_NEAREST)
Interpolation.bilinear = Interpolation(L.INTERP_BILINEAR)
Interpolation.bicubic = Interpolation(L.INTERP_BICUBIC)
Interpolation.catmull_rom = Interpolation(L.INTERP_CATMULL_ROM)
Interpolation.lanczos = Interpolation(L.INTERP_LANCZOS)
# `.{ .b = 1 / 3, .c = 1 / 3 }` is comptime-integer division in the reference (interpolation.zig:65),
# i.e. b = c = 0; kept as written there. Interpolation.mitchell(1/3, 1/3) gives the textbook filter.
Interpolation.mitchell_default = Interpolation(L.INTERP_MITCHELL, 0.0, 0.0)
Interpolation.mitchell = staticmethod(lambda b, c: Interpolation(L.INTERP_MITCHELL, b, c))


class BorderMode:
    """reference src/image/border.zig:10-18"""
    zero, replicate, mirror, wrap = range(4)


class Blending:
    """reference src/blending.zig:8-22 (same ordinals)."""
    (none, normal, multiply, screen, overlay, soft_light, hard_light, color_dodge, color_burn, darken, lighten,
     difference, exclusion) = range(13)


_PIXEL_BY_LAYOUT = {
    ("uint8", 1): L.PIXEL_U8, ("float32", 1): L.PIXEL_F32,
    ("uint8", 3): L.PIXEL_RGB_U8, ("uint8", 4): L.PIXEL_RGBA_U8,
    ("float32", 3): L.PIXEL_RGB_F32, ("float32", 4): L.PIXEL_RGBA_F32,
}


def _is_torch(a) -> bool:
    return torch is not None and isinstance(a, torch.Tensor)


def _f32_array(values) -> Tuple[np.ndarray, "C._Pointer"]:
    a = np.ascontiguousarray(values, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


class Image:
    """Image(T): rows, cols, stride (pixels) and a pixel buffer, on the host or on the device."""

    def __init__(self, data):
        if _is_torch(data):
            if not data.is_cuda:
                raise ValueError("torch tensors are accepted as device memory only; pass numpy for host pixels")
            dtype = str(data.dtype).replace("torch.", "")
            itemsize = data.element_size()
            strides = tuple(s * itemsize for s in data.stride())
        else:
            if data is None:
                raise TypeError("expected an array, got None")  # numpy_interop.zig: TypeError for None / a wrong dtype
            data = np.asarray(data)
            dtype = data.dtype.name
            itemsize = data.itemsize
            strides = data.strides
        if data.ndim not in (2, 3):
            raise ValueError("expected (rows, cols) or (rows, cols, channels)")
        ch = 1 if data.ndim == 2 else int(data.shape[2])
        key = (dtype, ch)
        if key not in _PIXEL_BY_LAYOUT:
            if dtype not in ("uint8", "float32"):
                raise TypeError(f"unsupported dtype {dtype} (uint8 or float32)")
            raise ValueError(f"unsupported channel count {ch} (1, 3 or 4)")  # a wrong shape is a ValueError, as in the reference
        self.data = data
        self.pixel = _PIXEL_BY_LAYOUT[key]
        self.rows, self.cols = int(data.shape[0]), int(data.shape[1])
        psize = itemsize * ch
        if data.ndim == 3 and ch > 1 and strides[2] != itemsize:
            raise ValueError("the channel axis must be contiguous")
        if self.rows and self.cols:
            if strides[1] != psize or strides[0] % psize:
                raise ValueError("pixels of a row must be contiguous and rows pixel-aligned")
            self.stride = strides[0] // psize if self.rows > 1 else max(self.cols, strides[0] // psize)
        else:
            self.stride = self.cols

    # ---- plumbing ---------------------------------------------------------------------------
    @property
    def on_device(self) -> bool:
        return _is_torch(self.data)

    @property
    def shape(self):
        return tuple(self.data.shape)

    def _desc(self) -> L.ZgImage:
        ptr = self.data.data_ptr() if self.on_device else self.data.ctypes.data
        return L.ZgImage(ptr, self.stride, self.rows, self.cols, self.pixel)

    def _stream(self):
        # the stream of the device that OWNS the pixels, not of whatever device happens to be current
        return C.c_void_p(torch.cuda.current_stream(self.data.device).cuda_stream)

    def _like(self, rows: Optional[int] = None, cols: Optional[int] = None, dtype=None, channels=None) -> "Image":
        rows = self.rows if rows is None else rows
        cols = self.cols if cols is None else cols
        if channels is None:
            shape = (rows, cols) + tuple(self.data.shape[2:])
        else:
            shape = (rows, cols) if channels == 1 else (rows, cols, channels)
        if self.on_device:
            return Image(torch.empty(shape, dtype=dtype or self.data.dtype, device=self.data.device))
        return Image(np.empty(shape, dtype or self.data.dtype))

    def _call(self, name: str, *args):
        """Run zg_<name> (device) or zg_<name>_host (host) with the stream appended as needed."""
        lib = L.lib()
        if self.on_device:
            # scratch, per-device tables and the launch itself follow hipGetDevice(): make the owner current for the call
            with torch.cuda.device(self.data.device):
                L.check(getattr(lib, f"zg_{name}")(*args, self._stream()))
        else:
            L.check(getattr(lib, f"zg_{name}_host")(*args))

    def _same_side(self, other: "Image"):
        if self.on_device != other.on_device:
            raise ValueError("source and destination must both be host or both be device images")
        if self.on_device and self.data.device != other.data.device:
            raise ValueError(f"source is on {self.data.device}, destination on {other.data.device}: one call runs on one device")

    @staticmethod
    def _wrap(x) -> "Image":
        return x if isinstance(x, Image) else Image(x)

    def to_numpy(self) -> np.ndarray:
        return self.data.cpu().numpy() if self.on_device else self.data

    @staticmethod
    def from_numpy(a) -> "Image":
        """Image.from_numpy (bindings/python/src/image/numpy_interop.zig:114-210): zero-copy; rows may be strided, pixels of a
        row must be contiguous; TypeError for None / a wrong dtype, ValueError for a wrong shape or incompatible strides."""
        if a is None:
            raise TypeError("expected an array, got None")
        return Image(np.asarray(a))

    def to_device(self, device="cuda") -> "Image":
        if self.on_device:
            return self
        return Image(torch.from_numpy(np.ascontiguousarray(self.data)).to(device))

    # ---- file I/O (reference src/image.zig:239-287 -> src/image/format.zig:14-81; PNG only here) ----------------------
    @staticmethod
    def load_from_bytes(data: bytes, kind: Optional[str] = None, device: Optional[str] = "cuda") -> "Image":
        """Image(T).loadFromBytes: the format comes from the signature; `kind` names T ("u8" / "rgb_u8" / "rgba_u8", None
        = the file's native type). BMP / GIF are not part of this build."""
        from . import jpeg, png
        data = bytes(data)
        if data[:8] == bytes([137, 80, 78, 71, 13, 10, 26, 10]):
            return png.load_from_bytes(data, kind, None, device)
        if data[:2] == b"\xff\xd8":
            return jpeg.load_from_bytes(data, kind, None, device)
        raise L.ZignalError(L.ERR_UNSUPPORTED, "UnsupportedImageFormat (PNG and JPEG are decoded by this library)")

    @staticmethod
    def load(path: str, kind: Optional[str] = None, device: Optional[str] = "cuda") -> "Image":
        """Image(T).load: detection from the file's first bytes, as ImageFormat.detectFromPath."""
        with open(path, "rb") as f:
            return Image.load_from_bytes(f.read(), kind, device)

    def save(self, path: str) -> None:
        """Image(T).save: the format comes from the extension (case-insensitive); .bmp / .gif are UnsupportedImageFormat here."""
        from . import jpeg, png
        low = path.lower()
        if low.endswith(".png"):
            return png.save(self, path)
        if low.endswith(".jpg") or low.endswith(".jpeg"):
            return jpeg.save(self, path)
        raise L.ZignalError(L.ERR_UNSUPPORTED, "UnsupportedImageFormat (.png and .jpg / .jpeg are encoded by this library)")

    # ---- container ops (reference src/image.zig:304-392) -------------------------------------
    def has_same_shape(self, other: "Image") -> bool:
        return self.rows == other.rows and self.cols == other.cols

    def get_rectangle(self) -> Tuple[int, int, int, int]:
        """Image.getRectangle (image.zig:304-312): (l, t, r, b) = (0, 0, cols, rows)."""
        return (0, 0, self.cols, self.rows)

    def is_contiguous(self) -> bool:
        return self.cols == self.stride

    def view(self, rect: Sequence[int]) -> "Image":
        """Image.view (image.zig:332-352): rect = (l, t, r, b), clipped; shares memory."""
        l, t, r, b = (int(v) for v in rect)
        l, t, r, b = max(l, 0), max(t, 0), min(r, self.cols), min(b, self.rows)
        if l >= r or t >= b:
            return Image(self.data[0:0, 0:0])
        return Image(self.data[t:b, l:r])

    def _pixel_value(self, value):
        """One pixel of this image's type as bytes: a scalar for one channel, a sequence otherwise; None is all zeros."""
        ch = 1 if self.data.ndim == 2 else int(self.data.shape[2])
        np_dtype = np.float32 if "float" in str(self.data.dtype) else np.uint8
        if value is None:
            v = np.zeros(ch, np_dtype)
        else:
            v = np.atleast_1d(np.asarray(value, np_dtype))
            if v.shape != (ch,):
                raise ValueError(f"expected {ch} channel value(s), got {v.shape}")
        return (C.c_uint8 * v.nbytes).from_buffer_copy(v.tobytes())

    def fill(self, value) -> "Image":
        """Image.fill (image.zig:191-198): every pixel of the image (or view) becomes `value`."""
        d, v = self._desc(), self._pixel_value(value)
        self._call("fill", C.byref(d), v)
        return self

    def set_border(self, rect: Sequence[int], value=None) -> "Image":
        """Image.setBorder (image.zig:200-230): every pixel outside rect = (l, t, r, b) becomes `value` (zero by default); a rect
        that misses the image fills all of it. The reference binding raises TypeError for a missing rect."""
        if rect is None:
            raise TypeError("set_border requires a rectangle (l, t, r, b)")
        l, t, r, b = (max(int(x), 0) for x in rect)
        d, v = self._desc(), self._pixel_value(value)
        self._call("set_border", C.byref(d), (C.c_uint32 * 4)(l, t, r, b), v)
        return self

    def copy(self, dst: Optional["Image"] = None) -> "Image":
        dst = self._like() if dst is None else self._wrap(dst)
        self._same_side(dst)
        if not self.has_same_shape(dst):
            raise L.DimensionMismatch(L.ERR_DIMENSION_MISMATCH, "copy: shapes differ")
        if self.on_device:
            s, d = self._desc(), dst._desc()
            L.check(L.lib().zg_copy(C.byref(s), C.byref(d), self._stream()))
        else:
            dst.data[...] = self.data
        return dst

    # ---- filters ----------------------------------------------------------------------------
    def convolve_separable(self, kernel_x, kernel_y, border: int = BorderMode.mirror,
                           out: Optional["Image"] = None) -> "Image":
        """Image.convolveSeparable (image.zig:935-951)."""
        out = self._like() if out is None else self._wrap(out)
        self._same_side(out)
        kx, kxp = _f32_array(kernel_x)
        ky, kyp = _f32_array(kernel_y)
        s, d = self._desc(), out._desc()
        self._call("conv_separable", C.byref(s), C.byref(d), kxp, len(kx), kyp, len(ky), int(border))
        return out

    def gaussian_blur(self, sigma: float, out: Optional["Image"] = None) -> "Image":
        """Image.gaussianBlur (image.zig:954-994): sigma 0 copies, sigma < 0 raises InvalidArgument."""
        out = self._like() if out is None else self._wrap(out)
        self._same_side(out)
        s, d = self._desc(), out._desc()
        self._call("gaussian_blur", C.byref(s), C.byref(d), C.c_float(sigma))
        return out

    def convolve(self, kernel, border: int = BorderMode.mirror, out: Optional["Image"] = None) -> "Image":
        """Image.convolve (image.zig:917-932); kernel is a 2-D array."""
        out = self._like() if out is None else self._wrap(out)
        self._same_side(out)
        k, kp = _f32_array(kernel)
        if k.ndim != 2:
            raise ValueError("Kernel must be a 2D array")
        s, d = self._desc(), out._desc()
        self._call("convolve", C.byref(s), C.byref(d), kp, k.shape[0], k.shape[1], int(border))
        return out

    def box_blur(self, radius: int, out: Optional["Image"] = None) -> "Image":
        """Image.boxBlur (image.zig:635-648)."""
        out = self._like() if out is None else self._wrap(out)
        self._same_side(out)
        s, d = self._desc(), out._desc()
        self._call("box_blur", C.byref(s), C.byref(d), int(radius))
        return out

    def sharpen(self, radius: int, out: Optional["Image"] = None) -> "Image":
        """Image.sharpen (image.zig:785-801): 2 * original - boxBlur(radius)."""
        out = self._like() if out is None else self._wrap(out)
        self._same_side(out)
        s, d = self._desc(), out._desc()
        self._call("sharpen", C.byref(s), C.byref(d), int(radius))
        return out

    def integral(self):
        """Image.integral (image.zig:628-630): the summed-area planes, shape (channels, rows, cols) f32 (same side as self)."""
        ch = 1 if self.data.ndim == 2 else int(self.data.shape[2])
        s = self._desc()
        if self.on_device:
            planes = torch.empty((ch, self.rows, self.cols), dtype=torch.float32, device=self.data.device)
            L.check(L.lib().zg_integral(C.byref(s), C.cast(C.c_void_p(planes.data_ptr()), C.POINTER(C.c_float)), self._stream()))
        else:
            planes = np.empty((ch, self.rows, self.cols), np.float32)
            L.check(L.lib().zg_integral_host(C.byref(s), planes.ctypes.data_as(C.POINTER(C.c_float))))
        return planes

    def invert(self) -> "Image":
        """Image.invert (image.zig:494-513), in place."""
        s = self._desc()
        self._call("invert", C.byref(s))
        return self

    # ---- resampling -------------------------------------------------------------------------
    def resize(self, size_or_out, method: Interpolation = Interpolation.bilinear, lanczos_weights=None) -> "Image":
        """Image.resize (image.zig:523): `size_or_out` is (rows, cols) or a pre-allocated Image.
        `lanczos_weights` = (wx, wy): for Rgb(u8) / Rgba(u8) with `.lanczos`, the plane kernel's weights made by the caller
        (channel_ops.zig:446-466; arrays of out.cols x 6 and out.rows x 6 f32) — how a host with its own `sin` keeps its bits."""
        out = size_or_out if isinstance(size_or_out, Image) else self._like(int(size_or_out[0]), int(size_or_out[1]))
        self._same_side(out)
        s, d, m = self._desc(), out._desc(), method._c()
        if lanczos_weights is not None:
            if method.kind != L.INTERP_LANCZOS:
                raise ValueError("lanczos_weights only apply to Interpolation.lanczos")
            wx = np.ascontiguousarray(lanczos_weights[0], np.float32).reshape(-1)
            wy = np.ascontiguousarray(lanczos_weights[1], np.float32).reshape(-1)
            if wx.size != out.cols * 6 or wy.size != out.rows * 6:
                raise ValueError(f"lanczos_weights: expected {out.cols} x 6 and {out.rows} x 6 values")
            self._call("resize_lanczos_weights", C.byref(s), C.byref(d), wx.ctypes.data_as(L._F32P), wy.ctypes.data_as(L._F32P))
            return out
        self._call("resize", C.byref(s), C.byref(d), C.byref(m))
        return out

    def scale(self, factor: float, method: Interpolation = Interpolation.bilinear) -> "Image":
        """Image.scale (image.zig:530-541): InvalidScaleFactor / InvalidDimensions as InvalidArgument."""
        if factor <= 0:
            raise L.InvalidArgument(L.ERR_INVALID_ARGUMENT, "InvalidScaleFactor")
        f = np.float32(factor)
        new_rows = int(_round_half_away(np.float32(self.rows) * f))
        new_cols = int(_round_half_away(np.float32(self.cols) * f))
        if new_rows == 0 or new_cols == 0:
            raise L.InvalidArgument(L.ERR_INVALID_ARGUMENT, "InvalidDimensions")
        return self.resize((new_rows, new_cols), method)

    def letterbox(self, size_or_out, method: Interpolation = Interpolation.bilinear):
        """Image.letterbox (image.zig:546): returns (image, content_rect(l, t, r, b))."""
        out = size_or_out if isinstance(size_or_out, Image) else self._like(int(size_or_out[0]), int(size_or_out[1]))
        self._same_side(out)
        rect = (C.c_uint32 * 4)()
        s, d, m = self._desc(), out._desc(), method._c()
        self._call("letterbox", C.byref(s), C.byref(d), C.byref(m), rect)
        return out, tuple(rect)

    def warp(self, transform, shape_or_out=None, method: Interpolation = Interpolation.bilinear) -> "Image":
        """Image.warp (image.zig:621): `transform` exposes .kind and .coefficients()."""
        if shape_or_out is None:
            out = self._like()
        elif isinstance(shape_or_out, Image):
            out = shape_or_out
        else:
            out = self._like(int(shape_or_out[0]), int(shape_or_out[1]))
        self._same_side(out)
        coef, cp = _f32_array(transform.coefficients())
        s, d, m = self._desc(), out._desc(), method._c()
        self._call("warp", C.byref(s), C.byref(d), int(transform.kind), cp, C.byref(m))
        return out

    def rotate_bounds(self, angle: float, cos_sin: Optional[Tuple[float, float]] = None) -> Tuple[int, int]:
        """Image.rotateBounds (transforms.zig:112-148) -> (rows, cols)."""
        ca, sa = cos_sin if cos_sin is not None else _cos_sin(angle)
        r, c = C.c_uint32(), C.c_uint32()
        L.check(L.lib().zg_rotate_bounds(self.rows, self.cols, C.c_float(angle), C.c_float(ca), C.c_float(sa),
                                         C.byref(r), C.byref(c)))
        return r.value, c.value

    def rotate_into(self, out: "Image", angle: float, method: Interpolation = Interpolation.bilinear,
                    border: int = BorderMode.zero, cos_sin: Optional[Tuple[float, float]] = None) -> "Image":
        """Image.rotateInto (image.zig:566)."""
        self._same_side(out)
        ca, sa = cos_sin if cos_sin is not None else _cos_sin(angle)
        s, d, m = self._desc(), out._desc(), method._c()
        self._call("rotate_into", C.byref(s), C.byref(d), C.c_float(angle), C.c_float(ca), C.c_float(sa),
                   C.byref(m), int(border))
        return out

    def rotate(self, angle: float, method: Interpolation = Interpolation.bilinear,
               border: int = BorderMode.zero, cos_sin: Optional[Tuple[float, float]] = None) -> "Image":
        """Image.rotate (image.zig:558): new image sized by rotateBounds."""
        rows, cols = self.rotate_bounds(angle, cos_sin)
        return self.rotate_into(self._like(rows, cols), angle, method, border, cos_sin)

    def extract(self, rect: Sequence[float], angle: float = 0.0, size_or_out=None,
                method: Interpolation = Interpolation.bilinear, border: int = BorderMode.zero,
                cos_sin: Optional[Tuple[float, float]] = None) -> "Image":
        """Image.extract (image.zig:593); rect = (l, t, r, b) in source coordinates."""
        if size_or_out is None:
            l, t, r, b = rect
            out = self._like(int(_round_half_away(b - t)), int(_round_half_away(r - l)))
        elif isinstance(size_or_out, Image):
            out = size_or_out
        else:
            out = self._like(int(size_or_out[0]), int(size_or_out[1]))
        self._same_side(out)
        ca, sa = cos_sin if cos_sin is not None else _cos_sin(angle)
        ra, rp = _f32_array(rect)
        s, d, m = self._desc(), out._desc(), method._c()
        self._call("extract", C.byref(s), C.byref(d), rp, C.c_float(angle), C.c_float(ca), C.c_float(sa),
                   C.byref(m), int(border))
        return out

    def crop(self, rect: Sequence[float]) -> "Image":
        """Image.crop (image.zig:582): rounded size, out-of-bounds zero filled, bit-exact copy."""
        ra, rp = _f32_array(rect)
        r, c = C.c_uint32(), C.c_uint32()
        L.check(L.lib().zg_crop_dims(rp, C.byref(r), C.byref(c)))
        out = self._like(r.value, c.value)
        s, d = self._desc(), out._desc()
        self._call("crop", C.byref(s), C.byref(d), rp)
        return out

    def insert(self, source: "Image", rect: Sequence[float], angle: float = 0.0,
               method: Interpolation = Interpolation.bilinear, blend_mode: int = Blending.none,
               cos_sin: Optional[Tuple[float, float]] = None) -> "Image":
        """Image.insert (image.zig:606): mutates self in place."""
        source = self._wrap(source)
        self._same_side(source)
        ca, sa = cos_sin if cos_sin is not None else _cos_sin(angle)
        ra, rp = _f32_array(rect)
        s, d, m = self._desc(), source._desc(), method._c()
        self._call("insert", C.byref(s), C.byref(d), rp, C.c_float(angle), C.c_float(ca), C.c_float(sa),
                   C.byref(m), int(blend_mode))
        return self

    def flip_left_right(self) -> "Image":
        """Image.flipLeftRight (transforms.zig:28-33), in place."""
        s = self._desc()
        self._call("flip_left_right", C.byref(s))
        return self

    def flip_top_bottom(self) -> "Image":
        """Image.flipTopBottom (transforms.zig:36-44), in place."""
        s = self._desc()
        self._call("flip_top_bottom", C.byref(s))
        return self

    # ---- callers of the path -----------------------------------------------------
    def sobel(self, out: Optional["Image"] = None) -> "Image":
        """Image.sobel (image.zig:1001-1010): gradient magnitude as Image(u8), one fused kernel."""
        if out is None:
            out = self._like(dtype=torch.uint8 if self.on_device else np.uint8, channels=1)
        self._same_side(out)
        s, d = self._desc(), out._desc()
        self._call("sobel", C.byref(s), C.byref(d))
        return out

    def canny(self, sigma: float, low_threshold: float, high_threshold: float, out: Optional["Image"] = None) -> "Image":
        """Image.canny (image.zig:1047-1063): binary edge map (0 / 255) as Image(u8). Raises InvalidArgument for the
        reference's InvalidParameter / InvalidSigma / InvalidThreshold. Asynchronous on the current stream."""
        if out is None:
            out = self._like(dtype=torch.uint8 if self.on_device else np.uint8, channels=1)
        self._same_side(out)
        s, d = self._desc(), out._desc()
        self._call("canny", C.byref(s), C.byref(d), C.c_float(sigma), C.c_float(low_threshold), C.c_float(high_threshold))
        return out

    # ---- order-statistic blurs (image.zig:653-783 -> order_statistic_blur.zig) -------------------------------
    def _order_stat(self, radius, op, param, border, out):
        out = self._like() if out is None else self._wrap(out)
        self._same_side(out)
        s, d = self._desc(), out._desc()
        self._call("order_statistic_blur", C.byref(s), C.byref(d), C.c_uint32(int(radius)), int(op), C.c_double(param), int(border))
        return out

    def median_blur(self, radius: int, out: Optional["Image"] = None) -> "Image":
        return self._order_stat(radius, 0, 0.5, BorderMode.mirror, out)

    def percentile_blur(self, radius: int, percentile: float, border: int = BorderMode.mirror, out: Optional["Image"] = None) -> "Image":
        return self._order_stat(radius, 0, percentile, border, out)

    def min_blur(self, radius: int, border: int = BorderMode.mirror, out: Optional["Image"] = None) -> "Image":
        return self._order_stat(radius, 0, 0.0, border, out)

    def max_blur(self, radius: int, border: int = BorderMode.mirror, out: Optional["Image"] = None) -> "Image":
        return self._order_stat(radius, 0, 1.0, border, out)

    def midpoint_blur(self, radius: int, border: int = BorderMode.mirror, out: Optional["Image"] = None) -> "Image":
        return self._order_stat(radius, 1, 0.0, border, out)

    def alpha_trimmed_mean_blur(self, radius: int, trim_fraction: float, border: int = BorderMode.mirror, out: Optional["Image"] = None) -> "Image":
        return self._order_stat(radius, 2, trim_fraction, border, out)

    def autocontrast(self, cutoff: float = 0.0) -> "Image":
        """Image.autocontrast (image.zig:804), in place."""
        s = self._desc()
        self._call("autocontrast", C.byref(s), C.c_float(cutoff))
        return self

    def equalize(self) -> "Image":
        """Image.equalize (image.zig:824), in place."""
        s = self._desc()
        self._call("equalize", C.byref(s))
        return self

    # ---- binarisation and binary morphology (image.zig:845-914 -> binary.zig; Image(u8) only) ------------------
    def threshold_otsu(self, out: Optional["Image"] = None):
        """Image.thresholdOtsu: returns (binary image, threshold)."""
        out = self._like() if out is None else self._wrap(out)
        self._same_side(out)
        t = C.c_uint8(0)
        s, d = self._desc(), out._desc()
        self._call("threshold_otsu", C.byref(s), C.byref(d), C.byref(t))
        return out, int(t.value)

    def threshold_adaptive_mean(self, radius: int, c: float, out: Optional["Image"] = None) -> "Image":
        """Image.thresholdAdaptiveMean (radius 0 raises InvalidArgument: the reference's InvalidRadius)."""
        out = self._like() if out is None else self._wrap(out)
        self._same_side(out)
        s, d = self._desc(), out._desc()
        self._call("threshold_adaptive_mean", C.byref(s), C.byref(d), C.c_uint32(int(radius)), C.c_float(c))
        return out

    def _morph(self, kernel, iterations: int, op: int, out) -> "Image":
        out = self._like() if out is None else self._wrap(out)
        self._same_side(out)
        k = np.ascontiguousarray(kernel, np.uint8)
        if k.ndim != 2:
            raise L.InvalidArgument(L.ERR_INVALID_ARGUMENT, "morphology: the structuring element is a 2-D array")
        s, d = self._desc(), out._desc()
        self._call("morph", C.byref(s), C.byref(d), k.ctypes.data_as(C.POINTER(C.c_uint8)), int(k.shape[0]), int(k.shape[1]), int(iterations), op)
        return out

    def dilate_binary(self, kernel, iterations: int = 1, out: Optional["Image"] = None) -> "Image":
        return self._morph(kernel, iterations, 0, out)

    def erode_binary(self, kernel, iterations: int = 1, out: Optional["Image"] = None) -> "Image":
        return self._morph(kernel, iterations, 1, out)

    def open_binary(self, kernel, iterations: int = 1, out: Optional["Image"] = None) -> "Image":
        return self._morph(kernel, iterations, 2, out)

    def close_binary(self, kernel, iterations: int = 1, out: Optional["Image"] = None) -> "Image":
        return self._morph(kernel, iterations, 3, out)

    def shen_castan(self, smooth: float = 0.9, window_size: int = 7, high_ratio: float = 0.99, low_rel: float = 0.5,
                    hysteresis: bool = True, use_nms: bool = False, out: Optional["Image"] = None) -> "Image":
        """Image.shenCastan (image.zig:1015-1027) with the reference's ShenCastan option defaults; binary edge map as Image(u8)."""
        if out is None:
            out = self._like(dtype=torch.uint8 if self.on_device else np.uint8, channels=1)
        self._same_side(out)
        s, d = self._desc(), out._desc()
        self._call("shen_castan", C.byref(s), C.byref(d), C.c_float(smooth), C.c_uint32(int(window_size)), C.c_float(high_ratio), C.c_float(low_rel),
                   int(bool(hysteresis)), int(bool(use_nms)))
        return out

    def isef_smooth(self, smooth: float = 0.9, out: Optional["Image"] = None) -> "Image":
        """Diagnostics: shenCastan's smoothing stage alone (isefFilter2D, edges.zig:308-349) on a device Image(f32) or Image(u8) plane, into Image(f32) (zg_isef_smooth)."""
        if not self.on_device:
            raise ValueError("isef_smooth is a device-side diagnostic")
        if out is None:
            out = self._like(dtype=torch.float32)
        self._same_side(out)
        s, d = self._desc(), out._desc()
        self._call("isef_smooth", C.byref(s), C.byref(d), C.c_float(smooth))
        return out

    def motion_blur_linear(self, angle: float, distance: int, out: Optional["Image"] = None,
                           cos_sin: Optional[Tuple[float, float]] = None) -> "Image":
        """Image.motionBlur(.{ .linear = .{ .angle, .distance } }) (image.zig:1077, motion_blur.zig:65-236)."""
        out = self._like() if out is None else self._wrap(out)
        self._same_side(out)
        ca, sa = cos_sin if cos_sin is not None else _cos_sin(angle)
        s, d = self._desc(), out._desc()
        self._call("motion_blur_linear", C.byref(s), C.byref(d), C.c_float(angle), C.c_float(ca), C.c_float(sa), C.c_uint32(int(distance)))
        return out

    def motion_blur_radial(self, center_x: float, center_y: float, strength: float, spin: bool = False,
                           out: Optional["Image"] = None) -> "Image":
        """Image.motionBlur(.{ .radial_zoom / .radial_spin = .{ .center_x, .center_y, .strength } }) (motion_blur.zig:240-440)."""
        out = self._like() if out is None else self._wrap(out)
        self._same_side(out)
        s, d = self._desc(), out._desc()
        self._call("motion_blur_radial", C.byref(s), C.byref(d), C.c_float(center_x), C.c_float(center_y), C.c_float(strength), int(bool(spin)))
        return out

    # ---- colour -----------------------------------------------------------------------------
    def convert(self, dst_space: int, dtype=np.float32, src_space: Optional[int] = None,
                out: Optional["Image"] = None, srgb_lut=None) -> "Image":
        """Image.convert (image.zig:418): e.g. rgba_u8.convert(CS_OKLAB, np.float32), lab_f32.convert(CS_RGB, np.uint8,
        src_space=CS_LAB). A three-channel source is Rgb unless src_space says otherwise (the colour space is part of
        the Zig pixel type; here it travels beside the array)."""
        if src_space is None:
            src_space = {1: L.CS_GRAY, 3: L.CS_RGB, 4: L.CS_RGBA}[1 if self.data.ndim == 2 else self.data.shape[2]]
        ch = 1 if dst_space == L.CS_GRAY else (4 if dst_space == L.CS_RGBA else 3)
        if out is None:
            if self.on_device:
                tdtype = {np.uint8: torch.uint8, np.float32: torch.float32}[np.dtype(dtype).type]
                out = self._like(dtype=tdtype, channels=ch)
            else:
                out = self._like(dtype=np.dtype(dtype), channels=ch)
        self._same_side(out)
        lut = None
        if srgb_lut is not None:
            _keep, lut = _f32_array(srgb_lut)
        s, d = self._desc(), out._desc()
        self._call("convert", C.byref(s), int(src_space), C.byref(d), int(dst_space), lut)
        return out


    def resize_convert(self, size_or_out, dst_space: int, dtype=np.float32, method: Interpolation = Interpolation.bilinear,
                       src_space: Optional[int] = None, srgb_lut=None) -> "Image":
        """The pipeline steps [resize, convert] as one call (zg_resize_convert): `resize(size).convert(dst_space)` without the
        intermediate image wherever a fused kernel exists (Rgba(u8), bilinear -> Oklab / Xyz f32), bit-identical to the two calls."""
        if src_space is None:
            src_space = {1: L.CS_GRAY, 3: L.CS_RGB, 4: L.CS_RGBA}[1 if self.data.ndim == 2 else self.data.shape[2]]
        ch = 1 if dst_space == L.CS_GRAY else (4 if dst_space == L.CS_RGBA else 3)
        if isinstance(size_or_out, Image):
            out = size_or_out
        elif self.on_device:
            tdtype = {np.uint8: torch.uint8, np.float32: torch.float32}[np.dtype(dtype).type]
            out = self._like(int(size_or_out[0]), int(size_or_out[1]), dtype=tdtype, channels=ch)
        else:
            out = self._like(int(size_or_out[0]), int(size_or_out[1]), dtype=np.dtype(dtype), channels=ch)
        self._same_side(out)
        lut = None
        if srgb_lut is not None:
            _keep, lut = _f32_array(srgb_lut)
        s, d, m = self._desc(), out._desc(), method._c()
        self._call("resize_convert", C.byref(s), int(src_space), C.byref(d), int(dst_space), C.byref(m), lut)
        return out


def _round_half_away(v) -> float:
    v = float(v)
    return math.floor(abs(v) + 0.5) * (1.0 if v >= 0 else -1.0)


def _cos_sin(angle: float) -> Tuple[float, float]:
    """cos/sin in f32 as the caller's maths library gives them (numpy here; Zig's @cos/@sin in the
    Zig shim). Passed explicitly so the kernels never depend on a device libm."""
    a = np.float32(angle)
    return float(np.cos(a, dtype=np.float32)), float(np.sin(a, dtype=np.float32))


def gaussian_kernel(sigma: float) -> np.ndarray:
    """The taps Image.gaussianBlur builds (image.zig:973-990)."""
    lib = L.lib()
    n = lib.zg_gaussian_kernel(C.c_float(sigma), None, 0)
    if n < 0:
        L.check(-n)
    out = np.empty(n, np.float32)
    lib.zg_gaussian_kernel(C.c_float(sigma), out.ctypes.data_as(C.POINTER(C.c_float)), n)
    return out


def _plane_arrays(planes: Sequence["Image"], outs: Optional[Sequence["Image"]]):
    planes = [Image._wrap(p) for p in planes]
    outs = [p._like() for p in planes] if outs is None else [Image._wrap(o) for o in outs]
    if len(outs) != len(planes):
        raise ValueError(f"{len(planes)} source planes but {len(outs)} destination planes")
    for p, o in zip(planes, outs):
        p._same_side(o)
    n = len(planes)
    src = (L.ZgImage * max(n, 1))(*[p._desc() for p in planes])
    dst = (L.ZgImage * max(n, 1))(*[o._desc() for o in outs])
    return planes, outs, src, dst


def convolve_separable_planes(planes: Sequence["Image"], kernel_x, kernel_y, border: int = BorderMode.mirror,
                              outs: Optional[Sequence["Image"]] = None) -> List["Image"]:
    """Image.convolveSeparable (image.zig:935-951) on several planes of one shape in one device launch (zg_conv_separable_planes):
    how RGBA f32 data goes through the reference's API, which has no Rgba(f32) convolution (convolution.zig:431-435) — four
    Image(f32) planes. Host-side planes run one after the other through the host layer."""
    planes, outs, src, dst = _plane_arrays(planes, outs)
    kx, kxp = _f32_array(kernel_x)
    ky, kyp = _f32_array(kernel_y)
    if planes and planes[0].on_device:
        with torch.cuda.device(planes[0].data.device):
            L.check(L.lib().zg_conv_separable_planes(src, dst, len(planes), kxp, len(kx), kyp, len(ky), int(border), planes[0]._stream()))
    else:
        for p, o in zip(planes, outs):
            p.convolve_separable(kx, ky, border, out=o)
    return outs


def gaussian_blur_planes(planes: Sequence["Image"], sigma: float, outs: Optional[Sequence["Image"]] = None) -> List["Image"]:
    """Image.gaussianBlur (image.zig:954-994) on several planes of one shape in one device launch (zg_gaussian_blur_planes)."""
    planes, outs, src, dst = _plane_arrays(planes, outs)
    if planes and planes[0].on_device:
        with torch.cuda.device(planes[0].data.device):
            L.check(L.lib().zg_gaussian_blur_planes(src, dst, len(planes), C.c_float(sigma), planes[0]._stream()))
    else:
        for p, o in zip(planes, outs):
            p.gaussian_blur(sigma, out=o)
    return outs


def lanczos_plane_weights(src_n: int, dst_n: int) -> np.ndarray:
    """The library's own Lanczos3 plane weights of one axis (channel_ops.zig:446-466): dst_n x 6 f32."""
    out = np.empty((int(dst_n), 6), np.float32)
    L.check(L.lib().zg_lanczos_plane_weights(int(src_n), int(dst_n), out.ctypes.data_as(L._F32P)))
    return out


class ImagePyramid:
    """ImagePyramid(T) (reference src/image/pyramid.zig:11-170): level 0 is the source itself, level i is the source
    blurred with sigma_i = blur_sigma * sqrt(scale_i^2 - 1) (only if > 0.5) and resized bilinearly to trunc(dim / scale_i),
    scale_i = pow(scale_factor, i); a level below 8 x 8 truncates the pyramid. Pure composition of gaussianBlur + resize."""

    def __init__(self, levels, scale_factor: float, blur_sigma: float):
        self.levels, self.scale_factor, self.blur_sigma = levels, scale_factor, blur_sigma

    @property
    def n_levels(self) -> int:
        return len(self.levels)

    @staticmethod
    def build(source: "Image", n_levels: int, scale_factor: float, blur_sigma: float) -> "ImagePyramid":
        assert n_levels > 0 and scale_factor > 1.0 and blur_sigma > 0
        lib = L.lib()
        levels, sigmas = [source], []
        for i in range(1, n_levels):
            scale = lib.zg_pyramid_scale(C.c_float(scale_factor), i)
            r, c, sigma = C.c_uint32(), C.c_uint32(), C.c_float()
            L.check(lib.zg_pyramid_level(source.rows, source.cols, C.c_float(scale), C.c_float(blur_sigma),
                                         C.byref(r), C.byref(c), C.byref(sigma)))
            if r.value < 8 or c.value < 8:
                break
            sigmas.append(sigma.value)
            if source.on_device:
                levels.append(source._like(r.value, c.value))
            else:
                base = source.gaussian_blur(sigma.value) if sigma.value > 0.5 else source
                levels.append(base.resize((r.value, c.value), Interpolation.bilinear))
        if source.on_device and len(levels) > 1:  # ONE call for the whole pyramid: the levels fork over internal streams and join back
            descs = (L.ZgImage * (len(levels) - 1))(*[l._desc() for l in levels[1:]])
            sig = (C.c_float * len(sigmas))(*sigmas)
            sd = source._desc()
            with torch.cuda.device(source.data.device):
                L.check(lib.zg_pyramid_build(C.byref(sd), descs, sig, len(sigmas), source._stream()))
        return ImagePyramid(levels, scale_factor, blur_sigma)

    @staticmethod
    def build_default(source: "Image") -> "ImagePyramid":
        return ImagePyramid.build(source, 8, 1.2, 1.6)

    def get_scale(self, level: int) -> float:
        return L.lib().zg_pyramid_scale(C.c_float(self.scale_factor), level)

    def total_pixels(self) -> int:
        return sum(l.rows * l.cols for l in self.levels)


class ProjectiveTransform:
    """reference src/geometry/transforms.zig:197-231 (f32 matrix, project = M [x y 1]^T scaled by 1/w)."""
    kind = L.TRANSFORM_PROJECTIVE

    def __init__(self, matrix):
        self.matrix = np.asarray(matrix, np.float32).reshape(3, 3)

    @staticmethod
    def from_points(from_points, to_points) -> "ProjectiveTransform":
        """ProjectiveTransform.init with four correspondences (geometry/transforms.zig:242-263): the 8 x 8 system solved on
        the host in f64 (h22 = 1), then .as(f32). Host-side set-up, as in the reference."""
        f, t = np.asarray(from_points, np.float64).reshape(4, 2), np.asarray(to_points, np.float64).reshape(4, 2)
        a, b = np.zeros((8, 8)), np.zeros(8)
        for i in range(4):
            fx, fy, tx, ty = f[i, 0], f[i, 1], t[i, 0], t[i, 1]
            a[2 * i] = (fx, fy, 1, 0, 0, 0, -tx * fx, -tx * fy)
            a[2 * i + 1] = (0, 0, 0, fx, fy, 1, -ty * fx, -ty * fy)
            b[2 * i], b[2 * i + 1] = tx, ty
        h = np.linalg.solve(a, b)
        return ProjectiveTransform(np.append(h, 1.0).astype(np.float32))

    def coefficients(self):
        return self.matrix.reshape(9)


class AffineTransform:
    """reference src/geometry/transforms.zig:118-150 (2x2 matrix + bias)."""
    kind = L.TRANSFORM_AFFINE

    def __init__(self, matrix, bias):
        self.matrix = np.asarray(matrix, np.float32).reshape(2, 2)
        self.bias = np.asarray(bias, np.float32).reshape(2)

    def coefficients(self):
        return np.concatenate([self.matrix.reshape(4), self.bias])


class SimilarityTransform(AffineTransform):
    """reference src/geometry/transforms.zig:10-42"""
    kind = L.TRANSFORM_SIMILARITY
