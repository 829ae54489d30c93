"""CPU-side checks of the drop-in boundary: libzignal_hip.so loads, exports every symbol that
include/zignal_hip.h declares, and the host-side mirror validates arguments. No compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest

import zignal_amd as zg
from zignal_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "zignal_hip.h")).read()
    return sorted(set(re.findall(r"ZG_API\s+[\w\s\*]+?\b(zg_\w+)\s*\(", text)))


def test_library_builds_and_loads():
    assert os.path.exists(L.LIB_PATH), "run __graft_entry__.build() first"
    lib = zg.lib()
    assert lib.zg_version() >= 100


def test_every_declared_symbol_is_exported_and_bound():
    declared = _header_symbols()
    assert len(declared) >= 48
    raw = ctypes.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/zignal_hip.h but not exported"
    # the Python binding covers the whole header, nothing more
    assert sorted(L.EXPORTED_SYMBOLS) == declared


def test_zig_shim_declares_the_whole_header():
    """zig/zignal_hip.zig cannot be compiled here (no Zig toolchain): at least keep its extern block in step with the header."""
    shim = open(os.path.join(ROOT, "zig", "zignal_hip.zig")).read()
    externs = set(re.findall(r"pub extern fn (zg_\w+)\(", shim))
    assert externs == set(_header_symbols()), sorted(externs ^ set(_header_symbols()))


def test_enum_ordinals_follow_reference_declaration_order():
    # border.zig:10-18, interpolation.zig:53-68
    assert (zg.BorderMode.zero, zg.BorderMode.replicate, zg.BorderMode.mirror, zg.BorderMode.wrap) == (0, 1, 2, 3)
    kinds = [zg.Interpolation.nearest.kind, zg.Interpolation.bilinear.kind, zg.Interpolation.bicubic.kind,
             zg.Interpolation.catmull_rom.kind, zg.Interpolation.mitchell_default.kind, zg.Interpolation.lanczos.kind]
    assert kinds == [0, 1, 2, 3, 4, 5]


def test_pixel_sizes():
    lib = zg.lib()
    assert [lib.zg_pixel_size(p) for p in range(6)] == [1, 4, 3, 4, 12, 16]
    assert lib.zg_pixel_size(99) == 0


def test_image_descriptor_mirrors_image_t():
    base = np.zeros((6, 8, 4), np.uint8)
    img = zg.Image(base)
    assert (img.rows, img.cols, img.stride, img.pixel) == (6, 8, 8, L.PIXEL_RGBA_U8)
    v = img.view((2, 1, 6, 5))  # l, t, r, b
    assert (v.rows, v.cols, v.stride) == (4, 4, 8) and not v.is_contiguous()
    assert v.data.ctypes.data == base.ctypes.data + (1 * 8 + 2) * 4
    assert zg.Image(np.zeros((3, 5), np.float32)).pixel == L.PIXEL_F32
    with pytest.raises(TypeError):
        zg.Image(np.zeros((3, 5), np.int16))


def test_gaussian_kernel_matches_reference_taps():
    # gaussianBlur(0.6) -> 5 taps, integer taps [1, 42, 170, 42, 1]; host-only call
    k = zg.gaussian_kernel(0.6)
    assert len(k) == 5 and [int(np.round(v * 256)) for v in k] == [1, 42, 170, 42, 1]
    with pytest.raises(zg.InvalidArgument):
        zg.gaussian_kernel(-1.0)


def test_library_taps_equal_oracle_taps(oracle):
    for sigma in (0.3, 0.5, 0.6, 1.0, 1.7, 2.5, 4.0):
        assert np.array_equal(zg.gaussian_kernel(sigma), oracle.gaussian_kernel(sigma))


# ---- argument validation happens before any HIP call, so it is testable without a GPU -----------------------
def _img(rows, cols, pixel, data=0x1000, stride=None):
    return L.ZgImage(data, cols if stride is None else stride, rows, cols, pixel)


def _status(fn, *args):
    return fn(*args)


def test_error_convention_without_a_gpu():
    lib = zg.lib()
    k = (ctypes.c_float * 3)(0.25, 0.5, 0.25)
    a, b = _img(4, 4, L.PIXEL_U8), _img(4, 5, L.PIXEL_U8)
    # error.DimensionMismatch (src/image.zig:947, :962, :927, :636)
    assert lib.zg_conv_separable(ctypes.byref(a), ctypes.byref(b), k, 3, k, 3, 0, None) == L.ERR_DIMENSION_MISMATCH
    assert lib.zg_gaussian_blur(ctypes.byref(a), ctypes.byref(b), ctypes.c_float(1.0), None) == L.ERR_DIMENSION_MISMATCH
    assert lib.zg_convolve(ctypes.byref(a), ctypes.byref(b), k, 1, 3, 0, None) == L.ERR_DIMENSION_MISMATCH
    assert lib.zg_box_blur(ctypes.byref(a), ctypes.byref(b), 1, None) == L.ERR_DIMENSION_MISMATCH
    # error.InvalidSigma (src/image.zig:970)
    assert lib.zg_gaussian_blur(ctypes.byref(a), ctypes.byref(a), ctypes.c_float(-0.5), None) == L.ERR_INVALID_ARGUMENT
    assert b"InvalidSigma" in lib.zg_last_error()
    # malformed descriptors
    assert lib.zg_conv_separable(ctypes.byref(_img(4, 4, 99)), ctypes.byref(a), k, 3, k, 3, 0, None) == L.ERR_INVALID_ARGUMENT
    assert lib.zg_conv_separable(ctypes.byref(_img(4, 4, L.PIXEL_U8, stride=3)), ctypes.byref(a), k, 3, k, 3, 0, None) == L.ERR_INVALID_ARGUMENT
    assert lib.zg_conv_separable(ctypes.byref(_img(4, 4, L.PIXEL_U8, data=None)), ctypes.byref(a), k, 3, k, 3, 0, None) == L.ERR_INVALID_ARGUMENT
    assert lib.zg_conv_separable(ctypes.byref(a), ctypes.byref(a), k, 3, k, 3, 7, None) == L.ERR_INVALID_ARGUMENT  # border
    assert lib.zg_conv_separable(ctypes.byref(a), ctypes.byref(a), k, 0, k, 3, 0, None) == L.ERR_INVALID_ARGUMENT  # empty kernel
    misaligned = _img(4, 4, L.PIXEL_RGBA_F32, data=0x1004)
    assert lib.zg_conv_separable(ctypes.byref(misaligned), ctypes.byref(misaligned), k, 3, k, 3, 0, None) == L.ERR_INVALID_ARGUMENT
    # pixel types must agree; unsupported combinations say so
    f = _img(4, 4, L.PIXEL_F32)
    assert lib.zg_conv_separable(ctypes.byref(a), ctypes.byref(f), k, 3, k, 3, 0, None) == L.ERR_INVALID_ARGUMENT
    m = L.ZgMethod(9, 0, 0, None)
    assert lib.zg_resize(ctypes.byref(a), ctypes.byref(b), ctypes.byref(m), None) == L.ERR_INVALID_ARGUMENT
    rgb = _img(4, 4, L.PIXEL_RGB_U8)
    assert lib.zg_convert(ctypes.byref(a), L.CS_RGB, ctypes.byref(rgb), L.CS_RGB, None, None) == L.ERR_INVALID_ARGUMENT  # layout != space
    assert lib.zg_convert(ctypes.byref(rgb), L.CS_RGB, ctypes.byref(rgb), L.CS_OKLAB, None, None) == L.ERR_UNSUPPORTED   # Oklab needs floats
    # the detectors' options (ShenCastan.zig:35-45) and the smoothing stage's diagnostic entry point
    f2 = _img(4, 4, L.PIXEL_F32, data=0x2000)
    assert lib.zg_isef_smooth(ctypes.byref(f), ctypes.byref(a), ctypes.c_float(0.9), None) == L.ERR_INVALID_ARGUMENT      # the plane comes out as f32
    assert lib.zg_isef_smooth(ctypes.byref(f), ctypes.byref(f2), ctypes.c_float(1.0), None) == L.ERR_INVALID_ARGUMENT
    assert b"InvalidBParameter" in lib.zg_last_error()
    assert lib.zg_isef_smooth(ctypes.byref(f), ctypes.byref(_img(4, 5, L.PIXEL_F32, data=0x2000)), ctypes.c_float(0.9), None) == L.ERR_DIMENSION_MISMATCH
    assert lib.zg_shen_castan(ctypes.byref(a), ctypes.byref(a), ctypes.c_float(0.9), 4, ctypes.c_float(0.99), ctypes.c_float(0.5), 1, 0, None) == L.ERR_INVALID_ARGUMENT
    assert b"WindowSizeMustBeOdd" in lib.zg_last_error()
    # empty images are legal and do nothing (Image.empty)
    e = _img(0, 0, L.PIXEL_U8, data=None)
    assert lib.zg_conv_separable(ctypes.byref(e), ctypes.byref(e), k, 3, k, 3, 0, None) == L.OK
    assert lib.zg_flip_left_right(ctypes.byref(e), None) == L.OK


def test_crop_and_rotate_bounds_host_arithmetic():
    lib = zg.lib()
    r, c = ctypes.c_uint32(), ctypes.c_uint32()
    rect = (ctypes.c_float * 4)(1.2, 2.6, 11.7, 9.4)  # l, t, r, b -> round(height), round(width)
    assert lib.zg_crop_dims(rect, ctypes.byref(r), ctypes.byref(c)) == 0 and (r.value, c.value) == (7, 11)  # 6.8 -> 7, 10.5 -> 11 (half away)
    import math
    for angle, want in ((0.0, (3, 4)), (math.pi / 2, (4, 3)), (math.pi, (3, 4)), (3 * math.pi / 2, (4, 3)), (2 * math.pi, (3, 4))):
        lib.zg_rotate_bounds(3, 4, ctypes.c_float(angle), ctypes.c_float(math.cos(angle)), ctypes.c_float(math.sin(angle)), ctypes.byref(r), ctypes.byref(c))
        assert (r.value, c.value) == want  # tests/transforms.zig:160-229
    lib.zg_rotate_bounds(10, 10, ctypes.c_float(math.pi / 4), ctypes.c_float(math.cos(math.pi / 4)), ctypes.c_float(math.sin(math.pi / 4)), ctypes.byref(r), ctypes.byref(c))
    assert r.value > 10 and c.value > 10


def test_python_mirror_validation():
    img = zg.Image(np.zeros((4, 4), np.uint8))
    with pytest.raises(zg.InvalidArgument):
        img.scale(-1.0)
    with pytest.raises(zg.DimensionMismatch):
        img.copy(zg.Image(np.zeros((4, 5), np.uint8)))
    assert img.view((10, 10, 20, 20)).rows == 0  # no overlap -> Image.empty
    assert np.array_equal(img.copy().data, img.data)


def test_ctypes_signatures_match_header_arity():
    """Every prototype in include/zignal_hip.h has the same number of parameters as its ctypes binding."""
    import re
    hdr = open(os.path.join(ROOT, "include", "zignal_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = re.findall(r"ZG_API\s+[\w\s\*]+?\b(zg_\w+)\s*\(([^;]*?)\)\s*;", hdr, flags=re.S)
    assert len(protos) >= 70
    for name, args in protos:
        args = args.strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        assert name in L._SIGNATURES, name
        assert len(L._SIGNATURES[name]) == n, f"{name}: header has {n} parameters, _lib.py binds {len(L._SIGNATURES[name])}"
