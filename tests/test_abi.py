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
