"""Pins the CPU oracle's border / convolution restatement to the reference's own known answers.

Every case cites the reference test it re-encodes (paths relative to /root/reference/src).
CPU only — no GPU, no libzignal_hip.
"""
import numpy as np
import pytest

Z, REP, MIR, WRAP = 0, 1, 2, 3


# ---- image/border.zig:65-139 ---------------------------------------------------------------------
def test_resolve_index_basic(oracle):
    for mode in (Z, REP, MIR, WRAP):
        assert oracle.resolve_index(5, 10, mode) == 5
        assert oracle.resolve_index(0, 0, mode) is None


def test_resolve_index_zero(oracle):
    for idx in (-1, -5, 10, 15):
        assert oracle.resolve_index(idx, 10, Z) is None


def test_resolve_index_replicate(oracle):
    assert oracle.resolve_index(-1, 10, REP) == 0
    assert oracle.resolve_index(-5, 10, REP) == 0
    assert oracle.resolve_index(10, 10, REP) == 9
    assert oracle.resolve_index(15, 10, REP) == 9


def test_resolve_index_mirror(oracle):
    assert oracle.resolve_index(-1, 5, MIR) == 1
    assert oracle.resolve_index(-2, 5, MIR) == 2
    assert oracle.resolve_index(5, 5, MIR) == 3
    assert oracle.resolve_index(6, 5, MIR) == 2
    assert oracle.resolve_index(-1, 1, MIR) == 0
    assert oracle.resolve_index(5, 1, MIR) == 0


def test_resolve_index_wrap(oracle):
    assert oracle.resolve_index(-1, 5, WRAP) == 4
    assert oracle.resolve_index(-6, 5, WRAP) == 4
    assert oracle.resolve_index(5, 5, WRAP) == 0
    assert oracle.resolve_index(6, 5, WRAP) == 1


# ---- image/tests/filters.zig ---------------------------------------------------------------------
IDENTITY = [[0, 0, 0], [0, 1, 0], [0, 0, 0]]
BOX = np.full((3, 3), 1.0 / 9.0, np.float32)


def test_convolve_identity_kernel(oracle):  # filters.zig:370-398
    img = (np.arange(9, dtype=np.uint8) + 10).reshape(3, 3)
    assert np.array_equal(oracle.convolve(img, IDENTITY, Z), img)


def test_convolve_blur_kernel(oracle):  # filters.zig:400-425
    img = np.zeros((5, 5), np.uint8)
    img[:, 2:] = 255
    out = oracle.convolve(img, BOX, REP)
    assert 0 < out[2, 2] < 255
    # weights round(256/9) = 28, six 255 taps: round(255*28*6/256) = 167 (28*9 = 252 != 256)
    assert out[2, 2] == 167


def test_convolve_border_modes(oracle):  # filters.zig:427-467
    img = np.zeros((3, 3), np.uint8)
    img[1, 1] = 255
    k = [[0.25, 0.25, 0], [0.25, 0.25, 0], [0, 0, 0]]
    assert oracle.convolve(img, k, REP)[0, 0] == 0
    oracle.convolve(img, k, Z)
    oracle.convolve(img, k, MIR)


def test_conv_separable_impulse(oracle):  # filters.zig:469-491
    img = np.zeros((7, 7), np.float32)
    img[3, 3] = 1.0
    g = [0.25, 0.5, 0.25]
    out = oracle.conv_separable(img, g, g, Z)
    assert out[3, 3] < 1.0 and out[3, 2] > 0 and out[3, 3] > out[3, 2]
    assert out[3, 3] == np.float32(0.25) and out[3, 2] == np.float32(0.125)  # exact in binary


def test_gaussian_blur_basic(oracle):  # filters.zig:493-519
    img = np.zeros((11, 11), np.uint8)
    img[3:8, 3:8] = 255
    out = oracle.gaussian_blur(img, 1.0)
    assert img[2, 5] == 0 and out[2, 5] > 0 and out[5, 5] > 200


def test_gaussian_blur_sigma_variations(oracle):  # filters.zig:521-546
    img = np.zeros((15, 15), np.float32)
    img[7, 7] = 1.0
    small, large = oracle.gaussian_blur(img, 0.5), oracle.gaussian_blur(img, 2.0)
    assert small[7, 7] > large[7, 7] and large[7, 5] > small[7, 5]


def test_uniform_channel_zero_border(oracle):  # filters.zig:571-600
    img = np.full((5, 5, 3), 255, np.uint8)
    out = oracle.convolve(img, BOX, Z)
    assert out[0, 0, 0] != 255 and abs(int(out[0, 0, 0]) - 113) <= 1


def test_separable_f32_view_stride(oracle):  # filters.zig:602-632
    base = (np.arange(5)[:, None] * 10 + np.arange(5)[None, :]).astype(np.float32)
    view = base[1:4, 1:4]
    out = oracle.conv_separable(view, [1.0], [1.0], Z, out=np.empty((3, 3), np.float32))
    assert np.array_equal(out, view)


def test_convolve_preserves_color_channels(oracle):  # filters.zig:662-699
    r, c = np.mgrid[0:5, 0:5]
    img = np.stack([(r * 20) % 256, (c * 20) % 256, ((r + c) * 10) % 256], -1).astype(np.uint8)
    out = oracle.convolve(img, IDENTITY, Z)
    assert np.array_equal(out[1:-1, 1:-1], img[1:-1, 1:-1])


def test_convolve_into_view(oracle):  # filters.zig:701-744
    base_src = (np.arange(6)[:, None] * 10 + np.arange(8)[None, :]).astype(np.uint8)
    base_dst = np.full((6, 8), 0xAA, np.uint8)
    oracle.convolve(base_src[1:5, 2:6], IDENTITY, Z, out=base_dst[1:5, 2:6])
    assert np.array_equal(base_dst[1:5, 2:6], base_src[1:5, 2:6])
    mask = np.ones((6, 8), bool)
    mask[1:5, 2:6] = False
    assert np.all(base_dst[mask] == 0xAA)


def test_conv_separable_into_view(oracle):  # filters.zig:746-783
    r, c = np.mgrid[0:7, 0:9]
    base_src = ((r * 7 + c * 3) % 256).astype(np.uint8)
    base_dst = np.full((7, 9), 0x55, np.uint8)
    oracle.conv_separable(base_src[2:6, 1:6], [1.0], [1.0], Z, out=base_dst[2:6, 1:6])
    assert np.array_equal(base_dst[2:6, 1:6], base_src[2:6, 1:6])
    mask = np.ones((7, 9), bool)
    mask[2:6, 1:6] = False
    assert np.all(base_dst[mask] == 0x55)


def test_gaussian_blur_preserves_color(oracle):  # filters.zig:785-815
    img = np.zeros((7, 7, 3), np.uint8)
    img[2:5, 2:5, 0] = 255
    out = oracle.gaussian_blur(img, 1.0)
    assert out[3, 3, 0] > 150 and out[3, 3, 1] < 20 and out[3, 3, 2] < 20
    e = out[2, 1]
    if e[0] > 0:
        assert e[1] < e[0] / 2 and e[2] < e[0] / 2


def test_gaussian_blur_sigma_zero_is_copy(oracle):  # filters.zig:1159-1180
    img = np.arange(25, dtype=np.float32).reshape(5, 5)
    assert np.array_equal(oracle.gaussian_blur(img, 0.0), img)


def test_gaussian_blur_negative_sigma(oracle):  # image.zig:970 error.InvalidSigma
    with pytest.raises(ValueError):
        oracle.gaussian_blur(np.zeros((3, 3), np.float32), -1.0)


def test_convolve_issue_255(oracle):  # filters.zig:1302-1342
    img = np.ones((10, 20), np.uint8)
    out = oracle.convolve(img, [[1, 1, 1], [1, 0, 1], [1, 1, 1]], Z, out=np.full((10, 20), 0xAA, np.uint8))
    assert np.all(out[1:9, 0] == 5) and np.all(out[1:9, 1] == 8)


# ---- the integer taps gaussianBlur(0.6) must produce ---------------------------------
def test_gaussian_kernel_int_taps(oracle):
    k = oracle.gaussian_kernel(0.6)
    assert len(k) == 5
    assert [int(np.round(v * 256)) for v in k] == [1, 42, 170, 42, 1]
    k05 = oracle.gaussian_kernel(0.5)
    assert sum(int(np.round(v * 256)) for v in k05) == 255  # the documented alpha-darkening quirk
    assert abs(float(k.sum()) - 1.0) < 1e-6


def test_u8_rgba_matches_plane_path(oracle):
    """Struct path == per-plane path (channels independent), incl. a uniform alpha channel."""
    img = oracle.synth_u8(7, (13, 17, 4))
    img[..., 3] = 255
    k = oracle.gaussian_kernel(0.6)
    out = oracle.conv_separable(img, k, k, MIR)
    for ch in range(4):
        plane = np.ascontiguousarray(img[..., ch])
        assert np.array_equal(out[..., ch], oracle.conv_separable(plane, k, k, MIR))
