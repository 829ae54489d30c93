"""The reference unit tests that the other oracle test files had not yet transcribed (found by listing every `test "..."` of
src/image/tests/*.zig against the file:line citations in tests/): boxBlur and sharpen properties, the integral-image helpers,
letterbox with every interpolation, the Shen-Castan property tests, and the container operations of Image(T). CPU only."""
import ctypes

import numpy as np
import pytest

import zignal_amd as zg


# ---- image/tests/filters.zig: boxBlur (:105-275), sharpen (:277-345) -------------------------------------------------------

def test_box_blur_zero_radius_and_border_effects(oracle):  # filters.zig:105-126, :128-154
    img = np.arange(9, dtype=np.uint8).reshape(3, 3)
    assert np.array_equal(oracle.box_blur(img, 0), img)
    dot = np.zeros((5, 5), np.uint8)
    dot[2, 2] = 255
    out = oracle.box_blur(dot, 1)
    assert out.shape == (5, 5) and out[0, 0] < out[2, 2] < 255


def test_box_blur_struct_types(oracle):  # filters.zig:156-184, :234-275
    px = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [255, 255, 255], [255, 0, 255], [0, 255, 255], [128, 128, 128], [0, 0, 0]], np.uint8)
    img = np.concatenate([px, np.full((9, 1), 255, np.uint8)], 1).reshape(3, 3, 4)
    centre = oracle.box_blur(img, 1)[1, 1]
    assert centre[0] != 255 and centre[1] != 255 and centre[2] != 255
    for size in (8, 32):
        for radius in (1, 3):
            r, c = np.mgrid[0:size, 0:size]
            img = np.stack([(255 * c) // size, np.full_like(c, 128), (255 * r) // size, np.full_like(c, 255)], -1).astype(np.uint8)
            out = oracle.box_blur(img, radius)
            assert (out[..., 3] == 255).all()
            col = out[1:size, size // 2, 0].astype(int)
            assert np.abs(np.diff(col)).max() <= 15  # "reasonable smoothness"


def test_box_blur_border_area_calculations(oracle):  # filters.zig:186-232
    assert (oracle.box_blur(np.full((12, 12), 200, np.uint8), 3) == 200).all()  # the window area shrinks at the border, the mean does not
    r, _ = np.mgrid[0:12, 0:12]
    assert oracle.box_blur(((r * 255) // 12).astype(np.uint8), 3).dtype == np.uint8


def test_sharpen_properties(oracle):  # filters.zig:277-345
    img = np.where(np.arange(5)[None, :] < 2, 64, 192).astype(np.uint8).repeat(5, 0)
    out = oracle.sharpen(img, 1)
    assert out.shape == (5, 5) and out[2, 0] <= 64 and out[2, 4] >= 192
    ramp = (np.arange(9, dtype=np.uint8) + 10).reshape(3, 3)
    assert np.array_equal(oracle.sharpen(ramp, 0), ramp)
    assert (oracle.sharpen(np.full((4, 4), 100, np.uint8), 1) == 100).all()


# ---- image/tests/integral.zig (:70-114, :116-147) ---------------------------------------------------------------------------

def test_integral_rgb_equals_rgba_with_full_alpha(oracle):
    seed, rgb = 0, np.zeros((10, 10, 3), np.uint8)
    for r in range(10):
        for c in range(10):
            seed = (seed + 17) & 255
            rgb[r, c] = (seed, (seed + 50) & 255, (seed + 100) & 255)
    rgba = np.concatenate([rgb, np.full((10, 10, 1), 255, np.uint8)], -1)
    a, b = oracle.integral(rgb), oracle.integral(rgba)
    assert np.array_equal(a[:3].view(np.uint32), b[:3].view(np.uint32))


def test_integral_window_sums(oracle):
    """Integral(T).sum over inclusive corners (integral.zig, tested at tests/integral.zig:116-147): S(r2,c2) - S(r1-1,c2) - S(r2,c1-1) + S(r1-1,c1-1)."""
    sat = oracle.integral((np.arange(9, dtype=np.uint8) + 1).reshape(3, 3))[0]

    def window(r1, c1, r2, c2):
        total = sat[r2, c2]
        if r1 > 0:
            total -= sat[r1 - 1, c2]
        if c1 > 0:
            total -= sat[r2, c1 - 1]
        if r1 > 0 and c1 > 0:
            total += sat[r1 - 1, c1 - 1]
        return float(total)
    assert window(0, 0, 2, 2) == 45 and window(0, 0, 1, 1) == 12 and window(1, 1, 2, 2) == 28 and window(1, 1, 1, 1) == 5


# ---- image/tests/resize.zig:164-200 -------------------------------------------------------------------------------------------

def test_letterbox_every_interpolation_fills_a_square(oracle):
    src = np.array([[0, 128, 255], [64, 128, 192], [128, 192, 255]], np.uint8)
    for kind in (oracle.NEAREST, oracle.BILINEAR, oracle.BICUBIC, oracle.LANCZOS):
        out = np.zeros((10, 10), np.uint8)
        l, t, r, b = oracle.letterbox(src, out, oracle.method(kind))
        assert (r - l, b - t) == (10, 10)


# ---- image/tests/shen_castan.zig:132-380 (the property tests) ------------------------------------------------------------------

def test_shen_castan_smoothing_parameter(oracle):  # :132-185
    r, c = np.mgrid[0:40, 0:40]
    dist = np.sqrt(((r - 20).astype(np.float32)) ** 2 + ((c - 20).astype(np.float32)) ** 2)
    img = np.where(dist <= 10, 200, 50 + (r + c) // 2).astype(np.uint8)
    assert (oracle.shen_castan(img, 0.7, 7, 0.9) > 0).any() and (oracle.shen_castan(img, 0.9, 7, 0.9) > 0).any()


def test_shen_castan_rgb_image(oracle):  # :187-233
    r, _ = np.mgrid[0:30, 0:30]
    img = np.zeros((30, 30, 3), np.uint8)
    img[..., 2] = np.minimum(100 + r * 3, 255)
    img[10:20, 10:20] = (200, 50, 50)
    assert (oracle.shen_castan(img, 0.8, 7, 0.9) > 0).any()


def test_shen_castan_threshold_monotonicity(oracle):  # :235-280
    img = np.zeros((40, 40), np.uint8)
    img[:20, :20] = 255
    img[20:, 20:] = 255
    low = (oracle.shen_castan(img, 0.8, 7, 0.95, 0.5) > 0).sum()
    high = (oracle.shen_castan(img, 0.8, 7, 0.999, 0.5) > 0).sum()
    assert high <= low and low > 0


def test_shen_castan_diagonal_edge(oracle):  # :282-325
    r, c = np.mgrid[0:30, 0:30]
    img = np.where(r > c + 2, 200, np.where(r + 2 < c, 50, np.clip(125 + (r - c) * 10, 50, 200))).astype(np.uint8)
    e = oracle.shen_castan(img, 0.8, 7, 0.9)
    assert any(e[i, i] > 0 or e[i, i - 1] > 0 or e[i - 1, i] > 0 for i in range(5, 25))


def test_shen_castan_window_size(oracle):  # :327-380
    r, c = np.mgrid[0:50, 0:50]
    base = np.where((r >= 15) & (r < 35) & (c >= 15) & (c < 35), 200, 50)
    img = (base + (r * 7 + c * 13) % 20).astype(np.uint8)
    assert (oracle.shen_castan(img, 0.8, 3, 0.9) > 0).any() and (oracle.shen_castan(img, 0.8, 11, 0.9) > 0).any()


# ---- image/tests/transforms.zig:14-130: the container operations, on the Python mirror's host flavour -----------------------------

def test_get_rectangle_view_and_contiguity():
    image = zg.Image(np.zeros((21, 13, 4), np.uint8))
    assert image.get_rectangle() == (0, 0, 13, 21)                       # :14-20
    view = image.view((0, 0, 8, 10))                                      # :107-118
    assert not view.is_contiguous() and image.is_contiguous()
    assert (view.cols, view.rows) == (8, 10) and view.get_rectangle() == (0, 0, 8, 10)
    assert np.shares_memory(view.data, image.data)


def test_copy_with_views_and_in_place():
    r, c = np.mgrid[0:5, 0:7]
    image = zg.Image((r * 10 + c).astype(np.uint8))
    view = image.view((1, 1, 4, 3))                                       # :22-76
    copied = view.copy(zg.Image(np.zeros((view.rows, view.cols), np.uint8)))
    assert np.array_equal(copied.data, image.data[1:3, 1:4])
    target = zg.Image(np.full((6, 8), 99, np.uint8))
    view.copy(target.view((2, 2, 5, 4)))
    assert np.array_equal(target.data[2:4, 2:5], image.data[1:3, 1:4])
    assert target.data[0, 0] == 99 and target.data[5, 7] == 99
    small = zg.Image(np.arange(9, dtype=np.uint8).reshape(3, 3))          # :78-105
    small.copy(small)
    assert np.array_equal(small.data, np.arange(9, dtype=np.uint8).reshape(3, 3))
    with pytest.raises(zg.DimensionMismatch):
        view.copy(zg.Image(np.zeros((4, 4), np.uint8)))


# ---- src/color.zig in-file tests not cited elsewhere (:1813-1865) ------------------------------------------------------------------

def test_color_known_answers_luma_invert_union_clamping(oracle):
    # "Luma calculation" (:1813-1833): the grey of a float Rgb is its luma
    def luma(r, g, b):
        return float(oracle.convert(np.array([[[r, g, b]]], np.float32), oracle.CS_RGB, oracle.CS_GRAY, np.float32, 1)[0, 0])
    assert abs(luma(1, 1, 1) - 1.0) < 1e-3 and abs(luma(0, 0, 0)) < 1e-3
    assert abs(luma(1, 0, 0) - 0.2126) < 1e-3 and abs(luma(0, 1, 0) - 0.7152) < 1e-3 and abs(luma(0, 0, 1) - 0.0722) < 1e-3
    rgba = np.array([[[1, 0, 0, 0.5]]], np.float32)  # Rgba ignores alpha
    assert abs(float(oracle.convert(rgba, oracle.CS_RGBA, oracle.CS_GRAY, np.float32, 1)[0, 0]) - 0.2126) < 1e-3
    # "Rgba invert" (:1835-1841): colour channels flip, alpha stays
    px = np.array([[[255, 255, 255, 0], [100, 150, 200, 255]]], np.uint8)
    assert oracle.invert(px.copy()).tolist() == [[[0, 0, 0, 0], [155, 105, 55, 255]]]
    # "Color union float" (:1843-1852): red is hsv (0, 100, 100)
    h, s, v = oracle.color_to([1.0, 0.0, 0.0], oracle.CS_RGB, oracle.CS_HSV, np.float32)
    assert abs(h) < 1e-3 and abs(s - 100) < 1e-3 and abs(v - 100) < 1e-3
    # "clamping out-of-range inputs" (:1854-1865): convertColor(u8, f32) and Rgb(f32).as(u8) clamp, 0.5 rounds to 128
    g = oracle.convert(np.array([[-0.5, 1.5]], np.float32), oracle.CS_GRAY, oracle.CS_GRAY, np.uint8, 1)
    assert g.tolist() == [[0, 255]]
    rgb = oracle.convert(np.array([[[1.2, -0.2, 0.5]]], np.float32), oracle.CS_RGB, oracle.CS_RGB, np.uint8, 3)
    assert rgb.tolist() == [[[255, 0, 128]]]


def test_lanczos_plane_weights_follow_channel_ops(oracle):
    """zg_lanczos_plane_weights (what zg_resize uses, and what a Zig host replaces with its own @sin): lanczosKernel of
    channel_ops.zig:446-454 at (k - 2) - frac((d + 0.5) ratio - 0.5), every step in f32, sin from the oracle's restatement."""
    import zignal_amd as zg

    l = oracle.lib()
    l.zo_sinf.restype = ctypes.c_float
    l.zo_sinf.argtypes = [ctypes.c_float]
    f32 = np.float32

    def kernel(x):
        if x == 0:
            return f32(1.0)
        if abs(x) >= 3:
            return f32(0.0)
        pi_x = f32(np.pi) * x
        return f32(f32(f32(f32(3.0) * f32(l.zo_sinf(float(pi_x)))) * f32(l.zo_sinf(float(f32(pi_x / f32(3.0)))))) / f32(pi_x * pi_x))

    for src_n, dst_n in ((16, 5), (7, 23), (4096, 1024), (100, 100)):
        got = zg.lanczos_plane_weights(src_n, dst_n)
        ratio = f32(src_n) / f32(dst_n)
        for d in list(range(min(dst_n, 40))) + [dst_n - 1]:
            s = f32(f32(f32(d) + f32(0.5)) * ratio) - f32(0.5)
            f = f32(s - np.floor(s))
            want = [kernel(f32(f32(k - 2) - f)) for k in range(6)]
            assert [w.tobytes() for w in want] == [w.tobytes() for w in got[d]], (src_n, dst_n, d, want, got[d])
