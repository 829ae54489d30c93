"""Pins the oracle's interpolation / resize / transform / integral / colour restatement to the known
answers of the reference's own unit tests (paths relative to /root/reference/src). CPU only."""
import math

import numpy as np
import pytest

Z, REP, MIR, WRAP = 0, 1, 2, 3


def M(oracle, name, b=0.0, c=0.0):
    return oracle.method(getattr(oracle, name.upper()), b, c)


def gradient(rows, cols):
    r, c = np.mgrid[0:rows, 0:cols]
    return np.minimum(255, (r + c) * 255 // (rows + cols - 2)).astype(np.uint8)


def checker(rows, cols):
    r, c = np.mgrid[0:rows, 0:cols]
    return np.where((r + c) % 2 == 0, 0, 255).astype(np.uint8)


# ---- image/tests/interpolation.zig ----------------------------------------------------------------
def test_nearest_exact_and_rounding(oracle):  # :36-70
    img = gradient(10, 10)
    for p in (0, 5, 9):
        assert oracle.interpolate(img, p, p, M(oracle, "nearest"), MIR) == img[p, p]
    ck = checker(10, 10)
    assert oracle.interpolate(ck, 0.4, 0.4, M(oracle, "nearest"), MIR) == 0
    assert oracle.interpolate(ck, 0.6, 0.6, M(oracle, "nearest"), MIR) == 0
    assert oracle.interpolate(ck, 1.5, 0.5, M(oracle, "nearest"), MIR) == 255


def test_bilinear_exact_and_midpoints(oracle):  # :72-108
    img = gradient(10, 10)
    assert oracle.interpolate(img, 0, 0, M(oracle, "bilinear"), MIR) == img[0, 0]
    assert oracle.interpolate(img, 5, 5, M(oracle, "bilinear"), MIR) == img[5, 5]
    cols = np.tile(np.array([0, 100, 200], np.uint8), (3, 1))
    assert oracle.interpolate(cols, 0.5, 0, M(oracle, "bilinear"), MIR) == 50
    assert oracle.interpolate(cols, 0.5, 0.5, M(oracle, "bilinear"), MIR) == 50


@pytest.mark.parametrize("name", ["bicubic", "catmull_rom"])
def test_cubic_exact_pixels(oracle, name):  # :110-134
    img = gradient(10, 10)
    for p in (2, 5):
        assert oracle.interpolate(img, p, p, M(oracle, name), MIR) == img[p, p]


def test_lanczos_and_mitchell_exact_pixels_within_one(oracle):  # :136-163
    img = gradient(10, 10)
    for p in (3, 5):
        assert abs(int(oracle.interpolate(img, p, p, M(oracle, "lanczos"), MIR)) - int(img[p, p])) <= 1
    assert abs(int(oracle.interpolate(img, 2, 2, M(oracle, "mitchell", 1 / 3, 1 / 3), MIR)) - int(img[2, 2])) <= 1
    assert abs(int(oracle.interpolate(img, 5, 5, M(oracle, "mitchell", 0, 0), MIR)) - int(img[5, 5])) <= 1


def test_boundaries_never_null_with_mirror(oracle):  # :182-239
    img = gradient(10, 10)
    for x, y in ((-0.4, 0), (9.4, 9.4), (-1, 0), (0, -1), (10, 0), (0, 10)):
        assert oracle.interpolate(img, x, y, M(oracle, "nearest"), MIR) is not None
    for x, y in ((0, 0), (8.9, 8.9), (9.1, 9.1), (-0.1, 0)):
        assert oracle.interpolate(img, x, y, M(oracle, "bilinear"), MIR) is not None
    for x, y in ((1, 1), (7.9, 7.9), (0.5, 0.5), (8.1, 8.1)):
        assert oracle.interpolate(img, x, y, M(oracle, "bicubic"), MIR) is not None
    for x, y in ((2, 2), (6.9, 6.9), (1.5, 1.5), (7.1, 7.1)):
        assert oracle.interpolate(img, x, y, M(oracle, "lanczos"), MIR) is not None


def test_rgb_interpolation(oracle):  # :241-268
    r, c = np.mgrid[0:4, 0:4]
    img = np.stack([r * 85, c * 85, np.full_like(r, 128)], -1).astype(np.uint8)
    assert np.array_equal(oracle.interpolate(img, 1.6, 1.4, M(oracle, "nearest"), MIR), img[1, 2])
    assert list(oracle.interpolate(img, 0.5, 0.5, M(oracle, "bilinear"), MIR)) == [43, 43, 128]
    assert oracle.interpolate(img, 1.5, 1.5, M(oracle, "mitchell", 0, 0), MIR) is not None


def test_resize_preserves_value_range(oracle):  # :270-305
    r, c = np.mgrid[0:4, 0:4]
    src = ((r + c) * 40).astype(np.uint8)
    dst = oracle.resize(src, (8, 8), M(oracle, "bilinear"))
    assert dst.min() >= 0 and dst.max() <= 240


def test_catmull_rom_no_overshoot(oracle):  # :307-331
    r, c = np.mgrid[0:5, 0:5]
    img = (50 + (r + c) * 20).astype(np.uint8)
    rr = np.float32(1.5)
    while rr < 3.5:
        cc = np.float32(1.5)
        while cc < 3.5:
            v = oracle.interpolate(img, float(cc), float(rr), M(oracle, "catmull_rom"), MIR)
            assert 50 <= v <= 200
            cc += np.float32(0.1)
        rr += np.float32(0.1)


def test_float_image(oracle):  # :333-353
    r, c = np.mgrid[0:4, 0:4]
    img = (r * 0.25 + c * 0.25).astype(np.float32)
    assert abs(float(oracle.interpolate(img, 1.5, 1.5, M(oracle, "bilinear"), MIR)) - 0.75) < 1e-3
    assert oracle.interpolate(img, 1.5, 1.5, M(oracle, "bicubic"), MIR) is not None


def test_bilinear_exact_linear(oracle):  # :422-449
    img = np.array([[0, 100], [50, 150]], np.uint8)
    bl = M(oracle, "bilinear")
    assert oracle.interpolate(img, 0.5, 0, bl, MIR) == 50
    assert oracle.interpolate(img, 0, 0.5, bl, MIR) == 25
    assert oracle.interpolate(img, 0.5, 0.5, bl, MIR) == 75
    assert oracle.interpolate(img, 0.25, 0, bl, MIR) == 25


def test_nearest_discontinuity(oracle):  # :451-470
    img = np.array([[0, 255], [100, 200]], np.uint8)
    assert oracle.interpolate(img, 0.49, 0, M(oracle, "nearest"), MIR) == 0
    assert oracle.interpolate(img, 0.51, 0, M(oracle, "nearest"), MIR) == 255


def test_symmetry(oracle):  # :472-493
    r, c = np.mgrid[0:5, 0:5]
    img = np.minimum(255, (np.abs(r - 2) + np.abs(c - 2)) * 50).astype(np.uint8)
    bl = M(oracle, "bilinear")
    assert oracle.interpolate(img, 1.5, 2, bl, MIR) == oracle.interpolate(img, 2.5, 2, bl, MIR)
    assert oracle.interpolate(img, 2, 1.5, bl, MIR) == oracle.interpolate(img, 2, 2.5, bl, MIR)


def test_mitchell_parameter_effects(oracle):  # :495-529
    img = np.full((6, 6), 50, np.uint8)
    img[2:4] = 200
    vals = [oracle.interpolate(img, 2.5, 1.8, M(oracle, "mitchell", b, c), MIR) for b, c in ((0, 0), (1, 0), (0, 0.75))]
    assert not (vals[0] == vals[1] == vals[2])


def test_lanczos_weight_normalisation(oracle):  # :531-548
    img = np.full((8, 8), 128, np.uint8)
    assert oracle.interpolate(img, 4.3, 4.7, M(oracle, "lanczos"), MIR) == 128


def test_single_pixel_image(oracle):  # :587-601
    img = np.array([[42]], np.uint8)
    for name in ("nearest", "bilinear", "bicubic", "catmull_rom", "lanczos"):
        assert oracle.interpolate(img, 0, 0, M(oracle, name), MIR) == 42
    assert oracle.interpolate(img, 0, 0, M(oracle, "mitchell", 0, 0), MIR) == 42


def test_clamping_under_overshoot(oracle):  # :355-420, :603-632
    ck = checker(8, 8)
    for m in (M(oracle, "bicubic"), M(oracle, "catmull_rom"), M(oracle, "lanczos"), M(oracle, "mitchell", 0, 0),
              M(oracle, "mitchell", 0, 0.75)):
        for y in np.arange(2.0, 6.0, 0.3):
            for x in np.arange(2.0, 6.0, 0.3):
                v = oracle.interpolate(ck, float(x), float(y), m, MIR)
                assert 0 <= v <= 255


# ---- image/tests/resize.zig ------------------------------------------------------------------------
def test_letterbox_wide_to_square(oracle):  # :12-47
    r, c = np.mgrid[0:4, 0:8]
    src = (r * 20 + c * 10).astype(np.uint8)
    out = np.full((6, 6), 9, np.uint8)
    l, t, rr, b = oracle.letterbox(src, out, M(oracle, "bilinear"))
    assert (rr - l, b - t, l, t) == (6, 3, 0, 1)
    assert np.all(out[:t] == 0) and np.all(out[b:] == 0)


def test_letterbox_tall_to_wide(oracle):  # :48-94
    src = np.zeros((9, 3, 3), np.uint8)
    src[:, 0, 0] = 255
    src[:, 1, 1] = 255
    src[:, 2, 2] = 255
    out = np.full((4, 12, 3), 7, np.uint8)
    l, t, r, b = oracle.letterbox(src, out, M(oracle, "nearest"))
    assert (r - l, b - t, l) == (1, 4, 5)
    assert np.all(out[:, :l] == 0) and np.all(out[:, r:] == 0)


def test_letterbox_edge_cases(oracle):  # :96-161
    src = np.zeros((5, 5), np.uint8)
    assert oracle.letterbox(src, np.zeros((0, 10), np.uint8), M(oracle, "nearest")) == (0, 0, 0, 0)
    r, c = np.mgrid[0:4, 0:6]
    srcf = (r * 10 + c + 1.0).astype(np.float32)
    assert oracle.letterbox(srcf, np.zeros((8, 12), np.float32), M(oracle, "bicubic")) == (0, 0, 12, 8)
    one = np.array([[128]], np.uint8)
    out = np.zeros((10, 10), np.uint8)
    assert oracle.letterbox(one, out, M(oracle, "nearest")) == (0, 0, 10, 10)
    assert np.all(out == 128)


def test_letterbox_extreme_aspect(oracle):  # :202-256
    out = np.zeros((64, 64), np.uint8)
    assert oracle.letterbox(np.full((2, 32), 200, np.uint8), out, M(oracle, "bilinear")) == (0, 30, 64, 34)
    assert oracle.letterbox(np.full((32, 2), 100, np.uint8), out, M(oracle, "bicubic")) == (30, 0, 34, 64)


def test_resize_4to1_bilinear_is_floor_of_mean_of_four(oracle):
    # ratio 4 -> taps at 4d+1, 4d+2 with fx = fy = 128 -> floor((tl+tr+bl+br)/4)
    src = oracle.synth_u8(3, (64, 64, 4))
    out = oracle.resize(src, (16, 16), M(oracle, "bilinear"))
    s = src.astype(np.int32)
    want = (s[1::4, 1::4] + s[1::4, 2::4] + s[2::4, 1::4] + s[2::4, 2::4]) // 4
    assert np.array_equal(out, want.astype(np.uint8))


def test_resize_same_size_is_copy_and_nearest_is_exact(oracle):
    src = oracle.synth_u8(4, (9, 11, 3))
    assert np.array_equal(oracle.resize(src, (9, 11), M(oracle, "lanczos")), src)
    up = oracle.resize(src, (18, 22), M(oracle, "nearest"))
    assert np.array_equal(up[::2, ::2], src) and np.array_equal(up[1::2, 1::2], src)


# ---- image/tests/transforms.zig --------------------------------------------------------------------
def pattern5():
    r, c = np.mgrid[0:5, 0:5]
    return (r * 10 + c).astype(np.uint8)


def test_extract_basic_and_90deg(oracle):  # :231-278
    img = pattern5()
    out0 = oracle.extract(img, np.zeros((3, 3), np.uint8), (1, 1, 3, 3), 0.0, M(oracle, "nearest"), MIR)
    assert out0.tolist() == [[11, 12, 13], [21, 22, 23], [31, 32, 33]]
    out90 = oracle.extract(img, np.zeros((3, 3), np.uint8), (1, 1, 3, 3), math.pi / 2, M(oracle, "nearest"), MIR)
    assert out90.tolist() == [[13, 23, 33], [12, 22, 32], [11, 21, 31]]


def test_extract_single_pixel_axes(oracle):  # :280-315
    img = pattern5()
    n = M(oracle, "nearest")
    assert oracle.extract(img, np.zeros((1, 1), np.uint8), (1, 1, 3, 3), 0.0, n, MIR).tolist() == [[22]]
    assert oracle.extract(img, np.zeros((1, 3), np.uint8), (1, 1, 3, 3), 0.0, n, MIR).tolist() == [[21, 22, 23]]
    assert oracle.extract(img, np.zeros((3, 1), np.uint8), (1, 1, 3, 3), 0.0, n, MIR).tolist() == [[12], [22], [32]]


def test_rotate_dimensions(oracle):  # :160-229
    assert oracle.rotate_bounds(3, 4, 0.0) == (3, 4)
    assert oracle.rotate_bounds(3, 4, math.pi / 2) == (4, 3)
    assert oracle.rotate_bounds(3, 4, math.pi) == (3, 4)
    assert oracle.rotate_bounds(3, 4, 3 * math.pi / 2) == (4, 3)
    r, c = oracle.rotate_bounds(10, 10, math.pi / 4)
    assert r > 10 and c > 10


def test_rotate_orthogonal_are_permutations(oracle):
    img = np.arange(12, dtype=np.uint8).reshape(3, 4)
    bl = M(oracle, "bilinear")
    assert np.array_equal(oracle.rotate(img, 0.0, bl, MIR), img)
    assert np.array_equal(oracle.rotate(img, math.pi / 2, bl, MIR), np.rot90(img, 1))
    assert np.array_equal(oracle.rotate(img, math.pi, bl, MIR), np.rot90(img, 2))
    assert np.array_equal(oracle.rotate(img, 3 * math.pi / 2, bl, MIR), np.rot90(img, 3))


def test_insert_extract_round_trip(oracle):  # :317-380
    r, c = np.mgrid[0:64, 0:64]
    source = ((r + c) % 256).astype(np.uint8)
    cases = [((10, 10, 50, 50), 0.0, 40, "bilinear"), ((15, 15, 45, 45), math.pi / 4, 30, "bilinear"),
             ((20, 20, 40, 40), 0.0, 40, "bicubic")]
    for rect, angle, size, name in cases:
        m = M(oracle, name)
        ext = oracle.extract(source, np.zeros((size, size), np.uint8), rect, angle, m, MIR)
        canvas = oracle.insert(np.zeros((64, 64), np.uint8), ext, rect, angle, m, 0)
        cx, cy = (rect[0] + rect[2]) * 0.5, (rect[1] + rect[3]) * 0.5
        cs = min(rect[2] - rect[0], rect[3] - rect[1]) * 0.6
        sl = (slice(int(cy - cs / 2), int(cy + cs / 2)), slice(int(cx - cs / 2), int(cx + cs / 2)))
        err = np.abs(source[sl].astype(np.int32) - canvas[sl].astype(np.int32)).mean()
        assert err < 25


def test_insert_blending(oracle):  # :382-406 (expected = base.blend(overlay, .normal), src/blending.zig)
    base = np.array([[[0, 0, 255, 255]]], np.uint8)
    overlay = np.array([[[255, 0, 0, 128]]], np.uint8)
    n = M(oracle, "nearest")
    assert np.array_equal(oracle.insert(base.copy(), overlay, (0, 0, 1, 1), 0.0, n, 0), overlay)
    got = oracle.insert(base.copy(), overlay, (0, 0, 1, 1), 0.0, n, 1)
    # opaque base, alpha 128/255: out = overlay*a + base*(1-a), alpha stays 1
    a = 128 / 255
    assert got[0, 0].tolist() == [round(255 * a), 0, round(255 * (1 - a)), 255]


def test_extract_from_empty_image(oracle):  # :408-425
    empty = np.zeros((0, 0), np.uint8)
    for border in (REP, WRAP):
        out = oracle.extract(empty, np.full((2, 2), 9, np.uint8), (0, 0, 2, 2), 0.0, M(oracle, "nearest"), border)
        assert out[0, 0] == 0


def test_flips(oracle):  # :427-456
    a = np.array([[1, 2, 3], [4, 5, 6]], np.uint8)
    assert oracle.flip_left_right(a.copy()).tolist() == [[3, 2, 1], [6, 5, 4]]
    b = np.array([[1, 2], [3, 4], [5, 6]], np.uint8)
    assert oracle.flip_top_bottom(b.copy()).tolist() == [[5, 6], [3, 4], [1, 2]]


def test_crop_is_exact_copy_with_zero_fill(oracle):
    img = pattern5()
    assert np.array_equal(oracle.crop(img, (1, 1, 4, 3)), img[1:3, 1:4])
    out = oracle.crop(img, (-1, -1, 2, 2))
    assert out.shape == (3, 3) and np.array_equal(out[1:, 1:], img[:2, :2]) and not out[0].any() and not out[:, 0].any()


# ---- geometry/transforms.zig:294-520 ---------------------------------------------------------------
def test_projective_four_point_solve(oracle):
    src = [(0, 0), (4095, 0), (0, 4095), (4095, 4095)]
    dst = [(200, 120), (3900, 60), (90, 3980), (4000, 4050)]
    m = oracle.homography_from_4pts(src, dst)
    for (x, y), (u, v) in zip(src, dst):
        px, py = oracle.project(oracle.PROJECTIVE, m, x, y)
        assert abs(px - u) < 0.05 and abs(py - v) < 0.05
    assert oracle.project(oracle.AFFINE, [2, 0, 0, 3, 5, 7], 1.5, 2.0) == (8.0, 13.0)


# ---- image/tests/integral.zig, filters.zig box blur ------------------------------------------------
def test_box_blur_known_answers(oracle):
    assert np.all(oracle.box_blur(np.full((10, 10), 128, np.uint8), 2) == 128)  # filters.zig:87-103
    img = oracle.synth_u8(5, (8, 8))
    assert np.array_equal(oracle.box_blur(img, 0), img)  # :105-126
    rgba = np.full((12, 12, 4), 255, np.uint8)
    rgba[..., :3] = oracle.synth_u8(6, (12, 12, 3))
    assert np.all(oracle.box_blur(rgba, 3)[..., 3] == 255)  # :234-275
    ones = np.ones((21, 13), np.uint8)  # integral.zig:11-35: SAT of ones is (r+1)(c+1)
    assert np.all(oracle.box_blur(ones, 4) == 1)
    # interior = round(sum9 / 9); corner window is clipped to 2x2
    p = np.arange(1, 10, dtype=np.uint8).reshape(3, 3)
    out = oracle.box_blur(p, 1)
    assert out[1, 1] == 5 and out[0, 0] == 3 and out[2, 2] == 7  # 45/9, 12/4, 28/4


# ---- color.zig:1556-1583 ---------------------------------------------------------------------------
def test_gray_conversions(oracle):
    rgb = np.array([[[128, 128, 128], [255, 0, 0]]], np.uint8)
    g = oracle.convert(rgb, oracle.CS_RGB, oracle.CS_GRAY, np.uint8, 1)
    assert g.tolist() == [[128, 54]]
    assert oracle.convert(np.array([[128]], np.uint8), oracle.CS_GRAY, oracle.CS_RGB, np.uint8, 3).tolist() == [[[128, 128, 128]]]
    assert oracle.convert(np.array([[0.5]], np.float32), oracle.CS_GRAY, oracle.CS_RGB, np.uint8, 3).tolist() == [[[128, 128, 128]]]
    assert oracle.convert(np.array([[0.5]], np.float32), oracle.CS_GRAY, oracle.CS_GRAY, np.uint8, 1).tolist() == [[128]]
    f = oracle.convert(np.array([[128]], np.uint8), oracle.CS_GRAY, oracle.CS_GRAY, np.float32, 1)
    assert abs(float(f[0, 0]) - 128 / 255) < 1e-7


def test_oklab_sanity_against_published_values(oracle):
    """No forward Oklab golden values exist in the reference (round trips only, color.zig:1738-1773);
    these are the published Oklab coordinates of the sRGB primaries, loose enough for the reference's
    4-digit sRGB->XYZ matrix."""
    rgb = np.array([[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [0, 0, 0]]], np.uint8)
    lab = oracle.convert(rgb, oracle.CS_RGB, oracle.CS_OKLAB, np.float32, 3)[0]
    want = [(1.0, 0.0, 0.0), (0.628, 0.225, 0.126), (0.866, -0.234, 0.179), (0.452, -0.032, -0.312), (0, 0, 0)]
    for got, w in zip(lab, want):
        assert np.allclose(got, w, atol=4e-3), (got, w)
    rgba = np.concatenate([rgb, np.full((1, 5, 1), 7, np.uint8)], -1)
    assert np.array_equal(oracle.convert(rgba, oracle.CS_RGBA, oracle.CS_OKLAB, np.float32, 3)[0], lab)


# the restatements of Zig's std maths are pinned by dense sweeps in tests/test_math_pin.py (<= 1 ulp against correctly rounded values)


# ---- blending.zig:198-421 -------------------------------------------------------------------------------------------
def test_blend_modes_known_answers(oracle):
    B = oracle.blend_rgba_u8
    NONE, NORMAL, MULTIPLY, SCREEN, OVERLAY, SOFT, HARD, DODGE, BURN, DARKEN, LIGHTEN, DIFF, EXCL = range(13)
    r = B((100, 100, 100, 255), (200, 200, 200, 128), NORMAL)  # :198-206
    assert all(140 < c < 160 for c in r[:3])
    assert B((255, 255, 255, 255), (128, 128, 128, 255), MULTIPLY)[:3] == (128, 128, 128)  # :208-218
    assert B((0, 0, 0, 255), (128, 128, 128, 255), SCREEN)[:3] == (128, 128, 128)  # :220-230
    assert B((100, 100, 100, 255), (200, 200, 200, 0), NORMAL) == (100, 100, 100, 255)  # :232-243
    r = B((100, 100, 100, 128), (200, 200, 200, 128), NORMAL)  # :245-257
    assert 190 <= r[3] <= 192 and 130 < r[0] < 170
    r = B((0, 0, 0, 0), (200, 150, 100, 180), NORMAL)  # :259-273
    assert r[3] == 180 and abs(r[0] - 200) <= 1 and abs(r[1] - 150) <= 1 and abs(r[2] - 100) <= 1
    m = B((100, 100, 100, 200), (50, 50, 50, 100), MULTIPLY)  # :275-296
    s = B((100, 100, 100, 200), (50, 50, 50, 100), SCREEN)
    assert abs(m[3] - 221) <= 2 and abs(s[3] - 221) <= 2 and m[0] < s[0]
    for mode in (MULTIPLY, SCREEN, EXCL):  # :298-318 hidden base colour
        assert B((25, 75, 125, 0), (200, 150, 100, 180), mode) == (200, 150, 100, 180)
    assert B((100, 100, 100, 255), (200, 200, 200, 255), NONE)[:3] == (200, 200, 200)  # :346-354
    assert B((100, 200, 100, 255), (200, 100, 100, 255), DARKEN)[:3] == (100, 100, 100)  # :391-398
    assert B((100, 200, 100, 255), (200, 100, 100, 255), LIGHTEN)[:3] == (200, 200, 100)  # :400-407
    assert B((200, 100, 50, 255), (50, 200, 200, 255), DIFF)[:3] == (150, 100, 150)  # :409-416


# ---- tests/filters.zig:16-52 (invert), :277-370 (sharpen); tests/integral.zig:11-68 ------------------------------------
def test_invert_sharpen_integral_known_answers(oracle):
    g = np.array([[0, 255], [100, 128]], np.uint8)
    assert oracle.invert(g.copy()).tolist() == [[255, 0], [155, 127]]
    assert oracle.invert(np.array([[[0, 128, 255]]], np.uint8)).tolist() == [[[255, 127, 0]]]
    assert oracle.invert(np.array([[[0, 128, 255, 64]]], np.uint8)).tolist() == [[[255, 127, 0, 64]]]  # alpha kept
    edge = np.where(np.arange(5)[None, :] < 2, 64, 192).astype(np.uint8).repeat(5, 0).reshape(5, 5)
    sh = oracle.sharpen(edge, 1)
    assert sh[2, 0] <= 64 and sh[2, 4] >= 192
    pat = (np.arange(9) + 10).astype(np.uint8).reshape(3, 3)
    assert (oracle.sharpen(pat, 0) == pat).all()
    assert (oracle.sharpen(np.full((4, 4), 100, np.uint8), 1) == 100).all()
    rgba = np.full((3, 3, 4), (64, 64, 64, 255), np.uint8)
    rgba[1, 1] = (192, 192, 192, 255)
    assert (oracle.sharpen(rgba, 1)[1, 1, :3] >= 192).all()
    ones = np.ones((21, 13), np.uint8)
    want = (np.arange(1, 22)[:, None] * np.arange(1, 14)[None, :]).astype(np.float32)
    assert (oracle.integral(ones)[0] == want).all()
    big = np.ones((25, 20), np.uint8)
    assert (oracle.integral(np.ascontiguousarray(big[3:10, 2:8]))[0] == (np.arange(1, 8)[:, None] * np.arange(1, 7)[None, :])).all()
    planes = oracle.integral(np.ones((21, 13, 4), np.uint8))
    assert planes.shape == (4, 21, 13) and all((planes[ch] == want).all() for ch in range(4))
