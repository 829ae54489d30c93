"""GPU parity: resize / letterbox / warp / rotate / extract / crop / insert / flips / convolve / boxBlur /
convert through the C ABI vs the CPU oracle, on seeded inputs. Bit-exact everywhere: u8 paths are integer or
clamp-rounded f32 with the reference's operation order, f32 paths keep that order and never fuse mul+add."""
import math

import numpy as np
import pytest

import zignal_amd as zg
from tests.util import ALL_TYPES, assert_bits_equal, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

I = zg.Interpolation
METHODS = {"nearest": I.nearest, "bilinear": I.bilinear, "bicubic": I.bicubic, "catmull_rom": I.catmull_rom,
           "mitchell": I.mitchell(1 / 3, 1 / 3), "mitchell0": I.mitchell_default, "lanczos": I.lanczos}
BORDERS = (0, 1, 2, 3)


def om(oracle, m):
    return oracle.method(m.kind, m.b, m.c)


def dev(a):
    return zg.Image(torch.from_numpy(np.ascontiguousarray(a)).cuda())


def sync(img):
    torch.cuda.synchronize()
    return img.to_numpy()


# ---- resize ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ALL_TYPES)
@pytest.mark.parametrize("mname", list(METHODS))
def test_resize_parity(oracle, kind, mname):
    m = METHODS[mname]
    for (sr, sc), (dr, dc) in (((37, 53), (19, 71)), ((16, 16), (64, 48)), ((1, 1), (5, 7)), ((9, 4), (9, 4)),
                               ((64, 64), (16, 16)), ((3, 50), (11, 2)), ((70, 301), (33, 260)), ((40, 60), (50, 512))):
        src = synth(oracle, kind, 20, sr, sc)
        want = oracle.resize(src, (dr, dc), om(oracle, m))
        assert_bits_equal(sync(dev(src).resize((dr, dc), m)), want, f"resize {kind} {mname} {sr}x{sc}->{dr}x{dc}")
    # host layer, into a view
    src = synth(oracle, kind, 21, 23, 31)
    base = np.zeros((40, 50) + src.shape[2:], src.dtype)
    zg.Image(src).resize(zg.Image(base).view((3, 2, 3 + 29, 2 + 17)), m)
    want = oracle.resize(src, (17, 29), om(oracle, m))
    assert_bits_equal(base[2:19, 3:32], want, f"resize view {kind} {mname}")
    base[2:19, 3:32] = 0
    assert not base.any()


def test_grey_u8_bilinear_resize_four_pixels_per_lane(oracle):
    """k_resize_bilinear_u8 (geom.hip): what ImagePyramid resizes with. Shrinking (every neighbour inside), growing (the outer ring
    goes through the generic sampler), rows that end inside a lane's four pixels and inside a 256-pixel tile, a destination view
    whose rows do not start on four bytes, and a batch of frames in one launch — all equal to the oracle's generic path."""
    bil = om(oracle, I.bilinear)
    for (sr, sc), (dr, dc) in (((600, 700), (500, 583)), ((301, 517), (150, 259)), ((64, 1030), (64, 257)), ((90, 130), (271, 521)),
                               ((2, 2), (9, 1025)), ((1, 300), (3, 1)), ((1200, 1100), (335, 307))):
        src = oracle.synth_u8(61, (sr, sc))
        assert_bits_equal(sync(dev(src).resize((dr, dc), I.bilinear)), oracle.resize(src, (dr, dc), bil), f"grey {sr}x{sc}->{dr}x{dc}")
    src = oracle.synth_u8(62, (333, 777))
    base = torch.zeros((200, 515), dtype=torch.uint8, device="cuda")
    for c0 in (0, 1, 2, 3):
        out = zg.Image(base[5:5 + 170, c0:c0 + 401])
        dev(src).resize(out, I.bilinear)
        torch.cuda.synchronize()
        assert_bits_equal(out.to_numpy(), oracle.resize(src, (170, 401), bil), f"into a view at column {c0}")
    frames = oracle.synth_u8(63, (5, 120, 300))
    got = zg.Pipeline([zg.Step.resize(77, 260, I.bilinear)]).run(torch.from_numpy(frames).cuda())
    torch.cuda.synchronize()
    for f in range(5):
        assert_bits_equal(got[f].cpu().numpy(), oracle.resize(frames[f], (77, 260), bil), f"frame {f} of a batch")


@pytest.mark.parametrize("kind", ("rgb_u8", "rgba_u8"))
def test_resize_lanczos_with_caller_made_plane_weights(oracle, kind):
    """resizePlaneLanczosU8's weights come from @sin (channel_ops.zig:446-454): a caller may supply its own (R6). The library's own
    weights reproduce the default path and the oracle; perturbed weights are followed exactly (checked against a plain f32
    restatement of the 6 x 6 accumulation, channel_ops.zig:468-489), on both layers."""
    f32 = np.float32
    for (sr, sc), (dr, dc) in (((37, 53), (19, 71)), ((64, 64), (16, 16)), ((9, 14), (30, 5))):
        src = synth(oracle, kind, 33, sr, sc)
        wx, wy = zg.lanczos_plane_weights(sc, dc), zg.lanczos_plane_weights(sr, dr)
        want = oracle.resize(src, (dr, dc), om(oracle, I.lanczos))
        assert_bits_equal(sync(dev(src).resize((dr, dc), I.lanczos, lanczos_weights=(wx, wy))), want, f"{kind} library weights {sr}x{sc}")
        assert_bits_equal(zg.Image(src).resize((dr, dc), I.lanczos, lanczos_weights=(wx, wy)).data, want, f"{kind} library weights, host layer")
    # a caller whose sin differs: weights rounded to 12 bits of significand
    sr, sc, dr, dc = 11, 13, 6, 9
    src = synth(oracle, kind, 34, sr, sc)
    q = lambda w: (np.round(w.astype(np.float64) * 4096) / 4096).astype(f32)
    wx, wy = q(zg.lanczos_plane_weights(sc, dc)), q(zg.lanczos_plane_weights(sr, dr))
    got = sync(dev(src).resize((dr, dc), I.lanczos, lanczos_weights=(wx, wy)))
    mirror = lambda i, n: int(oracle.resolve_index(i, n, oracle.MIRROR))
    rx, ry = f32(sc) / f32(dc), f32(sr) / f32(dr)
    want = np.zeros_like(got)
    for r in range(dr):
        y0 = int(np.floor(f32(f32(f32(r) + f32(0.5)) * ry) - f32(0.5)))
        for c in range(dc):
            x0 = int(np.floor(f32(f32(f32(c) + f32(0.5)) * rx) - f32(0.5)))
            for ch in range(src.shape[2]):
                acc, wsum = f32(0), f32(0)
                for ky in range(6):
                    py = mirror(y0 + ky - 2, sr)
                    for kx in range(6):
                        w = f32(wx[c, kx] * wy[r, ky])
                        acc = f32(acc + f32(f32(src[py, mirror(x0 + kx - 2, sc), ch]) * w))
                        wsum = f32(wsum + w)
                val = f32(acc / wsum) if wsum != 0 else f32(0)
                want[r, c, ch] = int(oracle.lib().zo_clamp_u8_f32(float(val)))
    assert_bits_equal(got, want, f"{kind} caller-made weights")
    with pytest.raises(zg.ZignalError):  # the plane kernels are the Rgb(u8) / Rgba(u8) path only
        dev(synth(oracle, "u8", 1, 8, 8)).resize((4, 4), I.lanczos, lanczos_weights=(zg.lanczos_plane_weights(8, 4), zg.lanczos_plane_weights(8, 4)))


def test_resize_known_answers(oracle):
    # channel_ops.zig:144-190 at ratio 4: floor of the mean of the 2x2 block at (4d+1, 4d+2)
    src = oracle.synth_u8(3, (64, 64, 4))
    out = sync(dev(src).resize((16, 16), I.bilinear))
    s = src.astype(np.int32)
    assert np.array_equal(out, ((s[1::4, 1::4] + s[1::4, 2::4] + s[2::4, 1::4] + s[2::4, 2::4]) // 4).astype(np.uint8))
    # tests/resize.zig:140-161: 1x1 -> 10x10 is constant
    assert np.all(sync(dev(np.array([[128]], np.uint8)).resize((10, 10), I.nearest)) == 128)
    # tests/resize.zig:258-298: scale dims and errors
    img = zg.Image(np.zeros((100, 100), np.uint8))
    assert img.scale(0.5).shape == (50, 50) and img.scale(2.0).shape == (200, 200) and img.scale(1.5, I.nearest).shape == (150, 150)
    with pytest.raises(zg.InvalidArgument):
        img.scale(0)
    with pytest.raises(zg.InvalidArgument):
        zg.Image(np.zeros((2, 2), np.uint8)).scale(0.1)


@pytest.mark.parametrize("kind", ("u8", "rgb_u8", "rgba_f32"))
def test_letterbox_parity(oracle, kind):
    for (sr, sc), (dr, dc), mname in (((4, 8), (6, 6), "bilinear"), ((9, 3), (4, 12), "nearest"), ((4, 6), (8, 12), "bicubic"),
                                      ((2, 32), (64, 64), "bilinear"), ((5, 5), (5, 5), "lanczos")):
        src = synth(oracle, kind, 22, sr, sc)
        m = METHODS[mname]
        want = np.full((dr, dc) + src.shape[2:], 9, src.dtype)
        want_rect = oracle.letterbox(src, want, om(oracle, m))
        got = dev(np.full_like(want, 9))
        _, rect = dev(src).letterbox(got, m)
        assert rect == want_rect
        assert_bits_equal(sync(got), want, f"letterbox {kind} {mname}")
    assert zg.Image(np.zeros((4, 8), np.uint8)).letterbox((6, 6), I.bilinear)[1] == (0, 1, 6, 4)  # tests/resize.zig:12-47


# ---- warp -----------------------------------------------------------------------------------------
H = [[0.92, 0.05, 3.0], [-0.04, 1.05, -2.0], [1e-4, -2e-4, 1.0]]


@pytest.mark.parametrize("kind", ALL_TYPES)
@pytest.mark.parametrize("mname", list(METHODS))
def test_warp_projective_parity(oracle, kind, mname):
    m = METHODS[mname]
    src = synth(oracle, kind, 30, 61, 83)
    want = oracle.warp(src, (70, 90), oracle.PROJECTIVE, np.array(H, np.float32), om(oracle, m))
    got = sync(dev(src).warp(zg.ProjectiveTransform(H), (70, 90), m))
    assert_bits_equal(got, want, f"warp {kind} {mname}")


def test_warp_affine_similarity_and_horizon(oracle):
    src = synth(oracle, "rgba_u8", 31, 40, 50)
    aff = zg.AffineTransform([[0.8, -0.3], [0.25, 1.1]], [4.5, -3.25])
    want = oracle.warp(src, (33, 47), oracle.AFFINE, aff.coefficients(), om(oracle, I.bilinear))
    assert_bits_equal(sync(dev(src).warp(aff, (33, 47), I.bilinear)), want, "affine")
    sim = zg.SimilarityTransform([[0.5, -0.5], [0.5, 0.5]], [10, 2])
    want = oracle.warp(src, (33, 47), oracle.SIMILARITY, sim.coefficients(), om(oracle, I.bicubic))
    assert_bits_equal(sync(dev(src).warp(sim, (33, 47), I.bicubic)), want, "similarity")
    # a homography whose horizon crosses the output: w -> 0 makes coordinates astronomically large; the
    # mirror index of those must still agree (64-bit index path), non-finite ones give zero pixels
    hz = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.02, 0.0, -0.4]]
    for name in ("nearest", "bilinear", "bicubic"):
        want = oracle.warp(src, (30, 60), oracle.PROJECTIVE, np.array(hz, np.float32), om(oracle, METHODS[name]))
        assert_bits_equal(sync(dev(src).warp(zg.ProjectiveTransform(hz), (30, 60), METHODS[name])), want, f"horizon {name}")


# ---- rotate / extract / crop / flips / insert -------------------------------------------------------
@pytest.mark.parametrize("kind", ("u8", "f32", "rgb_u8", "rgba_u8", "rgba_f32"))
def test_rotate_parity(oracle, kind):
    src = synth(oracle, kind, 40, 23, 37)
    for angle in (0.3, -1.1, 2.5, math.pi / 4):
        cs = oracle.cos_sin(angle)
        rows, cols = oracle.rotate_bounds(23, 37, angle)
        assert dev(src).rotate_bounds(angle, cs) == (rows, cols)
        for mname in ("nearest", "bilinear", "bicubic", "lanczos"):
            for border in BORDERS:
                want = oracle.rotate(src, angle, om(oracle, METHODS[mname]), border)
                got = sync(dev(src).rotate(angle, METHODS[mname], border, cos_sin=cs))
                assert_bits_equal(got, want, f"rotate {kind} {angle} {mname} border={border}")
    # exact permutations, also into larger / smaller outputs (centred, zero border)
    for angle, k in ((0.0, 0), (math.pi / 2, 1), (math.pi, 2), (3 * math.pi / 2, 3)):
        assert np.array_equal(sync(dev(src).rotate(angle)), np.rot90(src, k))
        for shape in ((41, 45), (11, 9)):
            want = np.full(shape + src.shape[2:], 3, src.dtype)
            oracle.rotate_into(src, want, angle, om(oracle, I.bilinear), 0)
            got = dev(np.full_like(want, 3))
            dev(src).rotate_into(got, angle)
            assert_bits_equal(sync(got), want, f"rotate_into {kind} {k} {shape}")


@pytest.mark.parametrize("kind", ("u8", "rgba_u8", "rgba_f32"))
def test_extract_crop_parity(oracle, kind):
    src = synth(oracle, kind, 41, 50, 64)
    for rect, angle, shape in (((5.5, 7.25, 40.0, 33.5), 0.4, (21, 30)), ((10, 10, 30, 30), 0.0, (40, 40)),
                               ((-6, -4, 20, 60), -0.8, (17, 9)), ((1, 1, 3, 3), 0.0, (1, 1))):
        cs = oracle.cos_sin(angle)
        for mname in ("nearest", "bilinear", "catmull_rom"):
            for border in BORDERS:
                want = oracle.extract(src, np.empty(shape + src.shape[2:], src.dtype), rect, angle, om(oracle, METHODS[mname]), border)
                got = sync(dev(src).extract(rect, angle, shape, METHODS[mname], border, cos_sin=cs))
                assert_bits_equal(got, want, f"extract {kind} {rect} {mname} {border}")
    # aligned extract == copyRect for every border mode; crop is a bit-exact copy with zero fill
    for border in BORDERS:
        want = oracle.extract(src, np.empty((20, 30) + src.shape[2:], src.dtype), (-5, 40, 25, 60), 0.0, om(oracle, I.nearest), border)
        assert_bits_equal(sync(dev(src).extract((-5, 40, 25, 60), 0.0, (20, 30), I.nearest, border)), want, f"copyRect {border}")
    for rect in ((3, 4, 33, 24), (-7.4, -2.6, 12.5, 9.5), (60, 45, 80, 70), (100, 100, 120, 110)):
        assert_bits_equal(sync(dev(src).crop(rect)), oracle.crop(src, rect), f"crop {rect}")
        assert_bits_equal(zg.Image(src).crop(rect).data, oracle.crop(src, rect), f"crop host {rect}")


def test_extract_known_answers():  # tests/transforms.zig:231-315
    r, c = np.mgrid[0:5, 0:5]
    img = dev((r * 10 + c).astype(np.uint8))
    assert sync(img.extract((1, 1, 3, 3), 0.0, (3, 3), I.nearest, 2)).tolist() == [[11, 12, 13], [21, 22, 23], [31, 32, 33]]
    assert sync(img.extract((1, 1, 3, 3), math.pi / 2, (3, 3), I.nearest, 2)).tolist() == [[13, 23, 33], [12, 22, 32], [11, 21, 31]]
    assert sync(img.extract((1, 1, 3, 3), 0.0, (1, 1), I.nearest, 2)).tolist() == [[22]]
    assert sync(img.extract((1, 1, 3, 3), 0.0, (1, 3), I.nearest, 2)).tolist() == [[21, 22, 23]]
    assert sync(img.extract((1, 1, 3, 3), 0.0, (3, 1), I.nearest, 2)).tolist() == [[12], [22], [32]]


@pytest.mark.parametrize("kind", ALL_TYPES)
def test_flips(oracle, kind):  # tests/transforms.zig:427-456
    for shape in ((2, 3), (3, 2), (17, 31), (1, 1), (64, 65)):
        src = synth(oracle, kind, 42, *shape)
        assert np.array_equal(sync(dev(src).flip_left_right()), src[:, ::-1])
        assert np.array_equal(sync(dev(src).flip_top_bottom()), src[::-1])
        assert np.array_equal(zg.Image(src.copy()).flip_left_right().data, src[:, ::-1])
    base = synth(oracle, kind, 43, 20, 24)
    t = torch.from_numpy(base.copy()).cuda()
    zg.Image(t).view((3, 2, 19, 11)).flip_left_right()
    want = base.copy()
    want[2:11, 3:19] = want[2:11, 3:19][:, ::-1]
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), want)


@pytest.mark.parametrize("kind", ("u8", "rgba_u8", "rgb_f32"))
def test_insert_parity(oracle, kind):
    canvas = synth(oracle, kind, 44, 64, 64)
    source = synth(oracle, kind, 45, 30, 20)
    cases = [((10, 10, 30, 40), 0.0, "nearest"), ((15.5, 12.25, 45, 50), math.pi / 5, "bilinear"),
             ((-8, 30, 30, 70), -0.3, "bicubic"), ((50, 50, 90, 90), 1.0, "bilinear")]
    for rect, angle, mname in cases:
        cs = oracle.cos_sin(angle)
        for blend in (range(13) if kind == "rgba_u8" else (0, 1)):  # every Blending mode (only Rgba(u8) sources composite)
            want = oracle.insert(canvas.copy(), source, rect, angle, om(oracle, METHODS[mname]), blend)
            got = dev(canvas.copy()).insert(dev(source), rect, angle, METHODS[mname], blend, cos_sin=cs)
            assert_bits_equal(sync(got), want, f"insert {kind} {rect} blend={blend}")


def test_insert_blend_known_answer():  # tests/transforms.zig:382-406
    base = np.array([[[0, 0, 255, 255]]], np.uint8)
    overlay = np.array([[[255, 0, 0, 128]]], np.uint8)
    assert np.array_equal(sync(dev(base).insert(dev(overlay), (0, 0, 1, 1), 0.0, I.nearest, 0)), overlay)
    a = 128 / 255
    assert sync(dev(base).insert(dev(overlay), (0, 0, 1, 1), 0.0, I.nearest, 1))[0, 0].tolist() == [round(255 * a), 0, round(255 * (1 - a)), 255]


@pytest.mark.parametrize("skind", ALL_TYPES)
def test_insert_mixed_types(oracle, skind):
    """insert's source is `anytype` (transforms.zig:293): differing types go through assignPixel's convertColor, and Rgba(u8)
    sources with a blend mode composite through Rgba(u8) whatever the destination type (image.zig:67-94)."""
    source = synth(oracle, skind, 47, 23, 31)
    if skind == "rgba_u8":
        source[..., 3] = (source[..., 3].astype(np.int32) * 3 % 256).astype(np.uint8)  # a spread of alphas incl. small ones
        source[0:4, 0:6, 3] = 0
        source[4:8, 0:6, 3] = 255
    for dkind in ALL_TYPES:
        if dkind == skind:
            continue
        canvas = synth(oracle, dkind, 46, 64, 70)
        for rect, angle, mname in (((10, 12, 41, 35), 0.0, "nearest"), ((5.5, 8.25, 60, 50), 0.4, "bilinear"), ((-6, 30, 40, 75), -0.7, "bicubic")):
            cs = oracle.cos_sin(angle)
            for blend in ((0, 1, 2, 7, 12) if skind == "rgba_u8" else (0, 1)):
                want = oracle.insert(canvas.copy(), source, rect, angle, om(oracle, METHODS[mname]), blend)
                got = dev(canvas.copy()).insert(dev(source), rect, angle, METHODS[mname], blend, cos_sin=cs)
                assert_bits_equal(sync(got), want, f"insert {skind} -> {dkind} {rect} blend={blend}")


def test_insert_every_blend_mode_all_alpha_cases(oracle):
    """blendColors (blending.zig:27-157) over the whole alpha / value lattice: a 256 x 256 canvas whose (row, col) sweep base
    and overlay values, inserted 1:1, for every mode; alpha combinations include 0, 255 and both partial."""
    v = np.arange(256, dtype=np.int32)
    for ba, oa in ((255, 255), (255, 128), (200, 100), (0, 180), (90, 0), (1, 254)):
        canvas = np.zeros((256, 256, 4), np.uint8)
        canvas[..., 0] = v[:, None]; canvas[..., 1] = 255 - v[:, None]; canvas[..., 2] = (v[:, None] * 7) % 256; canvas[..., 3] = ba
        over = np.zeros((256, 256, 4), np.uint8)
        over[..., 0] = v[None, :]; over[..., 1] = (v[None, :] * 3) % 256; over[..., 2] = 255 - v[None, :]; over[..., 3] = oa
        for mode in range(13):
            want = oracle.insert(canvas.copy(), over, (0, 0, 256, 256), 0.0, om(oracle, I.nearest), mode)
            got = dev(canvas.copy()).insert(dev(over), (0, 0, 256, 256), 0.0, I.nearest, mode)
            assert_bits_equal(sync(got), want, f"blend mode {mode} base a={ba} overlay a={oa}")


# ---- convolve (2-D) and boxBlur ----------------------------------------------------------------------
@pytest.mark.parametrize("kind", ALL_TYPES)
def test_convolve_parity(oracle, kind):
    rng = np.random.default_rng(50)
    for kh, kw in ((3, 3), (5, 5), (7, 7), (1, 7), (4, 2), (9, 9), (15, 15), (15, 2)):
        k = (rng.random((kh, kw)).astype(np.float32) - np.float32(0.35)) / np.float32(kh * kw * 0.2)
        for border in BORDERS:
            for shape in ((1, 1), (3, 5), (33, 70), (75, 200)):  # the last one has interior 64 x 16 tiles
                if shape == (75, 200) and border != BORDERS[2] and (kh, kw) not in ((3, 3), (15, 15)):
                    continue
                src = synth(oracle, kind, 51, *shape)
                assert_bits_equal(sync(dev(src).convolve(k, border)), oracle.convolve(src, k, border), f"conv2d {kind} {kh}x{kw} {border} {shape}")
    big = np.full((3, 3), 3.0e6, np.float32)  # i64 accumulate path for u8
    src = synth(oracle, kind, 52, 20, 20)
    assert_bits_equal(sync(dev(src).convolve(big, 1)), oracle.convolve(src, big, 1), f"conv2d wide {kind}")


@pytest.mark.parametrize("kind", ALL_TYPES)
def test_sharpen_integral_invert_parity(oracle, kind):
    for (rows, cols) in ((1, 1), (5, 5), (37, 53), (70, 1100)):
        img = synth(oracle, kind, 61, rows, cols)
        for radius in (0, 1, 3, 40):
            assert_bits_equal(sync(dev(img).sharpen(radius)), oracle.sharpen(img, radius), f"sharpen {kind} {rows}x{cols} r={radius}")
        got = dev(img).integral()
        torch.cuda.synchronize()
        assert_bits_equal(got.cpu().numpy(), oracle.integral(img), f"integral {kind} {rows}x{cols}")
        if kind != "f32":
            assert_bits_equal(sync(dev(img.copy()).invert()), oracle.invert(img.copy()), f"invert {kind} {rows}x{cols}")
    host = synth(oracle, kind, 62, 9, 17)
    assert_bits_equal(zg.Image(host).sharpen(2).data, oracle.sharpen(host, 2), "sharpen host layer")
    assert_bits_equal(zg.Image(host).integral(), oracle.integral(host), "integral host layer")
    if kind == "f32":
        with pytest.raises(zg.ZignalError):
            zg.Image(host.copy()).invert()
    else:
        assert_bits_equal(zg.Image(host.copy()).invert().data, oracle.invert(host.copy()), "invert host layer")
    # a view: stride != cols on the source side
    base = synth(oracle, kind, 63, 40, 60)
    v = zg.Image(torch.from_numpy(base).cuda()).view((5, 3, 45, 33))
    assert_bits_equal(sync(v.sharpen(2)), oracle.sharpen(np.ascontiguousarray(base[3:33, 5:45]), 2), "sharpen view")


def test_convolve_known_answers():  # tests/filters.zig:370-398, 571-600, 701-744, 1302-1342
    ident = [[0, 0, 0], [0, 1, 0], [0, 0, 0]]
    img = (np.arange(9, dtype=np.uint8) + 10).reshape(3, 3)
    assert np.array_equal(sync(dev(img).convolve(ident, 0)), img)
    white = np.full((5, 5, 3), 255, np.uint8)
    corner = sync(dev(white).convolve(np.full((3, 3), 1 / 9, np.float32), 0))[0, 0, 0]
    assert corner != 255 and abs(int(corner) - 113) <= 1
    ones = np.ones((10, 20), np.uint8)
    out = sync(dev(ones).convolve([[1, 1, 1], [1, 0, 1], [1, 1, 1]], 0))
    assert np.all(out[1:9, 0] == 5) and np.all(out[1:9, 1] == 8)
    base_src = (np.arange(6)[:, None] * 10 + np.arange(8)[None, :]).astype(np.uint8)
    td = torch.full((6, 8), 0xAA, dtype=torch.uint8).cuda()
    dev(base_src).view((2, 1, 6, 5)).convolve(ident, 0, out=zg.Image(td).view((2, 1, 6, 5)))
    got = sync(zg.Image(td))
    assert np.array_equal(got[1:5, 2:6], base_src[1:5, 2:6])
    got[1:5, 2:6] = 0xAA
    assert np.all(got == 0xAA)


@pytest.mark.parametrize("kind", ALL_TYPES)
def test_box_blur_parity(oracle, kind):
    for shape, radius in (((1, 1), 1), ((7, 5), 1), ((33, 70), 2), ((130, 67), 5), ((20, 20), 40), ((9, 9), 0)):
        src = synth(oracle, kind, 53, *shape)
        assert_bits_equal(sync(dev(src).box_blur(radius)), oracle.box_blur(src, radius), f"boxBlur {kind} {shape} r={radius}")
    src = synth(oracle, kind, 54, 40, 41)
    t = dev(src)
    t.box_blur(3, out=t)  # in place, as examples/src/face_alignment.zig:95 does
    assert_bits_equal(sync(t), oracle.box_blur(src, 3), f"boxBlur in place {kind}")


def test_box_blur_config1(oracle):
    """BASELINE.json configs[0]: 3x3 box blur on 256x256 u8 (radius 1); SAT stays below 2^24 so it is exact."""
    src = oracle.synth_u8(1, (256, 256))
    want = oracle.box_blur(src, 1)
    assert_bits_equal(sync(dev(src).box_blur(1)), want, "config1")
    s = np.pad(src.astype(np.int64), 1)
    sums = sum(s[1 + dy:257 + dy, 1 + dx:257 + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1))
    assert np.array_equal(want[1:-1, 1:-1], np.floor(sums[1:-1, 1:-1] / 9 + 0.5).astype(np.uint8))
    # images of rows * cols * 255 < 2^24 take no integral image at all (k_box_direct: the window's integer sum, which is what the reference's then-exact
    # f32 SAT yields): the largest such image all white, every radius the direct kernel takes, every u8 type, sharpen, a view as destination; one pixel
    # more and the SAT kernels run — both against the oracle
    white = np.full((256, 257), 255, np.uint8)
    assert white.size * 255 < 1 << 24 <= 257 * 257 * 255
    for radius in (1, 2, 3, 7):
        assert_bits_equal(sync(dev(white).box_blur(radius)), oracle.box_blur(white, radius), f"direct, all white r={radius}")
    big = np.full((257, 257), 255, np.uint8)
    assert_bits_equal(sync(dev(big).box_blur(2)), oracle.box_blur(big, 2), "one pixel past the direct path")
    for kind in ("u8", "rgb_u8", "rgba_u8"):
        img = synth(oracle, kind, 57, 120, 136)
        for radius in (1, 4, 7, 8):  # 8: past the direct kernel's windows
            assert_bits_equal(sync(dev(img).box_blur(radius)), oracle.box_blur(img, radius), f"direct {kind} r={radius}")
        assert_bits_equal(sync(dev(img).sharpen(2)), oracle.sharpen(img, 2), f"direct sharpen {kind}")
    img = oracle.synth_u8(58, (90, 100, 4))
    td = torch.full((100, 120, 4), 0x5A, dtype=torch.uint8, device="cuda")
    dev(img).box_blur(2, out=zg.Image(td).view((10, 5, 110, 95)))
    got = sync(zg.Image(td))
    assert_bits_equal(got[5:95, 10:110], oracle.box_blur(img, 2), "direct into a view")
    got[5:95, 10:110] = 0x5A
    assert np.all(got == 0x5A)
    f = oracle.synth_f32(55, (300, 300)) * np.float32(1000)  # f32 SAT is inexact here: order must match
    assert_bits_equal(sync(dev(f).box_blur(4)), oracle.box_blur(f, 4), "f32 SAT order")


# ---- colour ---------------------------------------------------------------------------------------------
CONVERSIONS = [  # (src kind, src space, dst space, dst dtype)
    ("rgba_u8", zg.CS_RGBA, zg.CS_OKLAB, np.float32), ("rgb_u8", zg.CS_RGB, zg.CS_OKLAB, np.float32),
    ("rgba_f32", zg.CS_RGBA, zg.CS_OKLAB, np.float32), ("rgb_f32", zg.CS_RGB, zg.CS_OKLAB, np.float32),
    ("rgb_u8", zg.CS_RGB, zg.CS_XYZ, np.float32), ("rgb_f32", zg.CS_RGB, zg.CS_XYZ, np.float32),
    ("rgba_u8", zg.CS_RGBA, zg.CS_GRAY, np.uint8), ("rgb_u8", zg.CS_RGB, zg.CS_GRAY, np.float32),
    ("rgb_f32", zg.CS_RGB, zg.CS_GRAY, np.uint8), ("rgba_f32", zg.CS_RGBA, zg.CS_GRAY, np.float32),
    ("rgb_u8", zg.CS_RGB, zg.CS_RGBA, np.uint8), ("rgba_u8", zg.CS_RGBA, zg.CS_RGB, np.uint8),
    ("rgb_u8", zg.CS_RGB, zg.CS_RGB, np.float32), ("rgba_f32", zg.CS_RGBA, zg.CS_RGBA, np.uint8),
    ("rgb_f32", zg.CS_RGB, zg.CS_RGBA, np.float32), ("u8", zg.CS_GRAY, zg.CS_GRAY, np.float32),
    ("f32", zg.CS_GRAY, zg.CS_GRAY, np.uint8), ("u8", zg.CS_GRAY, zg.CS_RGB, np.uint8), ("f32", zg.CS_GRAY, zg.CS_RGBA, np.uint8),
    ("u8", zg.CS_GRAY, zg.CS_RGBA, np.float32), ("rgb_u8", zg.CS_RGB, zg.CS_YCBCR, np.uint8), ("rgba_u8", zg.CS_RGBA, zg.CS_YCBCR, np.uint8),
]


@pytest.mark.parametrize("kind,src_space,dst_space,dtype", CONVERSIONS)
def test_convert_parity(oracle, kind, src_space, dst_space, dtype):
    src = synth(oracle, kind, 60, 67, 129)
    if src.dtype == np.float32:
        src = src * np.float32(1.2) - np.float32(0.1)  # exercise the clamps
    ch = {zg.CS_GRAY: 1, zg.CS_RGBA: 4}.get(dst_space, 3)
    want = oracle.convert(src, src_space, dst_space, dtype, ch)
    got = sync(dev(src).convert(dst_space, dtype, src_space=src_space))
    assert_bits_equal(got, want, f"convert {kind} {src_space}->{dst_space}")
    host = zg.Image(src).convert(dst_space, dtype, src_space=src_space).data
    assert_bits_equal(host, want, "convert host layer")


def test_convert_known_answers_and_caller_lut(oracle):  # color.zig:1556-1583
    rgb = np.array([[[128, 128, 128], [255, 0, 0]]], np.uint8)
    assert sync(dev(rgb).convert(zg.CS_GRAY, np.uint8)).tolist() == [[128, 54]]
    assert sync(dev(np.array([[128]], np.uint8)).convert(zg.CS_RGB, np.uint8)).tolist() == [[[128, 128, 128]]]
    assert sync(dev(np.array([[0.5]], np.float32)).convert(zg.CS_RGB, np.uint8)).tolist() == [[[128, 128, 128]]]
    assert sync(dev(np.array([[0.5]], np.float32)).convert(zg.CS_GRAY, np.uint8)).tolist() == [[128]]
    # all 256 u8 levels through the library's own table == the oracle's table (both restate Zig's pow)
    ramp = np.stack([np.arange(256, dtype=np.uint8)] * 3, -1)[None]
    assert_bits_equal(sync(dev(ramp).convert(zg.CS_XYZ, np.float32)), oracle.convert(ramp, zg.CS_RGB, zg.CS_XYZ, np.float32, 3), "lut")
    # a caller-supplied table (a Zig host would pass std.math.pow's values) is honoured verbatim
    lut = np.linspace(0, 1, 256, dtype=np.float32) ** 2
    want = oracle.convert(ramp, zg.CS_RGB, zg.CS_OKLAB, np.float32, 3, srgb_lut=lut)
    assert_bits_equal(sync(dev(ramp).convert(zg.CS_OKLAB, np.float32, srgb_lut=lut)), want, "caller lut")


# ---- BASELINE.json configs[2..4] at full size --------------------------------------------------------------
def test_config3_resize_then_oklab(oracle):
    src = oracle.synth_u8(3, (4096, 4096, 4))
    small = dev(src).resize((1024, 1024), I.bilinear)
    lab = small.convert(zg.CS_OKLAB, np.float32)
    got_small, got_lab = sync(small), sync(lab)
    want_small = oracle.resize(src, (1024, 1024), om(oracle, I.bilinear))
    assert_bits_equal(got_small, want_small, "config3 resize")
    want_lab = oracle.convert(want_small, zg.CS_RGBA, zg.CS_OKLAB, np.float32, 3)
    assert_bits_equal(got_lab, want_lab, "config3 oklab")
    # size-independent property of ratio 4: floor of the 2x2 mean
    s = src.astype(np.int32)
    assert np.array_equal(got_small, ((s[1::4, 1::4] + s[1::4, 2::4] + s[2::4, 1::4] + s[2::4, 2::4]) // 4).astype(np.uint8))


def test_resize_convert_equals_the_two_steps(oracle):
    """zg_resize_convert: [resize, convert] in one call — fused for Rgba(u8) bilinear -> Oklab / Xyz, composed otherwise — equals the
    oracle's resize followed by its convert, bit for bit, on both layers; BASELINE configs[2] at full size below."""
    bil = oracle.method(oracle.BILINEAR)
    for (sr, sc), (dr, dc) in (((64, 64), (16, 16)), ((37, 53), (19, 71)), ((5, 2), (9, 30)), ((300, 200), (77, 130))):
        src = oracle.synth_u8(40, (sr, sc, 4))
        small = oracle.resize(src, (dr, dc), bil)
        for space, ospace in ((zg.CS_OKLAB, oracle.CS_OKLAB), (zg.CS_XYZ, oracle.CS_XYZ)):
            want = oracle.convert(small, oracle.CS_RGBA, ospace, np.float32, 3)
            assert_bits_equal(sync(dev(src).resize_convert((dr, dc), space)), want, f"fused {sr}x{sc}->{dr}x{dc} space {space}")
            assert_bits_equal(zg.Image(src).resize_convert((dr, dc), space).data, want, "fused, host layer")
        # a caller-made sRGB table is honoured by the fused kernel too
        lut = (np.arange(256, dtype=np.float32) / 255) ** 2
        want = oracle.convert(small, oracle.CS_RGBA, oracle.CS_OKLAB, np.float32, 3, srgb_lut=lut)
        assert_bits_equal(sync(dev(src).resize_convert((dr, dc), zg.CS_OKLAB, srgb_lut=lut)), want, "fused, caller lut")
    # combinations without a fused kernel run as the two steps
    src = oracle.synth_u8(41, (40, 60, 3))
    want = oracle.convert(oracle.resize(src, (25, 33), oracle.method(oracle.BICUBIC)), oracle.CS_RGB, oracle.CS_OKLAB, np.float32, 3)
    assert_bits_equal(sync(dev(src).resize_convert((25, 33), zg.CS_OKLAB, method=I.bicubic)), want, "composed rgb bicubic")
    srcf = oracle.synth_f32(42, (40, 60, 4))
    want = oracle.convert(oracle.resize(srcf, (20, 30), bil), oracle.CS_RGBA, oracle.CS_GRAY, np.uint8, 1)
    assert_bits_equal(sync(dev(srcf).resize_convert((20, 30), zg.CS_GRAY, dtype=np.uint8)), want, "composed f32 -> gray u8")
    # configs[2] at full size
    src = oracle.synth_u8(3, (4096, 4096, 4))
    want = oracle.convert(oracle.resize(src, (1024, 1024), bil), oracle.CS_RGBA, oracle.CS_OKLAB, np.float32, 3)
    assert_bits_equal(sync(dev(src).resize_convert((1024, 1024), zg.CS_OKLAB)), want, "configs[2] fused at full size")


def test_config3_rgba_f32_variant_at_full_size(oracle):
    """SURVEY 8(d) lists config 3 also for Image(Rgba(f32)): the generic resizer (interpolation.zig:194-214, lerpFloat) 4096^2 ->
    1024^2, then Rgba(f32) -> Oklab(f32) through the device's own gammaToLinear (pow) and cbrt."""
    src = oracle.synth_f32(3, (4096, 4096, 4))
    small = oracle.resize(src, (1024, 1024), oracle.method(oracle.BILINEAR))
    got = dev(src).resize((1024, 1024), I.bilinear)
    assert_bits_equal(sync(got), small, "configs[2] Rgba(f32) resize")
    assert_bits_equal(sync(got.convert(zg.CS_OKLAB, np.float32)), oracle.convert(small, oracle.CS_RGBA, oracle.CS_OKLAB, np.float32, 3), "configs[2] Rgba(f32) -> Oklab")
    assert_bits_equal(sync(dev(src).resize_convert((1024, 1024), zg.CS_OKLAB)), oracle.convert(small, oracle.CS_RGBA, oracle.CS_OKLAB, np.float32, 3), "configs[2] Rgba(f32), one call")


@pytest.mark.parametrize("kind", ("rgba_u8", "rgba_f32"))
def test_config4_projective_bicubic(oracle, kind):
    src_pts = [(0, 0), (4095, 0), (0, 4095), (4095, 4095)]
    dst_pts = [(200, 120), (3900, 60), (90, 3980), (4000, 4050)]
    # backward map (output -> source), solved in f64 then cast to f32 as qrcode/detector.zig:667-677 does
    hmat = oracle.homography_from_4pts(src_pts, dst_pts)
    rows = 4096  # all of BASELINE configs[3] for both pixel types (the oracle needs a few seconds for the f32 one)
    src = synth(oracle, kind, 4, 4096, 4096)
    got = sync(dev(src).warp(zg.ProjectiveTransform(hmat), (rows, 4096), I.bicubic))
    want = oracle.warp(src, (rows, 4096), oracle.PROJECTIVE, hmat, om(oracle, I.bicubic))
    assert_bits_equal(got, want, f"config4 {kind}")


def test_config5_batch_pipeline(oracle):
    n, rows, cols = 6, 1080, 1920
    frames = oracle.synth_u8(5, (n, rows, cols, 4))
    t_in = torch.from_numpy(frames).cuda()
    t_out = torch.empty((n, 540, 960, 4), dtype=torch.uint8, device="cuda")
    m = I.bilinear._c()
    import ctypes as C
    rc = zg.lib().zg_batch_blur_resize(C.c_void_p(t_in.data_ptr()), n, rows, cols, 3, C.c_float(0.6), C.c_void_p(t_out.data_ptr()),
                                       540, 960, C.byref(m), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, zg.lib().zg_last_error()
    torch.cuda.synchronize()
    got = t_out.cpu().numpy()
    for i in range(n):
        want = oracle.resize(oracle.gaussian_blur(frames[i], 0.6), (540, 960), om(oracle, I.bilinear))
        assert_bits_equal(got[i], want, f"config5 frame {i}")


def test_config5_at_the_per_gpu_shard_size(oracle):
    """BASELINE configs[4]: 1024 frames over 8 GPUs = 128 frames of 1080p per GPU in ONE zg_batch_blur_resize call (1.06 GB in,
    265 MB out); every 16th frame and the last one against the oracle, and no frame may equal its neighbour's result (each slot
    of the batch was really written from its own source)."""
    import ctypes as C
    n, rows, cols = 128, 1080, 1920
    gen = torch.Generator(device="cuda")
    gen.manual_seed(55)
    t_in = torch.randint(0, 256, (n, rows, cols, 4), dtype=torch.uint8, device="cuda", generator=gen)
    t_out = torch.zeros((n, 540, 960, 4), dtype=torch.uint8, device="cuda")
    m = I.bilinear._c()
    rc = zg.lib().zg_batch_blur_resize(C.c_void_p(t_in.data_ptr()), n, rows, cols, 3, C.c_float(0.6), C.c_void_p(t_out.data_ptr()),
                                       540, 960, C.byref(m), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, zg.lib().zg_last_error()
    torch.cuda.synchronize()
    for i in list(range(0, n, 16)) + [n - 1]:
        frame = t_in[i].cpu().numpy()
        want = oracle.resize(oracle.gaussian_blur(frame, 0.6), (540, 960), om(oracle, I.bilinear))
        assert_bits_equal(t_out[i].cpu().numpy(), want, f"config5 128-frame batch, frame {i}")
    sums = t_out.view(n, -1).to(torch.int64).sum(dim=1)
    assert int((sums > 0).sum()) == n and len(set(sums.tolist())) == n


@pytest.mark.parametrize("case", ["half_sigma1", "third_bicubic", "odd_cols", "f32"])
def test_batch_pipeline_paths(oracle, case):
    """Every dispatch path of zg_batch_blur_resize equals gaussianBlur followed by resize, frame by frame."""
    import ctypes as C
    n = 3
    if case == "half_sigma1":      # fused blur + 2:1 bilinear, 7 taps, zero-sum alpha handled like any channel
        rows, cols, orows, ocols, sigma, m, kind = 66, 320, 33, 160, 1.0, I.bilinear, "rgba_u8"
    elif case == "third_bicubic":  # batched blur, then per-frame plane resize
        rows, cols, orows, ocols, sigma, m, kind = 60, 256, 20, 100, 0.6, I.bicubic, "rgba_u8"
    elif case == "odd_cols":       # cols % 4 != 0: general per-frame path
        rows, cols, orows, ocols, sigma, m, kind = 31, 131, 16, 65, 0.6, I.bilinear, "rgba_u8"
    else:
        rows, cols, orows, ocols, sigma, m, kind = 40, 72, 20, 36, 0.6, I.bilinear, "rgba_f32"
    frames = np.stack([synth(oracle, kind, 70 + i, rows, cols) for i in range(n)])
    t_in = torch.from_numpy(frames).cuda()
    t_out = torch.zeros((n, orows, ocols, 4), dtype=t_in.dtype, device="cuda")
    mm = m._c()
    pixel = 3 if kind == "rgba_u8" else 5
    rc = zg.lib().zg_batch_blur_resize(C.c_void_p(t_in.data_ptr()), n, rows, cols, pixel, C.c_float(sigma), C.c_void_p(t_out.data_ptr()),
                                       orows, ocols, C.byref(mm), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, zg.lib().zg_last_error()
    torch.cuda.synchronize()
    got = t_out.cpu().numpy()
    for i in range(n):
        want = oracle.resize(oracle.gaussian_blur(frames[i], sigma), (orows, ocols), om(oracle, m))
        assert_bits_equal(got[i], want, f"batch {case} frame {i}")
