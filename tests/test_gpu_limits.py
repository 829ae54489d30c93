"""Sizes the reference accepts without limit and the library used to refuse: Gaussian kernels of any
radius (image.zig:973: radius = ceil(3 sigma)), 2-D kernels of any comptime size (convolution.zig:76), images taller than
HIP's 65 535-workgroup grid dimension (rows: u32, image.zig:97-103). All against the oracle, bit for bit."""
import numpy as np
import pytest

import zignal_amd as zg
from tests.util import assert_bits_equal, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

I = zg.Interpolation


def dev(a):
    return zg.Image(torch.from_numpy(np.ascontiguousarray(a)).cuda())


def sync(img):
    torch.cuda.synchronize()
    return img.to_numpy()


def test_gaussian_blur_sigma_100_on_a_70000_row_strip(oracle):
    """601 taps (past the 255 that travel as a kernel argument) on a strip taller than gridDim.y allows."""
    src = synth(oracle, "u8", 70, 70000, 24)
    assert_bits_equal(sync(dev(src).gaussian_blur(100.0)), oracle.gaussian_blur(src, 100.0), "u8 70000 x 24, sigma 100")
    src = synth(oracle, "f32", 71, 70000, 6)
    assert_bits_equal(sync(dev(src).gaussian_blur(100.0)), oracle.gaussian_blur(src, 100.0), "f32 70000 x 6, sigma 100")


@pytest.mark.parametrize("kind", ("u8", "rgba_u8", "f32", "rgb_f32"))
def test_long_separable_kernels(oracle, kind):
    rng = np.random.default_rng(5)
    src = synth(oracle, kind, 72, 90, 70)
    for n in (257, 301, 1001):  # longer than the image: every tap resolves through the border rule
        k = rng.normal(0, 1, n).astype(np.float32)
        k /= np.abs(k).sum()
        for border in (0, 1, 2, 3):
            want = oracle.conv_separable(src, k, k[::-1].copy(), border)
            assert_bits_equal(sync(dev(src).convolve_separable(k, k[::-1].copy(), border)), want, f"{kind} {n} taps border {border}")
    assert_bits_equal(sync(dev(src).gaussian_blur(90.0)), oracle.gaussian_blur(src, 90.0), f"{kind} sigma 90")


@pytest.mark.parametrize("kind", ("u8", "rgb_u8", "rgba_u8", "f32", "rgba_f32"))
def test_large_2d_kernels(oracle, kind):
    rng = np.random.default_rng(6)
    src = synth(oracle, kind, 73, 60, 77)
    for kh, kw in ((17, 17), (21, 9), (3, 31), (33, 1)):
        k = rng.normal(0, 1, (kh, kw)).astype(np.float32)
        k /= np.abs(k).sum()
        for border in (0, 1, 2, 3):
            assert_bits_equal(sync(dev(src).convolve(k, border)), oracle.convolve(src, k, border), f"{kind} {kh}x{kw} border {border}")


def test_ops_on_an_image_taller_than_the_grid_limit(oracle):
    rows, cols = 66000, 40
    src = synth(oracle, "rgba_u8", 74, rows, cols)
    d = dev(src)
    assert_bits_equal(sync(d.gaussian_blur(0.6)), oracle.gaussian_blur(src, 0.6), "blur")
    assert_bits_equal(sync(d.gaussian_blur(2.5)), oracle.gaussian_blur(src, 2.5), "blur 17 taps")
    assert_bits_equal(sync(d.box_blur(2)), oracle.box_blur(src, 2), "boxBlur")
    assert_bits_equal(sync(d.convert(zg.CS_OKLAB, np.float32)), oracle.convert(src, oracle.CS_RGBA, oracle.CS_OKLAB, np.float32, 3), "convert")
    assert_bits_equal(sync(d.convert(zg.CS_LAB, np.float32)), oracle.convert(src, oracle.CS_RGBA, oracle.CS_LAB, np.float32, 3), "convert Lab")
    assert_bits_equal(sync(d.resize((rows // 3, 64), I.bilinear)), oracle.resize(src, (rows // 3, 64), oracle.method(oracle.BILINEAR)), "resize")
    assert_bits_equal(sync(d.crop((3, 100, 35, 65900))), src[100:65900, 3:35], "crop")
    k = np.array([[0, 1, 0], [1, 4, 1], [0, 1, 0]], np.float32) / 8
    assert_bits_equal(sync(d.convolve(k, 1)), oracle.convolve(src, k, 1), "convolve 3x3")
    flipped = dev(src.copy())
    flipped.flip_top_bottom()
    assert_bits_equal(sync(flipped), src[::-1], "flipTopBottom")
    flipped.flip_left_right()
    assert_bits_equal(sync(flipped), src[::-1, ::-1], "flipLeftRight")
    inv = dev(src.copy())
    inv.invert()
    want = 255 - src
    want[..., 3] = src[..., 3]
    assert_bits_equal(sync(inv), want, "invert")
    f = synth(oracle, "rgba_f32", 75, rows, 12)
    assert_bits_equal(sync(dev(f).gaussian_blur(0.6)), oracle.gaussian_blur(f, 0.6), "f32 blur")
    assert_bits_equal(sync(dev(f).gaussian_blur(3.0)), oracle.gaussian_blur(f, 3.0), "f32 blur 19 taps")


def test_canny_and_axis_aligned_motion_blur_past_255_taps(oracle):
    """The last two places that refused kernels longer than 255 taps: canny's own blur at sigma 45 (271 taps) and
    motionBlur(.linear) along an axis at distance 300. The reference has neither limit (edges.zig:212-260, motion_blur.zig:65-130)."""
    img = oracle.synth_u8(61, (96, 700, 4))
    got = zg.Image(torch.from_numpy(img).cuda()).canny(45.0, 5, 15)
    torch.cuda.synchronize()
    assert np.array_equal(got.to_numpy(), oracle.canny(img, 45.0, 5, 15))
    for angle, shape in ((0.0, (40, 900, 4)), (np.float32(np.pi / 2), (900, 40, 4))):
        img = oracle.synth_u8(62, shape)
        got = zg.Image(torch.from_numpy(img).cuda()).motion_blur_linear(float(angle), 300)
        torch.cuda.synchronize()
        assert np.array_equal(got.to_numpy(), oracle.motion_blur_linear(img, float(angle), 300)), angle
