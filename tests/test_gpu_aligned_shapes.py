"""GPU parity on shapes that take the wide paths: widths divisible by 4 / 64 / 256, whole 64 x 64 tiles, 16-byte aligned
planes. The other parity files use odd sizes on purpose (ragged edges, scalar fallbacks); these are the sizes production
frames have, where the four-pixels-per-lane, float4 and whole-tile kernels run. Bit-exact against the oracle."""
import numpy as np
import pytest

import zignal_amd as zg
from tests.util import assert_bits_equal, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return zg.Image(torch.from_numpy(np.ascontiguousarray(a)).cuda())


def sync(img):
    torch.cuda.synchronize()
    return img.to_numpy()


def structured(rows, cols, seed, kind):
    """Smooth blobs + a step + a little texture: edges that run across tiles, weak and strong responses both."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:rows, 0:cols].astype(np.float32)
    f = np.zeros((rows, cols), np.float32)
    for _ in range(14):
        cy, cx, s = rng.uniform(0, rows), rng.uniform(0, cols), rng.uniform(5, 50)
        f += rng.uniform(-1, 1) * np.exp(-((y - cy) ** 2 + (x - cx) ** 2) / (2 * s * s))
    f += (x > cols * 0.55) * 0.35 + (y > rows * 0.4) * 0.2
    f += rng.uniform(-0.02, 0.02, f.shape).astype(np.float32)
    f = (f - f.min()) / (f.max() - f.min() + 1e-9)
    if kind == "u8":
        return (f * 255).astype(np.uint8)
    if kind == "f32":
        return f.astype(np.float32)
    ch = 3 if kind.startswith("rgb_") else 4
    w = np.array([1.0, 0.75, 0.5, 1.0], np.float32)[:ch]
    if kind.endswith("u8"):
        return (f[..., None] * 255 * w).astype(np.uint8)
    return (f[..., None] * w).astype(np.float32)


ALIGNED = ((64, 64), (128, 256), (192, 320), (200, 260), (65, 512))


@pytest.mark.parametrize("kind", ("u8", "rgb_u8", "rgba_u8", "f32", "rgba_f32"))
def test_box_sharpen_integral_aligned(oracle, kind):
    for (rows, cols) in ALIGNED:
        src = synth(oracle, kind, 71, rows, cols)
        for radius in (1, 2, 9):
            assert_bits_equal(sync(dev(src).box_blur(radius)), oracle.box_blur(src, radius), f"boxBlur {kind} {rows}x{cols} r={radius}")
        assert_bits_equal(sync(dev(src).sharpen(2)), oracle.sharpen(src, 2), f"sharpen {kind} {rows}x{cols}")
        got = dev(src).integral()
        torch.cuda.synchronize()
        assert_bits_equal(got.cpu().numpy(), oracle.integral(src), f"integral {kind} {rows}x{cols}")


def test_box_blur_large_sat_values(oracle):
    """Sums far above 2^24: the f32 SAT rounds at every step, the order of the column recurrence is what is being checked."""
    src = np.full((1024, 1280, 4), 255, np.uint8)
    src[::7, ::5] = 3
    assert_bits_equal(sync(dev(src).box_blur(2)), oracle.box_blur(src, 2), "boxBlur bright 1024x1280 Rgba(u8)")
    g = np.full((1536, 1024), 250, np.uint8)
    g[::3, ::11] = 0
    assert_bits_equal(sync(dev(g).box_blur(3)), oracle.box_blur(g, 3), "boxBlur bright 1536x1024 u8")


@pytest.mark.parametrize("kind", ("u8", "rgba_u8", "f32"))
def test_sobel_canny_aligned(oracle, kind):
    for (rows, cols) in ALIGNED:
        img = structured(rows, cols, rows + 3 * cols, kind)
        assert_bits_equal(sync(dev(img).sobel()), oracle.sobel(img), f"sobel {kind} {rows}x{cols}")
        for (sigma, lo, hi) in ((1.4, 8, 24), (0.0, 4, 10)):
            want = oracle.canny(img, sigma, lo, hi)
            assert_bits_equal(sync(dev(img).canny(sigma, lo, hi)), want, f"canny {kind} {rows}x{cols} sigma={sigma}")
        assert want.any()


@pytest.mark.parametrize("kind", ("u8", "rgba_u8", "f32"))
def test_shen_castan_aligned(oracle, kind):
    for (rows, cols) in ALIGNED:
        img = structured(rows, cols, 5 * rows + cols, kind)
        for opts in (dict(), dict(use_nms=True, high_ratio=0.9), dict(hysteresis=False, high_ratio=0.8, window_size=3),
                     dict(smooth=0.7, window_size=11, high_ratio=0.95, low_rel=0.2)):
            want = oracle.shen_castan(img, **opts)
            assert_bits_equal(sync(dev(img).shen_castan(**opts)), want, f"shenCastan {kind} {rows}x{cols} {opts}")


def test_shen_castan_tiny_aligned(oracle):
    """cols % 4 == 0 below 3 rows: the bounded four-neighbour candidate rule of the four-pixel kernel."""
    for (rows, cols) in ((1, 4), (2, 8), (2, 64), (1, 260)):
        img = structured(rows, cols, rows + cols, "u8")
        for opts in (dict(high_ratio=0.5), dict(high_ratio=0.5, use_nms=True)):
            assert_bits_equal(sync(dev(img).shen_castan(**opts)), oracle.shen_castan(img, **opts), f"shenCastan tiny {rows}x{cols} {opts}")


def test_hysteresis_noise_components_across_tiles(oracle):
    """Dense candidates: components that percolate through many 64 x 64 tiles, strong pixels far from most of their weak ones."""
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (320, 448), dtype=np.uint8)
    for (lo, hi) in ((20, 400), (60, 250)):
        want = oracle.canny(img, 0.0, lo, hi)
        assert_bits_equal(sync(dev(img).canny(0.0, lo, hi)), want, f"canny noise lo={lo} hi={hi}")
    want = oracle.shen_castan(img, high_ratio=0.97, low_rel=0.1)
    assert_bits_equal(sync(dev(img).shen_castan(high_ratio=0.97, low_rel=0.1)), want, "shenCastan noise")


@pytest.mark.parametrize("kind", ("rgba_u8", "rgba_f32", "u8", "f32"))
def test_bicubic_family_interior_fast_path(oracle, kind):
    """Warp / resize / rotate with every tap inside the image: the buffer-load gather and the shared-reciprocal quotients."""
    I = zg.Interpolation
    src = synth(oracle, kind, 91, 96, 128)
    for m in (I.bicubic, I.catmull_rom, I.mitchell_default, I.lanczos):
        om = oracle.method(m.kind, m.b, m.c)
        want = oracle.resize(src, (77, 150), om)
        assert_bits_equal(sync(dev(src).resize((77, 150), m)), want, f"resize {kind} kind={m.kind}")


@pytest.mark.parametrize("kind", ("f32", "rgb_f32", "rgba_f32", "u8", "rgb_u8", "rgba_u8"))
def test_separable_equal_odd_taps_on_narrow_images(oracle, kind):
    """Equal odd tap counts up to 13 on images narrower than a tile: the fused kernels' widest instantiations (a fuzz run found the
    13-tap Rgba(f32) one returning a wrong third channel: register spills miscompiled; it now takes the two-pass kernels)."""
    rng = np.random.default_rng(17)
    for nk in (9, 11, 13):
        for cols in (1, 2, 5, 16, 40, 63):
            img = synth(oracle, kind, 90 + nk + cols, 37, cols)
            kx = rng.random(nk).astype(np.float32) - np.float32(0.3)
            ky = rng.random(nk).astype(np.float32) - np.float32(0.3)
            for border in (0, 1, 2, 3):
                assert_bits_equal(sync(dev(img).convolve_separable(kx, ky, border)), oracle.conv_separable(img, kx, ky, border),
                                  f"sep {kind} 37x{cols} n={nk} b={border}")
