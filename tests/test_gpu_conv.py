"""GPU parity: Image.convolveSeparable / gaussianBlur through the C ABI vs the CPU oracle.

Bit-exact for every pixel type (integer paths by construction; f32 because the kernels keep the
reference's operation order and never contract mul+add)."""
import numpy as np
import pytest

import zignal_amd as zg
from tests.util import ALL_TYPES, assert_bits_equal, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

BORDERS = (zg.BorderMode.zero, zg.BorderMode.replicate, zg.BorderMode.mirror, zg.BorderMode.wrap)


def dev(a):
    return zg.Image(torch.from_numpy(np.ascontiguousarray(a)).cuda())


def run_dev(a, kx, ky, border):
    out = dev(a).convolve_separable(kx, ky, border)
    torch.cuda.synchronize()
    return out.to_numpy()


# ---- reference known answers, through both layers (tests/filters.zig) -----------------------------
def test_impulse_f32():  # filters.zig:469-491
    img = np.zeros((7, 7), np.float32)
    img[3, 3] = 1.0
    g = [0.25, 0.5, 0.25]
    for out in (zg.Image(img).convolve_separable(g, g, zg.BorderMode.zero).data, run_dev(img, g, g, zg.BorderMode.zero)):
        assert out[3, 3] == np.float32(0.25) and out[3, 2] == np.float32(0.125)


def test_separable_identity_on_strided_f32_view():  # filters.zig:602-632
    base = (np.arange(5)[:, None] * 10 + np.arange(5)[None, :]).astype(np.float32)
    view = zg.Image(base).view((1, 1, 4, 4))
    out = view.convolve_separable([1.0], [1.0], zg.BorderMode.zero, out=zg.Image(np.empty((3, 3), np.float32)))
    assert np.array_equal(out.data, base[1:4, 1:4])


def test_separable_into_view_leaves_outside_untouched():  # filters.zig:746-783
    r, c = np.mgrid[0:7, 0:9]
    base_src = ((r * 7 + c * 3) % 256).astype(np.uint8)
    for device in (False, True):
        base_dst = np.full((7, 9), 0x55, np.uint8)
        if device:
            ts, td = torch.from_numpy(base_src).cuda(), torch.from_numpy(base_dst).cuda()
            zg.Image(ts).view((1, 2, 6, 6)).convolve_separable([1.0], [1.0], zg.BorderMode.zero, out=zg.Image(td).view((1, 2, 6, 6)))
            torch.cuda.synchronize()
            base_dst = td.cpu().numpy()
        else:
            zg.Image(base_src).view((1, 2, 6, 6)).convolve_separable([1.0], [1.0], zg.BorderMode.zero, out=zg.Image(base_dst).view((1, 2, 6, 6)))
        assert np.array_equal(base_dst[2:6, 1:6], base_src[2:6, 1:6])
        mask = np.ones((7, 9), bool)
        mask[2:6, 1:6] = False
        assert np.all(base_dst[mask] == 0x55)


def test_gaussian_sigma_zero_copies_and_negative_raises():  # filters.zig:1159-1180, image.zig:970
    img = np.arange(25, dtype=np.float32).reshape(5, 5)
    assert np.array_equal(zg.Image(img).gaussian_blur(0.0).data, img)
    out = dev(img).gaussian_blur(0.0)
    torch.cuda.synchronize()
    assert np.array_equal(out.to_numpy(), img)
    with pytest.raises(zg.InvalidArgument):
        zg.Image(img).gaussian_blur(-1.0)


def test_dimension_mismatch():  # image.zig:947
    with pytest.raises(zg.DimensionMismatch):
        zg.Image(np.zeros((4, 4), np.uint8)).convolve_separable([1.0], [1.0], 0, out=zg.Image(np.zeros((4, 5), np.uint8)))


# ---- seeded parity sweep ------------------------------------------------------------------------
SIZES = ((1, 1), (3, 5), (2, 9), (17, 31), (257, 63), (100, 300))


@pytest.mark.parametrize("kind", ALL_TYPES)
@pytest.mark.parametrize("border", BORDERS)
def test_parity_small_kernels(oracle, kind, border):
    rng = np.random.default_rng(5)
    for n in (1, 3, 5, 7, 9):
        k = rng.random(n).astype(np.float32)
        k /= k.sum()
        for (rows, cols) in SIZES:
            img = synth(oracle, kind, 100 + n, rows, cols)
            assert_bits_equal(run_dev(img, k, k, border), oracle.conv_separable(img, k, k, border),
                              f"{kind} {rows}x{cols} taps={n} border={border}")


@pytest.mark.parametrize("kind", ALL_TYPES)
def test_parity_two_pass_and_asymmetric(oracle, kind):
    rng = np.random.default_rng(6)
    for nx, ny in ((15, 15), (3, 7), (21, 1), (2, 4), (31, 31)):  # even lengths too: half = n / 2
        kx = (rng.random(nx).astype(np.float32) - np.float32(0.3))
        ky = (rng.random(ny).astype(np.float32) - np.float32(0.3))
        for border in BORDERS:
            img = synth(oracle, kind, 7, 37, 53)
            assert_bits_equal(run_dev(img, kx, ky, border), oracle.conv_separable(img, kx, ky, border),
                              f"{kind} taps=({nx},{ny}) border={border}")


@pytest.mark.parametrize("kind", ("f32", "rgba_f32"))
def test_negligible_taps_are_skipped_only_in_the_interior(oracle, kind):
    # convolution.zig:459-467,541,594: |k| < 1e-10 skipped for interior pixels, not for border pixels
    k = np.array([1e-12, 0.25, 0.5, 0.25, -3e-11], np.float32)
    img = synth(oracle, kind, 8, 40, 70) * np.float32(1e12)  # make k*x visible at f32 precision
    for border in BORDERS:
        assert_bits_equal(run_dev(img, k, k, border), oracle.conv_separable(img, k, k, border), f"skip {kind} {border}")
    k15 = np.zeros(15, np.float32)
    k15[7] = 1.0
    k15[0] = 5e-11
    assert_bits_equal(run_dev(img, k15, k15, 2), oracle.conv_separable(img, k15, k15, 2), "skip two-pass")


@pytest.mark.parametrize("kind", ("u8", "rgba_u8"))
def test_wide_integer_taps_use_the_i64_path(oracle, kind):
    # taps * 256 beyond 2^23: products leave i32, the reference accumulates in i64 and clamps temp to i32
    k = np.array([-40000.0, 70000.0, -29000.0], np.float32)
    k2 = np.array([0.001, 0.002, 0.001], np.float32)
    img = synth(oracle, kind, 9, 33, 65)
    for border in BORDERS:
        assert_bits_equal(run_dev(img, k, k2, border), oracle.conv_separable(img, k, k2, border), f"i64 {kind}")
        assert_bits_equal(run_dev(img, k, k, border), oracle.conv_separable(img, k, k, border), f"i64 clamp {kind}")


@pytest.mark.parametrize("kind", ALL_TYPES)
def test_views_on_both_sides(oracle, kind):
    base = synth(oracle, kind, 10, 64, 96)
    k = oracle.gaussian_kernel(1.0)
    src_t = torch.from_numpy(base).cuda()
    dst_t = torch.zeros_like(src_t)
    zg.Image(src_t).view((5, 3, 85, 60)).convolve_separable(k, k, 2, out=zg.Image(dst_t).view((7, 4, 87, 61)))
    torch.cuda.synchronize()
    want = oracle.conv_separable(base[3:60, 5:85], k, k, 2)
    got = dst_t.cpu().numpy()
    assert_bits_equal(got[4:61, 7:87], want, f"view {kind}")
    got[4:61, 7:87] = 0
    assert not got.any(), "pixels outside the destination view were written"


@pytest.mark.parametrize("kind,sigma", [("rgba_u8", 0.6), ("rgba_f32", 0.6), ("f32", 0.6), ("u8", 1.0), ("rgb_u8", 0.6)])
def test_gaussian_blur_host_and_device_layers(oracle, kind, sigma):
    img = synth(oracle, kind, 12, 123, 211)
    want = oracle.gaussian_blur(img, sigma)
    assert_bits_equal(zg.Image(img).gaussian_blur(sigma).data, want, f"host {kind}")
    out = dev(img).gaussian_blur(sigma)
    torch.cuda.synchronize()
    assert_bits_equal(out.to_numpy(), want, f"device {kind}")


# ---- BASELINE.json configs[1]: 5x5 Gaussian on 4096x4096 RGBA -------------------------------------
@pytest.mark.parametrize("kind", ("u8", "rgb_u8"))
@pytest.mark.parametrize("border", BORDERS)
def test_byte_stream_fast_path(oracle, kind, border):
    """Grey / Rgb u8 rows whose byte length is a multiple of 16 run on the 16-bytes-per-lane kernel (conv_sep_bytes.hip):
    non-negative integer taps summing to <= 257, 3..9 taps; tile seams at 1024 bytes, edges resolved per pixel."""
    rng = np.random.default_rng(11)
    for n in (3, 5, 7, 9):
        t = rng.integers(0, 60, n)
        t[n // 2] += 256 - t.sum()  # taps sum to 256 like every Gaussian the reference builds
        k = (t / 256.0).astype(np.float32)
        for (rows, cols) in ((9, 272), (70, 1040), (33, 2064), (5, 352)):
            img = synth(oracle, kind, 300 + n, rows, cols)
            assert_bits_equal(run_dev(img, k, k, border), oracle.conv_separable(img, k, k, border),
                              f"{kind} {rows}x{cols} taps={n} border={border}")
    # saturating kernel (sum 257 on both axes: 255 * 257 * 257 overflows u8 -> the clamped variant) on a white-ish frame
    k = (np.array([86, 85, 86]) / 256.0).astype(np.float32)
    img = np.maximum(synth(oracle, kind, 9, 40, 528), 250)
    assert_bits_equal(run_dev(img, k, k, border), oracle.conv_separable(img, k, k, border), f"{kind} clamp border={border}")
    # 16-byte aligned view inside a larger frame, into a view
    base = synth(oracle, kind, 10, 64, 1088)
    v = base[7:50, 16:16 + 1040]
    src = zg.Image(torch.from_numpy(base).cuda()).view((16, 7, 16 + 1040, 50))
    dst_base = torch.zeros((64, 1088) + base.shape[2:], dtype=torch.uint8, device="cuda")
    dst = zg.Image(dst_base).view((32, 3, 32 + 1040, 46))
    kk = (np.array([20, 60, 96, 60, 20]) / 256.0).astype(np.float32)
    src.convolve_separable(kk, kk, border, out=dst)
    torch.cuda.synchronize()
    want = oracle.conv_separable(np.ascontiguousarray(v), kk, kk, border)
    got = dst_base.cpu().numpy()
    assert_bits_equal(got[3:46, 32:32 + 1040], want, f"{kind} view border={border}")
    got[3:46, 32:32 + 1040] = 0
    assert not got.any(), "pixels outside the destination view were written"


@pytest.mark.parametrize("kind", ("u8", "rgb_u8", "rgba_u8"))
@pytest.mark.parametrize("border", BORDERS)
def test_long_kernels_packed_two_pass(oracle, kind, border):
    """11..65-tap (and unequal / even-length) non-negative kernels on 16-byte-multiple rows: conv_sep_bytes2.hip."""
    rng = np.random.default_rng(12)

    def taps(n):
        t = rng.integers(0, 256 // n + 1, n)
        t[n // 2] += 255 - t.sum()
        assert 0 <= t.min() and t.max() <= 255
        return (t / 256.0).astype(np.float32)

    for nx, ny in ((11, 11), (13, 13), (17, 17), (35, 35), (65, 65), (3, 21), (20, 6), (1, 33)):
        kx, ky = taps(nx), taps(ny)
        for (rows, cols) in ((9, 272), (70, 1040), (45, 352)):
            img = synth(oracle, kind, 500 + nx, rows, cols)
            assert_bits_equal(run_dev(img, kx, ky, border), oracle.conv_separable(img, kx, ky, border),
                              f"{kind} {rows}x{cols} taps=({nx},{ny}) border={border}")
    # the Gaussians an ORB pyramid asks for (sigma = 1.6 * sqrt(1.2^(2i) - 1), pyramid.zig:76-85)
    img = synth(oracle, kind, 77, 96, 528)
    for level in (2, 3, 5, 7):
        sigma = float(np.float32(1.6) * np.sqrt(np.float32(1.2 ** level) ** 2 - np.float32(1.0)))
        out = dev(img).gaussian_blur(sigma)
        torch.cuda.synchronize()
        assert_bits_equal(out.to_numpy(), oracle.gaussian_blur(img, sigma), f"{kind} pyramid level {level} sigma={sigma}")
    # saturating taps (sum 257 both ways) on a near-white frame: the clamped column pass
    k = np.zeros(11, np.float32); k[[0, 5, 10]] = np.array([86, 85, 86], np.float32) / 256
    img = np.maximum(synth(oracle, kind, 9, 40, 528), 250)
    assert_bits_equal(run_dev(img, k, k, border), oracle.conv_separable(img, k, k, border), f"{kind} clamp border={border}")


@pytest.mark.parametrize("kind", ("f32", "rgb_f32", "rgba_f32"))
@pytest.mark.parametrize("border", BORDERS)
def test_long_kernels_f32_two_pass(oracle, kind, border):
    """11..65-tap (and unequal / even-length) kernels on f32 rows whose length is a multiple of 4 elements: conv_sep_f32long.hip."""
    rng = np.random.default_rng(13)
    for nx, ny in ((11, 11), (13, 13), (17, 17), (35, 35), (65, 65), (3, 21), (20, 6), (1, 33)):
        kx = (rng.random(nx).astype(np.float32) - np.float32(0.3))
        ky = (rng.random(ny).astype(np.float32) - np.float32(0.3))
        for (rows, cols) in ((9, 272), (70, 1040), (45, 352)):
            if (rows, cols) != (45, 352) and (nx, ny) in ((35, 35), (65, 65), (20, 6)) and border in (BORDERS[0], BORDERS[3]):
                continue
            img = synth(oracle, kind, 600 + nx, rows, cols)
            assert_bits_equal(run_dev(img, kx, ky, border), oracle.conv_separable(img, kx, ky, border),
                              f"{kind} {rows}x{cols} taps=({nx},{ny}) border={border}")
    # pyramid / canny sigmas, and non-finite pixels: a tap outside the kernel must not exist (inf * 0 would be NaN)
    img = synth(oracle, kind, 78, 96, 528)
    img[40, 100] = np.inf
    img[41, 300] = -0.0
    for sigma in (1.4, 2.25, 4.0):
        out = dev(img).gaussian_blur(sigma)
        torch.cuda.synchronize()
        assert_bits_equal(out.to_numpy(), oracle.gaussian_blur(img, sigma), f"{kind} sigma={sigma}")


@pytest.mark.parametrize("kind", ("rgba_f32", "rgba_u8", "u8", "rgb_u8"))
def test_config2_full_size(oracle, kind):
    img = synth(oracle, kind, 2, 4096, 4096)
    out = dev(img).gaussian_blur(0.6)
    torch.cuda.synchronize()
    got = out.to_numpy()
    # size-independent property: a normalised kernel leaves a constant frame constant (u8 taps sum to 256)
    const = np.full_like(img[:64, :512], 77 if kind != "rgba_f32" else np.float32(0.5))
    cout = dev(const).gaussian_blur(0.6)
    torch.cuda.synchronize()
    if kind != "rgba_f32":
        assert np.all(cout.to_numpy() == 77)
    # full comparison against the oracle (a few seconds of CPU)
    assert_bits_equal(got, oracle.gaussian_blur(img, 0.6), f"4096^2 {kind}")
    if kind == "rgba_u8":  # the 3- and 7-tap forms of the large-frame kernel variant
        for sigma in (0.3, 1.0, 2.5):  # 2.5: 17 taps, the packed two-pass path
            out = dev(img).gaussian_blur(sigma)
            torch.cuda.synchronize()
            assert_bits_equal(out.to_numpy(), oracle.gaussian_blur(img, sigma), f"4096^2 {kind} sigma={sigma}")


@pytest.mark.gpu
def test_stream_kernel_is_deterministic_at_full_size(oracle):
    """Twelve runs of the 4096^2 Rgba(u8) Gaussian, each equal to the oracle. A chip full of waves is what it takes to hit the store-data
    hazard described in conv_sep_stream.hip (an even row stored with part of the odd row): small frames never showed it, and the wrong
    pixels (0.05 %) differed from run to run. Also a frame whose rows end inside a strip (1920 pixels: 7.5 strips) and the 3- / 7-tap forms."""
    img = synth(oracle, "rgba_u8", 31, 4096, 4096)
    want = oracle.gaussian_blur(img, 0.6)
    d = dev(img)
    for run in range(12):
        out = d.gaussian_blur(0.6)
        torch.cuda.synchronize()
        assert_bits_equal(out.to_numpy(), want, f"4096^2 rgba_u8, run {run}")
    img = synth(oracle, "rgba_u8", 32, 1080, 1920)
    for sigma in (0.3, 0.6, 1.0):
        want = oracle.gaussian_blur(img, sigma)
        for run in range(3):
            out = dev(img).gaussian_blur(sigma)
            torch.cuda.synchronize()
            assert_bits_equal(out.to_numpy(), want, f"1080p rgba_u8 sigma={sigma}, run {run}")
    for kind, border in (("rgb_u8", zg.BorderMode.replicate), ("rgba_u8", zg.BorderMode.wrap), ("rgba_u8", zg.BorderMode.zero), ("rgb_u8", zg.BorderMode.mirror)):
        img = synth(oracle, kind, 33, 600, 1360)  # 4080 / 5440 bytes per row: the last strip is partial
        k = np.array([1, 4, 6, 4, 1], np.float32) / 16
        out = dev(img).convolve_separable(k, k, border)
        torch.cuda.synchronize()
        assert_bits_equal(out.to_numpy(), oracle.conv_separable(img, k, k, int(border)), f"{kind} 600x1360 border={border}")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ("rgba_f32", "rgba_u8", "f32"))
def test_config2_binomial_zero_border_full_size(oracle, kind):
    """SURVEY 8(d)'s second input for config 2: the binomial [1, 4, 6, 4, 1] / 16 with BorderMode.zero through convolveSeparable, 4096^2."""
    k = np.array([1, 4, 6, 4, 1], np.float32) / 16
    img = synth(oracle, kind, 22, 4096, 4096)
    out = dev(img).convolve_separable(k, k, zg.BorderMode.zero)
    torch.cuda.synchronize()
    assert_bits_equal(out.to_numpy(), oracle.conv_separable(img, k, k, oracle.ZERO), f"4096^2 {kind} binomial .zero")


@pytest.mark.gpu
def test_c_abi_without_torch_objects(oracle):
    """The runtime entry points on their own: device memory from zg_malloc, a stream from zg_stream_create, copies through
    zg_memcpy_h2d / d2h — the way a Zig or C caller drives the library, no torch tensor anywhere near the pixels."""
    import ctypes as C
    from zignal_amd import _lib as L
    lib = zg.lib()
    assert lib.zg_device_count() >= 1
    host = oracle.synth_f32(21, (67, 130, 4))
    want = oracle.gaussian_blur(host, 0.6)
    nbytes = host.nbytes
    src, dst, stream = C.c_void_p(), C.c_void_p(), C.c_void_p()
    L.check(lib.zg_malloc(C.byref(src), nbytes))
    L.check(lib.zg_malloc(C.byref(dst), nbytes))
    L.check(lib.zg_stream_create(C.byref(stream)))
    try:
        L.check(lib.zg_memcpy_h2d(src, host.ctypes.data, nbytes, stream))
        s = L.ZgImage(src.value, 130, 67, 130, L.PIXEL_RGBA_F32)
        d = L.ZgImage(dst.value, 130, 67, 130, L.PIXEL_RGBA_F32)
        L.check(lib.zg_gaussian_blur(C.byref(s), C.byref(d), C.c_float(0.6), stream))
        L.check(lib.zg_stream_synchronize(stream))
        got = np.empty_like(host)
        L.check(lib.zg_memcpy_d2h(got.ctypes.data, dst, nbytes, stream))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        # a view described by hand: stride in pixels, origin inside the allocation; pixels outside it stay untouched
        zeros = np.zeros_like(host)  # kept alive across the call: .ctypes.data of a temporary would dangle
        L.check(lib.zg_memcpy_h2d(dst, zeros.ctypes.data, nbytes, stream))
        sv = L.ZgImage(src.value + (3 * 130 + 5) * 16, 130, 40, 100, L.PIXEL_RGBA_F32)
        dv = L.ZgImage(dst.value + (3 * 130 + 5) * 16, 130, 40, 100, L.PIXEL_RGBA_F32)
        L.check(lib.zg_gaussian_blur(C.byref(sv), C.byref(dv), C.c_float(0.6), stream))
        L.check(lib.zg_memcpy_d2h(got.ctypes.data, dst, nbytes, stream))
        sub = oracle.gaussian_blur(np.ascontiguousarray(host[3:43, 5:105]), 0.6)
        assert np.array_equal(got[3:43, 5:105].view(np.uint32), sub.view(np.uint32))
        got[3:43, 5:105] = 0
        assert not got.any()
        with pytest.raises(zg.DimensionMismatch):
            bad = L.ZgImage(dst.value, 130, 66, 130, L.PIXEL_RGBA_F32)
            L.check(lib.zg_gaussian_blur(C.byref(s), C.byref(bad), C.c_float(0.6), stream))
    finally:
        L.check(lib.zg_stream_destroy(stream))
        L.check(lib.zg_free(src))
        L.check(lib.zg_free(dst))


def test_convolve_u8_f32_accumulators_at_the_exactness_boundary(oracle):
    """k_conv2d's MODE 3 (u8 pixels, f32 accumulators) is chosen when 255 * sum|round(256 k)| < 2^24; kernels just below and just
    above that line, positive and mixed-sign, on frames of all-255 / all-0 / random bytes — sums that reach the largest partial
    values — must equal the oracle's i64 arithmetic."""
    rng = np.random.default_rng(9)
    frames = {"ones": np.full((40, 300, 4), 255, np.uint8), "rand": rng.integers(0, 256, (37, 261, 4), dtype=np.uint8),
              "grey": rng.integers(0, 256, (70, 515), dtype=np.uint8), "greymax": np.full((33, 200), 255, np.uint8)}
    for n in (3, 5, 7):
        line = (1 << 24) / 255 / 256 / (n * n)  # a tap value for which 255 * sum|.| sits on the line when all n^2 taps share it
        for scale in (0.97, 0.999, 1.001, 1.05):
            for signs in ("plus", "mixed"):
                k = np.full((n, n), line * scale, np.float32)
                if signs == "mixed":
                    k[::2, 1::2] *= -1
                    k[n // 2, n // 2] *= -1
                for name, f in frames.items():
                    want = oracle.convolve(f, k, oracle.MIRROR)
                    got = dev(f).convolve(k)
                    torch.cuda.synchronize()
                    assert_bits_equal(got.to_numpy(), want, f"{n}x{n} scale {scale} {signs} {name}")


def test_convolve_3x3_5x5_u8_register_stream_kernel(oracle):
    """k_conv2d_stream (conv2d_stream.hip): shapes that meet its preconditions (row bytes % 16 == 0, >= 64 pixels, >= 16 rows) over one
    and several strips across and down, every border rule, every u8 pixel type, positive and mixed-sign taps (negative sums clamp to 0,
    large ones to 255), and a view whose pitch is wider than its rows."""
    rng = np.random.default_rng(31)
    shapes = {"u8": [(16, 64), (33, 1040), (70, 2064), (200, 4096)], "rgb_u8": [(16, 64), (45, 352), (37, 1024)], "rgba_u8": [(16, 64), (40, 300), (97, 260), (130, 1028), (300, 516)]}
    for kind, shp in shapes.items():
        ch = {"u8": (), "rgb_u8": (3,), "rgba_u8": (4,)}[kind]
        for shape in shp:
            src = rng.integers(0, 256, shape + ch, dtype=np.uint8)
            for n in (3, 5):
                for signs in ("blur", "mixed", "big"):
                    k = rng.random((n, n), dtype=np.float32) / (n * n)
                    if signs == "mixed":
                        k[::2, 1::2] *= -2
                    if signs == "big":
                        k *= 3
                    for border in range(4):
                        got = dev(src).convolve(k, border)
                        torch.cuda.synchronize()
                        assert_bits_equal(got.to_numpy(), oracle.convolve(src, k, border), f"{kind} {shape} {n}x{n} {signs} border {border}")
    # a view: 16-byte aligned origin, pitch of the parent
    parent = rng.integers(0, 256, (90, 400, 4), dtype=np.uint8)
    k = rng.random((3, 3), dtype=np.float32) / 9
    t = torch.from_numpy(parent).cuda()
    out = torch.full((90, 400, 4), 7, dtype=torch.uint8, device="cuda")
    zg.Image(t).view((8, 5, 8 + 256, 5 + 60)).convolve(k, 2, out=zg.Image(out).view((8, 5, 8 + 256, 5 + 60)))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert_bits_equal(got[5:65, 8:264], oracle.convolve(np.ascontiguousarray(parent[5:65, 8:264]), k, 2), "view")
    got[5:65, 8:264] = 7
    assert (got == 7).all(), "pixels outside the destination view were written"
