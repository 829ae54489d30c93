"""GPU parity of the f32 tile-per-wave kernel (zignal_amd/csrc/conv_sep_tile_f32.hip) and of the multi-plane entry points
(zg_conv_separable_planes / zg_gaussian_blur_planes): Image(f32).convolveSeparable / gaussianBlur (reference
src/image/convolution.zig:441-647, src/image.zig:954-994) on the shapes the kernel takes — cols % 4 == 0, cols >= 64, rows >= 16,
3 / 5 / 7 taps — against the CPU oracle, bit for bit, in both of the kernel's halo forms."""
import os

import numpy as np
import pytest

import zignal_amd as zg
from tests.util import assert_bits_equal

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

BORDERS = (zg.BorderMode.zero, zg.BorderMode.replicate, zg.BorderMode.mirror, zg.BorderMode.wrap)
# one-lane-wide last tiles are excluded by the kernel ((cols * 4) % 1024 == 16 falls back): 260 is such a width, kept here on purpose
SHAPES = ((16, 64), (17, 68), (100, 300), (64, 256), (65, 260), (40, 1024), (300, 516), (33, 2052), (129, 1280))


def dev(a):
    return zg.Image(torch.from_numpy(np.ascontiguousarray(a)).cuda())


def signed_plane(oracle, seed, rows, cols):
    """Uniform values of both signs with zeros of both signs sprinkled in: 0 + (-0 * k) is +0, so an accumulator that is not
    started from zero the way the reference starts it shows up as a sign bit."""
    a = oracle.synth_f32(seed, (rows, cols)) - np.float32(0.5)
    a[::7, ::5] = np.float32(-0.0)
    a[3::11, 2::9] = np.float32(0.0)
    return a


@pytest.mark.parametrize("border", BORDERS)
@pytest.mark.parametrize("n", (3, 5, 7))
def test_parity_every_shape_class(oracle, n, border):
    rng = np.random.default_rng(50 + n)
    kx = (rng.random(n).astype(np.float32) - np.float32(0.2))  # asymmetric, mixed signs: tap order matters
    ky = (rng.random(n).astype(np.float32) - np.float32(0.2))
    for (rows, cols) in SHAPES:
        img = signed_plane(oracle, 300 + n, rows, cols)
        got = dev(img).convolve_separable(kx, ky, border)
        torch.cuda.synchronize()
        assert_bits_equal(got.to_numpy(), oracle.conv_separable(img, kx, ky, border), f"f32 {rows}x{cols} taps={n} border={border}")


@pytest.mark.parametrize("halo", ("all", "outer"))
def test_both_halo_forms_give_the_same_bits(oracle, halo):
    """The narrow halo loads issued by every lane (what a one-plane launch uses: "all") and by the tile's two outer lanes only (what a launch of
    several planes uses: "outer"), each on every shape class — ragged last tile row, partial last tile column, planes one tile high —, every
    border, 3 / 5 / 7 taps. The form follows the plane count, so the same plane goes in once and twice."""
    for n in (3, 5, 7):
        rng = np.random.default_rng(70 + n)
        kx = (rng.random(n).astype(np.float32) - np.float32(0.2))
        ky = (rng.random(n).astype(np.float32) - np.float32(0.2))
        for (rows, cols) in SHAPES:
            img = signed_plane(oracle, 500 + n, rows, cols)
            for border in BORDERS:
                want = oracle.conv_separable(img, kx, ky, border)
                if halo == "all":
                    gots = [dev(img).convolve_separable(kx, ky, border)]
                else:
                    gots = zg.convolve_separable_planes([dev(img), dev(img)], kx, ky, border)
                torch.cuda.synchronize()
                for got in gots:
                    assert_bits_equal(got.to_numpy(), want, f"halo {halo}: {rows}x{cols} taps={n} border={border}")
    planes = [signed_plane(oracle, 600 + p, 50, 516) for p in range(5 if halo == "outer" else 1)]
    k = oracle.gaussian_kernel(0.6)
    outs = zg.gaussian_blur_planes([dev(p) for p in planes], 0.6)
    torch.cuda.synchronize()
    for p, o in zip(planes, outs):
        assert_bits_equal(o.to_numpy(), oracle.conv_separable(p, k, k, 2), f"halo {halo}: {len(planes)} planes")


def test_a_plane_that_ends_with_its_allocation(oracle):
    """the FAST path loads through a whole-plane descriptor, whose range check sees the plane, not the row. In a partial last strip
    the lanes past the row's end used to load from offsets past it: harmless inside the plane (the next row), but in the plane's last row up to
    1 KiB past its end. The plane here is the tail of its allocation (a view that ends with the buffer's last byte) and its last tile is a FAST
    one (rows % 8 == 0 with 5 taps: the tile above the last takes rows up to rows - 1 ... the shapes cover both residues); results against the
    oracle, and a canary plane right behind a second copy stays untouched."""
    for rows, cols in ((1000, 1032), (1002, 1032), (64, 264), (66, 2056)):  # row bytes % 1024 = 32: a last strip of two lanes
        img = signed_plane(oracle, 900 + rows, rows, cols)
        buf = torch.empty(rows * cols + 4096, dtype=torch.float32, device="cuda")
        buf.fill_(float("nan"))
        tail = buf[-rows * cols:].view(rows, cols)  # ends exactly with the allocation
        tail.copy_(torch.from_numpy(img))
        for n in (3, 5, 7):
            kx = (np.random.default_rng(90 + n).random(n).astype(np.float32) - np.float32(0.2))
            got = zg.Image(tail).convolve_separable(kx, kx, 2)
            torch.cuda.synchronize()
            assert_bits_equal(got.to_numpy(), oracle.conv_separable(img, kx, kx, 2), f"tail plane {rows}x{cols} taps={n}")
        assert torch.isnan(buf[:4096]).all()


def test_matches_the_tiled_kernel_and_the_oracle_at_full_size(oracle):
    """BASELINE configs[1] as a zignal caller can express it: gaussianBlur(0.6) on a 4096 x 4096 Image(f32) plane."""
    img = oracle.synth_f32(2, (4096, 4096))
    want = oracle.gaussian_blur(img, 0.6)
    d = dev(img)
    out = d.gaussian_blur(0.6)
    torch.cuda.synchronize()
    assert_bits_equal(out.to_numpy(), want, "gaussianBlur(0.6) 4096^2 f32 plane")
    # (the LDS-tiled kernel behind ZIGNAL_HIP_NO_TILE_F32 is held to the same oracle in a child process: tests/test_gpu_runtime.py)
    for _ in range(6):  # waves share nothing, so the result must not depend on how they interleave
        again = d.gaussian_blur(0.6)
        torch.cuda.synchronize()
        assert torch.equal(out.data, again.data)


def test_views_on_both_sides(oracle):
    base = signed_plane(oracle, 10, 96, 400)
    k = oracle.gaussian_kernel(0.6)
    src_t = torch.from_numpy(base).cuda()
    dst_t = torch.full_like(src_t, 7.0)
    # 16-byte aligned view origins (columns 8 and 12), widths % 4 == 0: the tile kernel takes them; stride > cols on both sides
    zg.Image(src_t).view((8, 3, 8 + 320, 3 + 80)).convolve_separable(k, k, 2, out=zg.Image(dst_t).view((12, 5, 12 + 320, 5 + 80)))
    torch.cuda.synchronize()
    got = dst_t.cpu().numpy()
    assert_bits_equal(got[5:85, 12:332], oracle.conv_separable(base[3:83, 8:328], k, k, 2), "view f32")
    got[5:85, 12:332] = 7.0
    assert np.all(got == 7.0), "pixels outside the destination view were written"


@pytest.mark.parametrize("n_planes", (1, 4, 8, 9, 11))
def test_planes_equal_single_plane_calls(oracle, n_planes):
    rows, cols = 48, 324
    planes = [signed_plane(oracle, 900 + p, rows, cols) for p in range(n_planes)]
    k = oracle.gaussian_kernel(0.6)
    for border in (zg.BorderMode.mirror, zg.BorderMode.zero):
        outs = zg.convolve_separable_planes([dev(p) for p in planes], k, k, border)
        torch.cuda.synchronize()
        for p, o in zip(planes, outs):
            assert_bits_equal(o.to_numpy(), oracle.conv_separable(p, k, k, border), f"{n_planes} planes border={border}")
    outs = zg.gaussian_blur_planes([dev(p) for p in planes], 0.6)
    torch.cuda.synchronize()
    for p, o in zip(planes, outs):
        assert_bits_equal(o.to_numpy(), oracle.gaussian_blur(p, 0.6), f"gaussianBlur over {n_planes} planes")


def test_planes_of_one_allocation_and_mixed_shapes(oracle):
    """The four planes of a channel-major RGBA f32 frame (one allocation), then a list whose members cannot share a launch."""
    chw = oracle.synth_f32(5, (4, 64, 512))
    t = torch.from_numpy(chw).cuda()
    o = torch.empty_like(t)
    zg.gaussian_blur_planes([zg.Image(t[c]) for c in range(4)], 0.6, outs=[zg.Image(o[c]) for c in range(4)])
    torch.cuda.synchronize()
    for c in range(4):
        assert_bits_equal(o[c].cpu().numpy(), oracle.gaussian_blur(chw[c], 0.6), f"channel-major plane {c}")
    mixed = [oracle.synth_f32(1, (32, 128)), oracle.synth_f32(2, (32, 132)), oracle.synth_u8(3, (32, 128)), oracle.synth_f32(4, (20, 64, 4)),
             oracle.synth_f32(6, (32, 128))]
    outs = zg.gaussian_blur_planes([dev(m) for m in mixed], 1.0)  # 7 taps; u8 and Rgba(f32) members take their own kernels
    torch.cuda.synchronize()
    for m, got in zip(mixed, outs):
        assert_bits_equal(got.to_numpy(), oracle.gaussian_blur(m, 1.0), f"mixed list {m.shape} {m.dtype}")


def test_planes_errors_are_those_of_the_single_plane_call():
    a, b = dev(np.zeros((32, 128), np.float32)), dev(np.zeros((32, 132), np.float32))
    with pytest.raises(zg.DimensionMismatch):  # image.zig:947 / :962
        zg.gaussian_blur_planes([a, a], 0.6, outs=[a._like(), b])
    with pytest.raises(zg.InvalidArgument):    # image.zig:970
        zg.gaussian_blur_planes([a], -1.0)
    same = zg.gaussian_blur_planes([a], 0.0)   # image.zig:966: sigma 0 copies
    torch.cuda.synchronize()
    assert torch.equal(same[0].data, a.data)
    assert zg.gaussian_blur_planes([], 0.6) == []
