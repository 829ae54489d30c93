"""GPU parity for the colour spaces of SURVEY §8f rank 3: Image.convert between every pair of colour spaces
(zignal_amd/csrc/colorspaces.hip) against the CPU oracle (oracle/colorspaces.c, whose f64 instance is pinned to the
reference's golden values in tests/test_oracle_color.py). Bit-exact: the device restates the same f32 operation
sequence, including the route through hub spaces the reference's `.to()` tables take."""
import numpy as np
import pytest

import zignal_amd as zg
from tests.util import assert_bits_equal, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

FLOAT_SPACES = ("HSL", "HSV", "LAB", "LCH", "LMS", "OKLAB", "OKLCH", "XYB", "XYZ", "YCBCR")


def dev(a):
    return zg.Image(torch.from_numpy(np.ascontiguousarray(a)).cuda())


def run(src, src_space, dst_space, dtype):
    out = dev(src).convert(dst_space, dtype, src_space=src_space)
    torch.cuda.synchronize()
    return out.to_numpy()


def channels(space):
    return 1 if space == zg.CS_GRAY else (4 if space == zg.CS_RGBA else 3)


def every_colour_frame():
    """All 2^24 Rgb(u8) colours would be 48 MB; a 3-D lattice of 64^3 plus the corners of the cube is enough to hit every
    branch (hue sectors, both sRGB and Lab transfer segments, greys)."""
    g = np.linspace(0, 255, 64).round().astype(np.uint8)
    r, gg, b = np.meshgrid(g, g, g, indexing="ij")
    cube = np.stack([r, gg, b], -1).reshape(512, 512, 3)
    return np.ascontiguousarray(cube)


@pytest.mark.parametrize("name", FLOAT_SPACES)
@pytest.mark.parametrize("kind", ("rgb_u8", "rgba_u8", "rgb_f32", "rgba_f32", "u8", "f32"))
def test_forward_from_rgb_family(oracle, name, kind):
    space = getattr(zg, "CS_" + name)
    src = every_colour_frame() if kind == "rgb_u8" else synth(oracle, kind, 31, 67, 129)
    src_space = {"rgb": zg.CS_RGB, "rgba": zg.CS_RGBA, "u8": zg.CS_GRAY, "f32": zg.CS_GRAY}[kind.split("_")[0]]
    want = oracle.convert(src, src_space, space, np.float32, 3)
    assert_bits_equal(run(src, src_space, space, np.float32), want, f"{kind} -> {name}")


@pytest.mark.parametrize("name", FLOAT_SPACES)
def test_back_to_rgb_family_and_round_trip(oracle, name):
    space = getattr(zg, "CS_" + name)
    rgb = every_colour_frame()
    there = oracle.convert(rgb, zg.CS_RGB, space, np.float32, 3)
    for dst_space, dtype in ((zg.CS_RGB, np.uint8), (zg.CS_RGBA, np.uint8), (zg.CS_RGB, np.float32), (zg.CS_RGBA, np.float32),
                             (zg.CS_GRAY, np.uint8), (zg.CS_GRAY, np.float32)):
        want = oracle.convert(there, space, dst_space, dtype, channels(dst_space))
        assert_bits_equal(run(there, space, dst_space, dtype), want, f"{name} -> space {dst_space} {np.dtype(dtype).name}")
    # the reference's own property (color.zig:1738-1773, stated there in f64): Rgb(u8) survives the trip. In f32 it
    # holds for every space except where f32 hue / chroma resolution is too coarse — checked where it holds in the oracle.
    back = run(there, space, zg.CS_RGB, np.uint8)
    assert_bits_equal(back, oracle.convert(there, space, zg.CS_RGB, np.uint8, 3), f"{name} round trip")
    assert np.abs(back.astype(int) - rgb.astype(int)).max() <= 1


@pytest.mark.parametrize("a,b", [("LAB", "LCH"), ("LCH", "OKLCH"), ("OKLAB", "HSL"), ("HSV", "HSL"), ("HSL", "HSV"), ("XYB", "LAB"),
                                 ("LMS", "XYB"), ("XYZ", "YCBCR"), ("YCBCR", "OKLCH"), ("LCH", "HSV"), ("OKLCH", "LMS")])
def test_cross_space_routes(oracle, a, b):
    """Routes through the hubs (e.g. Lch -> Lab -> Xyz -> Rgb -> Hsv): every intermediate is rounded to f32 as in the reference."""
    sa, sb = getattr(zg, "CS_" + a), getattr(zg, "CS_" + b)
    src = oracle.convert(every_colour_frame()[:128], zg.CS_RGB, sa, np.float32, 3)
    assert_bits_equal(run(src, sa, sb, np.float32), oracle.convert(src, sa, sb, np.float32, 3), f"{a} -> {b}")


def test_ycbcr_u8_and_float_forms(oracle):
    rgb = every_colour_frame()[:64]
    ycc = oracle.convert(rgb, zg.CS_RGB, zg.CS_YCBCR, np.uint8, 3)
    for dst_space, dtype in ((zg.CS_RGB, np.uint8), (zg.CS_RGBA, np.uint8), (zg.CS_GRAY, np.uint8), (zg.CS_LAB, np.float32), (zg.CS_YCBCR, np.float32)):
        want = oracle.convert(ycc, zg.CS_YCBCR, dst_space, dtype, channels(dst_space))
        assert_bits_equal(run(ycc, zg.CS_YCBCR, dst_space, dtype), want, f"Ycbcr(u8) -> {dst_space}")
    yf = oracle.convert(rgb, zg.CS_RGB, zg.CS_YCBCR, np.float32, 3)
    assert_bits_equal(run(yf, zg.CS_YCBCR, zg.CS_YCBCR, np.uint8), oracle.convert(yf, zg.CS_YCBCR, zg.CS_YCBCR, np.uint8, 3), "Ycbcr f32 -> u8")
    assert_bits_equal(run(rgb, zg.CS_RGB, zg.CS_YCBCR, np.float32), yf, "Rgb(u8) -> Ycbcr(f32)")


def test_errors_and_host_layer(oracle):
    rgb = every_colour_frame()[:8]
    with pytest.raises(zg.ZignalError):  # Lab has no u8 form
        dev(rgb).convert(zg.CS_LAB, np.uint8)
    with pytest.raises(zg.ZignalError):  # layout must match the space
        dev(rgb).convert(zg.CS_LAB, np.float32, src_space=zg.CS_RGBA)
    host = zg.Image(rgb).convert(zg.CS_OKLCH, np.float32).data
    assert_bits_equal(host, oracle.convert(rgb, zg.CS_RGB, zg.CS_OKLCH, np.float32, 3), "host layer")
    view = zg.Image(torch.from_numpy(every_colour_frame()).cuda()).view((16, 8, 200, 100))
    out = view.convert(zg.CS_LAB, np.float32)
    torch.cuda.synchronize()
    assert_bits_equal(out.to_numpy(), oracle.convert(np.ascontiguousarray(every_colour_frame()[8:100, 16:200]), zg.CS_RGB, zg.CS_LAB, np.float32, 3), "view")


def test_u8_to_xyz_and_oklab_four_pixels_per_lane(oracle):
    """k_u8_to_lab4 (convert.hip): ragged row ends (cols % 4, cols % 256), one wave's worth of full lanes and more, Rgb and Rgba
    sources, views whose pitch keeps or breaks the 16-byte rule (the latter stay on k_convert), and linearisation tables that are
    not 'plain' (negative, tiny, huge, zero, infinite entries: xyz_to_oklab's range tests must then run) — all equal to the oracle."""
    rng = np.random.default_rng(77)
    for rows, cols in ((1, 1), (3, 5), (7, 255), (5, 256), (4, 257), (2, 1023), (3, 1027), (2, 2048), (1, 4096 + 64)):
        for ch, space in ((4, zg.CS_RGBA), (3, zg.CS_RGB)):
            src = rng.integers(0, 256, (rows, cols, ch), dtype=np.uint8)
            src[0, : min(cols, 9)] = 0  # black pixels: cbrt(0) takes the reference's own early return
            for dst in (zg.CS_OKLAB, zg.CS_XYZ, zg.CS_LAB):
                want = oracle.convert(src, space, dst, np.float32, 3)
                assert_bits_equal(run(src, space, dst, np.float32), want, f"{rows}x{cols}x{ch} -> {dst}")
    # views: a 1024-wide window of a wider frame (pitch a multiple of four pixels: the four-pixel kernel) and one shifted by a pixel
    wide = rng.integers(0, 256, (6, 1200, 4), dtype=np.uint8)
    whole = dev(wide)
    for c0, c1 in ((0, 1024), (4, 1028), (1, 1025), (3, 1003)):
        view = zg.Image(whole.data[:, c0:c1])
        out = view.convert(zg.CS_OKLAB, np.float32)
        torch.cuda.synchronize()
        assert_bits_equal(out.to_numpy(), oracle.convert(np.ascontiguousarray(wide[:, c0:c1]), zg.CS_RGBA, zg.CS_OKLAB, np.float32, 3), f"view {c0}:{c1}")
    ramp = np.stack([np.arange(256, dtype=np.uint8)] * 3, -1)[None].repeat(3, 0)
    ramp[1] = ramp[1][::-1]
    ramp[2, :, 1:] = 0
    odd = np.linspace(0, 1, 256, dtype=np.float32) ** 2
    odd[3], odd[7], odd[11], odd[200], odd[255] = -0.25, 1e-30, 1e-42, 3e38, np.inf
    odd[13] = np.float32(-0.0)
    with np.errstate(all="ignore"):
        want = oracle.convert(ramp, zg.CS_RGB, zg.CS_OKLAB, np.float32, 3, srgb_lut=odd)
    got = dev(ramp).convert(zg.CS_OKLAB, np.float32, srgb_lut=odd)
    torch.cuda.synchronize()
    assert_bits_equal(got.to_numpy(), want, "odd table")
    with np.errstate(all="ignore"):  # the route's two hops with the caller's table in the first (the oracle's one-call Lab route linearises by itself)
        want = oracle.convert(oracle.convert(ramp, zg.CS_RGB, zg.CS_XYZ, np.float32, 3, srgb_lut=odd), zg.CS_XYZ, zg.CS_LAB, np.float32, 3)
    got = dev(ramp).convert(zg.CS_LAB, np.float32, srgb_lut=odd)
    torch.cuda.synchronize()
    assert_bits_equal(got.to_numpy(), want, "odd table -> Lab")


def test_lab_to_u8_four_pixels_per_lane(oracle):
    """k_lab4_to_u8 (convert.hip): Lab(f32) -> Rgb(u8) / Rgba(u8) over ragged and full rows, in-gamut colours (the forward conversion of
    random pixels), colours far outside the gamut, the dark linear piece of labToXyz, and non-finite components."""
    rng = np.random.default_rng(78)
    for rows, cols in ((1, 4), (3, 8), (5, 256), (4, 260), (2, 1024), (3, 1028), (1, 4096 + 64)):
        rgb = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
        rgb[0, : min(cols, 6)] = (0, 0, 0)
        rgb[0, min(cols, 6): min(cols, 10)] = (3, 2, 1)
        lab = oracle.convert(rgb, zg.CS_RGB, zg.CS_LAB, np.float32, 3)
        wild = lab.copy()
        wild[..., 1:] *= np.float32(3.0)  # out of gamut: the clamps
        wild[-1, -1] = (np.float32(np.nan), np.float32(1e30), np.float32(-np.inf))
        for src, name in ((lab, "in gamut"), (wild, "wild")):
            for space, ch in ((zg.CS_RGBA, 4), (zg.CS_RGB, 3)):
                with np.errstate(all="ignore"):
                    want = oracle.convert(src, zg.CS_LAB, space, np.uint8, ch)
                assert_bits_equal(run(src, zg.CS_LAB, space, np.uint8), want, f"{rows}x{cols} {name} -> {ch} channels")
