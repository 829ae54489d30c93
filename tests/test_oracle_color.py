"""Pins the oracle's colour-space restatement (oracle/colorspaces.c) to the reference's own unit tests in
src/color.zig: the exact f64 Hsl / Hsv / Lab values of `testRoundTripConversion` (:1641-1725, compared with
expectEqualDeep, i.e. bit for bit), the grey known answers (:1555-1561), `ColorSpace.convert` (:1795-1806) and the
"100 random colors" round-trip property over every colour space (:1738-1773). CPU only.

The Lab values go through std.math.pow twice (gammaToLinear, labForward): matching them to the last bit is the evidence
that the oracle's restatement of Zig's pow (Go's algorithm over fdlibm exp / log) is the reference's."""
import numpy as np
import pytest

# (Rgb(u8), Hsl(f64), Hsv(f64), Lab(f64)) — src/color.zig:1641-1725, committed as a fixture
import json
import os

with open(os.path.join(os.path.dirname(__file__), "golden", "color_zig_roundtrip_f64.json")) as _f:
    GOLDEN = [(tuple(v["rgb_u8"]), tuple(v["hsl_f64"]), tuple(v["hsv_f64"]), tuple(v["lab_f64"])) for v in json.load(_f)["vectors"]]


def as_f64(rgb):  # Rgb(u8).as(f64): @as(f64, c) / 255
    return [c / 255.0 for c in rgb]


def as_u8(rgb):  # Rgb(f64).as(u8): @round(255 * clamp(c, 0, 1)), round half away from zero
    return tuple(int(np.floor(255 * min(max(float(c), 0.0), 1.0) + 0.5)) for c in rgb)


@pytest.mark.parametrize("rgb,hsl,hsv,lab", GOLDEN)
def test_golden_forward_exact_and_round_trip(oracle, rgb, hsl, hsv, lab):
    for space, want in ((oracle.CS_HSL, hsl), (oracle.CS_HSV, hsv), (oracle.CS_LAB, lab)):
        got = oracle.color_to(as_f64(rgb), oracle.CS_RGB, space)
        assert got.tolist() == [float(v) for v in want], f"{rgb} -> space {space}: {got.tolist()} != {want}"  # expectEqualDeep
        back = oracle.color_to(got, space, oracle.CS_RGB)
        assert as_u8(back) == rgb


def test_grey_known_answers(oracle):  # color.zig:1555-1561
    def to_gray_u8(space, vals):
        return int(np.floor(255 * min(max(float(oracle.color_to(vals, space, oracle.CS_GRAY)[0]), 0.0), 1.0) + 0.5))
    assert to_gray_u8(oracle.CS_HSL, (0, 100, 50)) == 54
    assert to_gray_u8(oracle.CS_HSV, (0, 100, 50)) == 27
    assert to_gray_u8(oracle.CS_LAB, (50, 0, 0)) == 119
    # the u8 fixed-point forms (image level)
    img = np.array([[[128, 128, 128], [255, 0, 0]]], np.uint8)
    got = oracle.convert(img, oracle.CS_RGB, oracle.CS_GRAY, np.uint8, 1)
    assert got.tolist() == [[128, 54]]


def test_colorspace_convert_f32(oracle):  # color.zig:1795-1806
    rgb = oracle.color_to((0, 100, 100), oracle.CS_HSV, oracle.CS_RGB, np.float32)
    assert rgb.tolist() == [1.0, 0.0, 0.0]
    hsv = oracle.color_to((1.0, 0.0, 0.0), oracle.CS_RGB, oracle.CS_HSV, np.float32)
    assert hsv.tolist() == [0.0, 100.0, 100.0]


SPACES = ("HSL", "HSV", "XYZ", "LAB", "LCH", "OKLAB", "OKLCH", "XYB", "LMS", "YCBCR")


@pytest.mark.parametrize("name", SPACES)
def test_random_colour_round_trips(oracle, name):  # "100 random colors", color.zig:1738-1773 (2000 here, fixed seed)
    space = getattr(oracle, "CS_" + name)
    rng = np.random.default_rng(2024)
    colours = [tuple(int(v) for v in c) for c in rng.integers(0, 256, (2000, 3))]
    colours += [(0, 0, 0), (255, 255, 255), (255, 0, 0), (0, 255, 0), (0, 0, 255), (1, 1, 1), (254, 255, 253)]
    for rgb in colours:
        there = oracle.color_to(as_f64(rgb), oracle.CS_RGB, space)
        assert as_u8(oracle.color_to(there, space, oracle.CS_RGB)) == rgb, f"{name} {rgb}"


def test_f32_instance_agrees_with_f64(oracle):
    """The f32 instance (the image path) is the same code at another width: it must track the pinned f64 instance to
    f32 precision on every space (a wrong constant or branch would show as a gross error, not as ulps)."""
    rng = np.random.default_rng(7)
    scale = {"HSL": 360, "HSV": 360, "XYZ": 110, "LAB": 130, "LCH": 360, "OKLAB": 1, "OKLCH": 360, "XYB": 1, "LMS": 1.1, "YCBCR": 1}
    for name in SPACES:
        space = getattr(oracle, "CS_" + name)
        for c in rng.integers(0, 256, (300, 3)):
            a = oracle.color_to(as_f64(c), oracle.CS_RGB, space)
            b = oracle.color_to([np.float32(v) / np.float32(255) for v in c], oracle.CS_RGB, space, np.float32)
            d = np.abs(a - b.astype(np.float64))
            if name in ("LCH", "OKLCH", "HSL", "HSV"):  # hue is ill-conditioned near grey; compare as an angle, scaled by chroma
                chroma = a[1] if name in ("LCH", "OKLCH") else a[1] / 100
                d[2 if name in ("LCH", "OKLCH") else 0] = min(d[2 if name in ("LCH", "OKLCH") else 0], 360 - d[2 if name in ("LCH", "OKLCH") else 0]) * min(1.0, chroma)
            assert d.max() <= 2e-5 * scale[name] + 1e-6, f"{name} {c}: f64 {a} f32 {b}"


def test_image_level_routes(oracle):
    """convertColor's scalar / colour / component-type rules at image level (color.zig:108-151)."""
    rgb = np.array([[[255, 136, 0], [12, 200, 99]]], np.uint8)
    lab = oracle.convert(rgb, oracle.CS_RGB, oracle.CS_LAB, np.float32, 3)
    for i in range(2):
        want = oracle.color_to([np.float32(v) / np.float32(255) for v in rgb[0, i]], oracle.CS_RGB, oracle.CS_LAB, np.float32)
        assert lab[0, i].tolist() == want.tolist()
    # Lab(f32) -> Rgb(u8): to(.rgb) in f32, then .as(u8); the u8 round trip of the reference's property test
    back = oracle.convert(lab, oracle.CS_LAB, oracle.CS_RGB, np.uint8, 3)
    assert back.tolist() == rgb.tolist()
    # Lab(f32) -> Rgba(u8): alpha comes out opaque
    back4 = oracle.convert(lab, oracle.CS_LAB, oracle.CS_RGBA, np.uint8, 4)
    assert back4[..., :3].tolist() == rgb.tolist() and (back4[..., 3] == 255).all()
    # Ycbcr(u8) -> Rgb(u8) fixed point round trip within 1 (color.zig:1766-1769)
    ycc = oracle.convert(rgb, oracle.CS_RGB, oracle.CS_YCBCR, np.uint8, 3)
    rt = oracle.convert(ycc, oracle.CS_YCBCR, oracle.CS_RGB, np.uint8, 3)
    assert np.abs(rt.astype(int) - rgb.astype(int)).max() <= 1
    # scalar image -> colour: the scalar is converted to the destination component type first
    g = np.array([[0.5, 0.25]], np.float32)
    hsv = oracle.convert(g, oracle.CS_GRAY, oracle.CS_HSV, np.float32, 3)
    assert hsv[0, 0].tolist() == [0.0, 0.0, 50.0]
    # float-only colour types cannot be u8 images
    with pytest.raises(Exception):
        oracle.convert(rgb, oracle.CS_RGB, oracle.CS_LAB, np.uint8, 3)
