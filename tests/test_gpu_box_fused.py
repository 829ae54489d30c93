"""boxBlur / sharpen of Image(u8) and Image(Rgba(u8)) through box_fused.hip (radius 1..3, at least 64 x 64): the shapes its geometry has edges at.
Strips are 16 output columns wide; a row's carries come in pieces of 16 columns (the last one may reach past the row: cols % 16 != 0 with four channels);
rows come in blocks of 64 (four channels) or 128 (one channel); one-channel rows that do not end on a dword take a loader and a store path of their own.
Reference: src/image.zig:635-648, 785-801; src/image/integral.zig:41-78, 194-269, 273-426. Bit-exact against the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def env():
    import torch

    import zignal_amd as zg
    from oracle import pyoracle as oracle

    return torch, zg, oracle


def _dev(torch, zg, a):
    return zg.Image(torch.from_numpy(a).cuda())


def _get(torch, img):
    torch.cuda.synchronize()
    return img.to_numpy()


@pytest.mark.parametrize("kind", ["u8", "rgba_u8"])
def test_strip_and_piece_edges(env, kind):
    """widths around multiples of 16 (whole / partial last strip, a carry piece that ends on, before and past the row's end) and of 4 (one channel's dword rule)"""
    torch, zg, oracle = env
    tail = (4,) if kind == "rgba_u8" else ()
    for cols in (64, 65, 66, 67, 79, 80, 81, 95, 97, 113, 127, 128, 129, 255, 1031):
        src = oracle.synth_u8(cols, (96, cols) + tail)
        for radius in (1, 2, 3):
            got = _get(torch, _dev(torch, zg, src).box_blur(radius))
            assert np.array_equal(got, oracle.box_blur(src, radius)), f"boxBlur {kind} 96x{cols} r={radius}"
        assert np.array_equal(_get(torch, _dev(torch, zg, src).sharpen(2)), oracle.sharpen(src, 2)), f"sharpen {kind} 96x{cols}"


@pytest.mark.parametrize("kind", ["u8", "rgba_u8"])
def test_block_edges(env, kind):
    """heights around the block sizes: one block and a bit, clipped bottom rows in the last / the last but one block, a single partial block"""
    torch, zg, oracle = env
    tail = (4,) if kind == "rgba_u8" else ()
    for rows in (64, 65, 66, 67, 68, 71, 126, 127, 128, 129, 130, 131, 191, 192, 193, 255, 256, 257, 259, 385, 515):
        src = oracle.synth_u8(rows * 3, (rows, 80) + tail)
        for radius in (1, 3):
            got = _get(torch, _dev(torch, zg, src).box_blur(radius))
            assert np.array_equal(got, oracle.box_blur(src, radius)), f"boxBlur {kind} {rows}x80 r={radius}"
        assert np.array_equal(_get(torch, _dev(torch, zg, src).sharpen(3)), oracle.sharpen(src, 3)), f"sharpen {kind} {rows}x80"


@pytest.mark.parametrize("kind", ["u8", "rgba_u8"])
def test_extreme_values_and_views(env, kind):
    """all white (the largest SAT values: inexact from the first rows on), all black; source and destination as views with strides that are not the row length;
    in place (the library copies the source aside)"""
    torch, zg, oracle = env
    tail = (4,) if kind == "rgba_u8" else ()
    for fill in (255, 0):
        src = np.full((1100, 1300) + tail, fill, np.uint8)
        assert np.array_equal(_get(torch, _dev(torch, zg, src).box_blur(2)), oracle.box_blur(src, 2)), f"boxBlur {kind} fill {fill}"
    big = oracle.synth_u8(31, (300, 420) + tail)
    canvas = torch.full((320, 460) + tail, 0x5A, dtype=torch.uint8, device="cuda")
    rect = (12, 5, 12 + 200, 5 + 270)  # l, t, r, b
    _dev(torch, zg, big).view((4, 2, 204, 272)).box_blur(3, out=zg.Image(canvas).view(rect))
    got = _get(torch, zg.Image(canvas))
    assert np.array_equal(got[5:275, 12:212], oracle.box_blur(np.ascontiguousarray(big[2:272, 4:204]), 3)), f"views {kind}"
    got[5:275, 12:212] = 0x5A
    assert np.all(got == 0x5A), f"views {kind}: wrote outside the destination view"
    src = oracle.synth_u8(77, (333, 203) + tail)
    t = _dev(torch, zg, src)
    t.sharpen(2, out=t)
    assert np.array_equal(_get(torch, t), oracle.sharpen(src, 2)), f"sharpen in place {kind}"


def test_frames_in_the_grid(env):
    """the batched step (zg_batch_pipeline): frames are the grid's second dimension, each with its own carries"""
    torch, zg, oracle = env
    for ch in (1, 4):
        host = np.stack([oracle.synth_u8(100 + f, (150, 210) + ((4,) if ch == 4 else ())) for f in range(5)])
        got = zg.Pipeline([zg.Step.box_blur(2)]).run(torch.from_numpy(host).cuda())
        torch.cuda.synchronize()
        got = got.cpu().numpy()
        for f in range(5):
            assert np.array_equal(got[f], oracle.box_blur(host[f], 2)), f"frame {f}, {ch} channel(s)"


def test_agrees_with_the_integral_image_route(env):
    """ZIGNAL_HIP_BOX_UNFUSED=1 (read once per process) selects round 5's three kernels: same bytes, so the A/B hook compares like with like"""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np, torch, zignal_amd as zg\n"
        "from oracle import pyoracle as oracle\n"
        "ok = True\n"
        "for tail in ((), (4,)):\n"
        "    src = oracle.synth_u8(9, (700, 900) + tail)\n"
        "    d = zg.Image(torch.from_numpy(src).cuda())\n"
        "    for r in (1, 2, 3):\n"
        "        g = d.box_blur(r); torch.cuda.synchronize(); ok &= bool(np.array_equal(g.to_numpy(), oracle.box_blur(src, r)))\n"
        "print('OK' if ok else 'FAIL')\n" % ROOT
    )
    out = subprocess.run([sys.executable, "-c", code], env={**os.environ, "ZIGNAL_HIP_BOX_UNFUSED": "1"}, capture_output=True, text=True, timeout=300)
    assert out.stdout.strip().endswith("OK"), out.stdout + out.stderr


def test_the_widest_and_the_tallest(env):
    """65 536 columns (4096 strips; the widest row whose sums stay exact), one column more (the integral-image route), 16 385 Rgba pixels (a last strip of one
    pixel, a carry piece that reaches past the row), a plane 70 000 rows tall (1094 / 547 steps)"""
    torch, zg, oracle = env
    for shape in ((64, 65536), (100, 65537), (66, 16385, 4), (70000, 64), (40000, 70, 4)):
        src = oracle.synth_u8(11, shape)
        got = _get(torch, _dev(torch, zg, src).box_blur(2))
        assert np.array_equal(got, oracle.box_blur(src, 2)), f"boxBlur {shape}"
