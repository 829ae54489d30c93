"""The host-pointer layer's banded full-duplex pipeline (zg_runtime.cpp: host_banded): row-local ops on host images are cut into
row bands that are uploaded, computed and downloaded concurrently. Whatever the band size, results must equal the oracle —
which also means the whole-frame call — bit for bit: the bands carry real neighbour rows as halo, the frame's true edges get
the border rule, and the interior / border classification of rows is the frame's."""
import os

import numpy as np
import pytest

import zignal_amd as zg
from tests.util import assert_bits_equal, synth

pytestmark = pytest.mark.gpu


@pytest.fixture
def small_bands():
    old = os.environ.get("ZIGNAL_HIP_BAND_MIB")
    os.environ["ZIGNAL_HIP_BAND_MIB"] = "1"  # 1 MiB bands: dozens of bands per frame instead of a handful
    yield
    if old is None:
        os.environ.pop("ZIGNAL_HIP_BAND_MIB", None)
    else:
        os.environ["ZIGNAL_HIP_BAND_MIB"] = old


@pytest.mark.parametrize("kind,rows,cols", [("rgba_u8", 2300, 2048), ("u8", 5000, 4096), ("rgba_f32", 1100, 1024), ("rgb_u8", 3001, 1999)])
def test_banded_convolutions_equal_the_oracle(oracle, small_bands, kind, rows, cols):
    src = synth(oracle, kind, 90, rows, cols)
    img = zg.Image(src)
    for sigma in (0.6, 2.5):  # 5 taps (fused kernels), 17 taps (two-pass kernels): halo 2 and 8 rows
        assert_bits_equal(img.gaussian_blur(sigma).data, oracle.gaussian_blur(src, sigma), f"{kind} gaussianBlur({sigma}) in bands")
    k = np.array([0.05, 0.2, 0.5, 0.2, 0.05], np.float32)
    for border in (zg.BorderMode.zero, zg.BorderMode.replicate, zg.BorderMode.mirror, zg.BorderMode.wrap):  # wrap takes the whole-frame path
        assert_bits_equal(img.convolve_separable(k, k[::-1].copy(), border).data, oracle.conv_separable(src, k, k[::-1].copy(), border), f"{kind} border {border}")
    k2 = np.arange(35, dtype=np.float32).reshape(7, 5) / 600
    assert_bits_equal(img.convolve(k2, zg.BorderMode.mirror).data, oracle.convolve(src, k2, oracle.MIRROR), f"{kind} 7x5 convolve in bands")


def test_banded_views_and_convert(oracle, small_bands):
    rows, cols = 2500, 2048
    src = synth(oracle, "rgba_u8", 91, rows, cols + 9)
    sv = zg.Image(src).view((4, 0, 4 + cols, rows))     # source view: stride != cols
    out = np.zeros((rows, cols + 5, 4), np.uint8)
    ov = zg.Image(out).view((5, 0, 5 + cols, rows))     # destination view
    sv.gaussian_blur(1.0, out=ov)
    assert_bits_equal(out[:, 5:], oracle.gaussian_blur(np.ascontiguousarray(src[:, 4:4 + cols]), 1.0), "views in bands")
    assert not out[:, :5].any()
    plain = np.ascontiguousarray(src[:, :cols])
    assert_bits_equal(zg.Image(plain).convert(zg.CS_OKLAB, np.float32).data, oracle.convert(plain, oracle.CS_RGBA, oracle.CS_OKLAB, np.float32, 3), "convert in bands")
    # in place (source and destination overlap): the pipeline steps aside, the result is still right
    same = plain.copy()
    zg.Image(same).gaussian_blur(0.6, out=zg.Image(same))
    assert_bits_equal(same, oracle.gaussian_blur(plain, 0.6), "in place")
