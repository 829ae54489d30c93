#!/usr/bin/env python
"""Regenerates tests/golden/oracle_hashes.json: SHA-256 of the CPU oracle's outputs on seeded inputs, one per operation.

These are REGRESSION pins of the oracle itself (so that a later edit to oracle/ cannot silently change what the GPU is
compared against); they are NOT reference-derived. The reference-derived pins are the known answers transcribed from the
reference's own unit tests (tests/test_oracle_*.py, tests/golden/color_zig_roundtrip_f64.json).

usage: python tests/golden/make_oracle_hashes.py   (from the repository root, after building oracle/)"""
import hashlib
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as o  # noqa: E402


def h(a) -> str:
    a = np.ascontiguousarray(a)
    return hashlib.sha256(str(a.dtype).encode() + str(a.shape).encode() + a.tobytes()).hexdigest()[:32]


def cases():
    u8 = o.synth_u8(11, (61, 93, 4))
    f32 = o.synth_f32(12, (61, 93, 4))
    g8 = o.synth_u8(13, (61, 93))
    gf = o.synth_f32(14, (61, 93))
    bil, bic, lan = o.method(o.BILINEAR), o.method(o.BICUBIC), o.method(o.LANCZOS)
    hm = o.homography_from_4pts([(0, 0), (92, 0), (0, 60), (92, 60)], [(3, 2), (88, 5), (1, 57), (90, 60)])
    yield "gaussian_blur_0.6_rgba_u8", o.gaussian_blur(u8, 0.6)
    yield "gaussian_blur_0.6_rgba_f32", o.gaussian_blur(f32, 0.6)
    yield "gaussian_blur_2.5_u8", o.gaussian_blur(g8, 2.5)
    yield "gaussian_blur_1.0_f32", o.gaussian_blur(gf, 1.0)
    k = o.gaussian_kernel(1.0)
    for b in (o.ZERO, o.REPLICATE, o.MIRROR, o.WRAP):
        yield f"conv_separable_border{b}_rgba_u8", o.conv_separable(u8, k, k, b)
    yield "convolve_3x3_rgba_f32", o.convolve(f32, np.full((3, 3), 1 / 9, np.float32), o.MIRROR)
    yield "box_blur_r2_rgba_u8", o.box_blur(u8, 2)
    yield "resize_bilinear_rgba_u8", o.resize(u8, (23, 41), bil)
    yield "resize_bicubic_rgba_u8", o.resize(u8, (90, 130), bic)
    yield "resize_lanczos_f32", o.resize(gf, (30, 50), lan)
    yield "warp_projective_bicubic_rgba_u8", o.warp(u8, (61, 93), o.PROJECTIVE, hm, bic)
    yield "warp_projective_bilinear_rgba_f32", o.warp(f32, (61, 93), o.PROJECTIVE, hm, bil)
    for name in ("OKLAB", "XYZ", "LAB", "LCH", "OKLCH", "XYB", "HSL", "HSV", "LMS"):
        yield f"convert_rgba_u8_to_{name.lower()}_f32", o.convert(u8, o.CS_RGBA, getattr(o, "CS_" + name), np.float32, 3)
    lab = o.convert(u8, o.CS_RGBA, o.CS_LAB, np.float32, 3)
    yield "convert_lab_f32_to_rgb_u8", o.convert(lab, o.CS_LAB, o.CS_RGB, np.uint8, 3)
    yield "convert_rgba_u8_to_ycbcr_u8", o.convert(u8, o.CS_RGBA, o.CS_YCBCR, np.uint8, 3)
    yield "sobel_rgba_u8", o.sobel(u8)
    yield "canny_u8", o.canny(g8, 1.0, 20, 60)
    yield "threshold_otsu_u8", o.threshold_otsu(g8)[0]
    yield "threshold_adaptive_mean_u8", o.threshold_adaptive_mean(g8, 3, 5.0)
    yield "morph_open_u8", o.morph((g8 > 128).astype(np.uint8) * 255, np.ones((3, 3), np.uint8), 2, o.MORPH_OPEN)
    yield "sharpen_rgba_u8", o.sharpen(u8, 2)
    yield "median_blur_r2_rgba_u8", o.order_statistic_blur(u8, 2, o.OS_PERCENTILE, 0.5, o.MIRROR)
    yield "alpha_trimmed_r1_u8", o.order_statistic_blur(g8, 1, o.OS_ALPHA_TRIMMED, 0.2, o.REPLICATE)
    yield "autocontrast_rgba_u8", o.autocontrast(u8.copy(), 0.02)
    yield "equalize_u8", o.equalize(g8.copy())
    yield "shen_castan_u8", o.shen_castan(g8, 0.8, 7, 0.9, 0.3)
    yield "shen_castan_nms_rgba_u8", o.shen_castan(u8, 0.85, 5, 0.8, 0.4, True, True)
    yield "motion_blur_linear_rgba_u8", o.motion_blur_linear(u8, 0.3, 8)
    yield "motion_blur_spin_f32", o.motion_blur_radial(gf, 0.4, 0.6, 0.5, True)
    # codecs: files built by the test-side writers from seeded samples (the pixels, not the compressed bytes, are hashed)
    from tests import jpeg_util as J
    from tests import png_util as P
    rng = np.random.default_rng(77)
    pal = rng.integers(0, 256, (16, 3)).tolist()
    yield "png_rgba16_adam7", o.png_decode_native(P.make_png(rng.integers(0, 65536, (19, 23, 4)), 16, P.RGBA, 1, filters=4))[0]
    yield "png_palette4_trns", o.png_decode_native(P.make_png(rng.integers(0, 16, (19, 23, 1)), 4, P.PALETTE, 0, filters=3, palette=pal, trns=[9, 8, 7]))[0]
    yield "png_gray2_trns_adam7", o.png_decode_native(P.make_png(rng.integers(0, 4, (9, 31, 1)), 2, P.GRAY, 1, filters=1, trns=[0, 85]))[0]
    yield "png_filter_adaptive_rgb", o.png_filter(o.synth_u8(15, (40, 33, 3)))
    yield "png_filter_adaptive_tall_u8", o.png_filter(np.repeat(o.synth_u8(16, (75, 21)), 8, axis=0))
    for name, (lh, lv) in (("444", (1, 1)), ("422", (2, 1)), ("420", (2, 2)), ("411", (4, 1))):
        comps = J.layout(lh, lv)
        yield f"jpeg_baseline_{name}", o.jpeg_decode_native(J.write_baseline(45, 37, comps, J.FLAT_Q, J.random_coefficients(rng, comps, 45, 37)))[0]
    sig, eoi = bytes([0xFF, 0xD8]), bytes([0xFF, 0xD9])
    prog = (sig + bytes([0xFF, 0xDB, 0x00, 0x43, 0x00]) + bytes([8] * 64) + bytes([0xFF, 0xC2, 0x00, 0x0B, 0x08, 0x00, 0x08, 0x00, 0x08, 0x01, 0x01, 0x11, 0x00])
            + bytes([0xFF, 0xC4, 0x00, 0x14, 0x00, 0x01]) + bytes(15) + bytes([0x02]) + bytes([0xFF, 0xDA, 0x00, 0x08, 0x01, 0x01, 0x00, 0x00, 0x00, 0x02, 0x7F])
            + bytes([0xFF, 0xDA, 0x00, 0x08, 0x01, 0x01, 0x00, 0x00, 0x00, 0x21, 0xFF, 0x00]) + eoi)
    yield "jpeg_progressive_hand_built", o.jpeg_decode_native(prog)[0]
    for name, sub in (("444", 0), ("422", 1), ("420", 2)):
        yield f"jpeg_encode_{name}_q70", np.frombuffer(o.jpeg_encode(o.synth_u8(17, (37, 53, 3)), 70, sub), np.uint8)
    yield "jpeg_encode_grey_q35", np.frombuffer(o.jpeg_encode(o.synth_u8(18, (37, 53)), 35), np.uint8)
    yield "jpeg_fdct_block", o.jpeg_fdct8x8((o.synth_u8(19, (8, 8)).astype(np.int32) - 128))
    blk = np.zeros((8, 8), np.int32)
    blk.flat[[0, 1, 8, 9, 17, 34, 63]] = [900, -310, 255, 77, -41, 19, -7]
    yield "jpeg_idct_block", o.jpeg_idct8x8(blk)


if __name__ == "__main__":
    out = {name: h(arr) for name, arr in cases()}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_hashes.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(f"wrote {len(out)} hashes to {path}")
