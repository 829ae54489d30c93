"""Parity of the MFMA u8 Gaussian (ZIGNAL_HIP_MFMA=1, set by the caller) against the oracle over shapes, pixel types, borders and tap
counts, with a description of where a mismatch sits. usage: ZIGNAL_HIP_MFMA=1 python tools/exp_mfma.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch

import zignal_amd as zg
from oracle import pyoracle as o

o.lib()
rng = np.random.default_rng(11)
bad = 0


def check(shape, sigma, border=None, what=""):
    global bad
    a = rng.integers(0, 256, shape, dtype=np.uint8)
    dev = zg.Image(torch.from_numpy(a).cuda())
    if border is None:
        got = dev.gaussian_blur(sigma).to_numpy()
        want = o.gaussian_blur(a, sigma)
    else:
        k = np.asarray(zg.gaussian_kernel(sigma), dtype=np.float32)
        got = dev.convolve_separable(k, k, border).to_numpy()
        want = o.conv_separable(a, k, k, border)
    torch.cuda.synchronize()
    if np.array_equal(got, want):
        print("ok  ", shape, sigma, border, what, flush=True)
        return
    bad += 1
    d = got.astype(int) - want.astype(int)
    nz = np.argwhere(d.reshape(shape[0], -1) != 0)
    rows, cols = np.unique(nz[:, 0]), np.unique(nz[:, 1])
    print("BAD ", shape, sigma, border, what, f"{len(nz)} of {d.size} bytes differ; rows {rows[:6]}..{rows[-3:]} ({len(rows)}), byte cols {cols[:8]}..{cols[-4:]} ({len(cols)}), "
          f"max |d| {np.abs(d).max()}, first: got {got.reshape(shape[0], -1)[nz[0][0], nz[0][1]]} want {want.reshape(shape[0], -1)[nz[0][0], nz[0][1]]}", flush=True)


for shape in ((64, 64, 4), (97, 128, 4), (200, 1100, 4), (33, 256), (130, 1296), (70, 320, 3), (1080, 1920, 4)):
    for sigma in (0.6, 1.0, 0.4):
        check(shape, sigma)
B = zg.BorderMode
if True:
    for border in (B.zero, B.replicate, B.mirror):
        check((100, 400, 4), 0.6, border)
        check((100, 400, 4), 1.0, border)
        check((50, 512), 1.3, border)
print("mismatching cases:", bad)
