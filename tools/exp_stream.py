"""Times the u8 Gaussian kernels under the tuning hooks of conv_sep_stream.hip (one process per setting: the hooks are read once).
usage: python tools/exp_stream.py [tag]"""
import ctypes as C
import os
import sys

sys.path.insert(0, ".")
import numpy as np
import torch

import bench
import zignal_amd as zg

tag = sys.argv[1] if len(sys.argv) > 1 else ""
R = 4096
I = zg.Interpolation


def frames(n, shape):
    return [torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda") for _ in range(n)]


def blur(shape, sigma=0.6, ring=8):
    im = [(zg.Image(s), zg.Image(torch.empty_like(s))) for s in frames(ring, shape)]
    return bench._time_kernel(torch, lambda i: im[i % ring][0].gaussian_blur(sigma, out=im[i % ring][1]), n=48, warm=8)


def batch(n=128):
    rows, cols = 1080, 1920
    src = torch.randint(0, 256, (n, rows, cols, 4), dtype=torch.uint8, device="cuda")
    dst = torch.empty((n, 540, 960, 4), dtype=torch.uint8, device="cuda")
    m = I.bilinear._c()
    lib = zg.lib()

    def run(_):
        rc = lib.zg_batch_blur_resize(C.c_void_p(src.data_ptr()), n, rows, cols, 3, C.c_float(0.6), C.c_void_p(dst.data_ptr()), 540, 960, C.byref(m),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, lib.zg_last_error()
    return bench._time_kernel(torch, run, n=10, warm=2)


# warm the clocks
x = torch.rand((4096, 4096), device="cuda")
for _ in range(200):
    x = x * 1.0001
torch.cuda.synchronize()
out = {"rgba": blur((R, R, 4)), "grey": blur((R, R)), "rgb": blur((R, R, 3)), "rgba_s1.0(7tap)": blur((R, R, 4), 1.0), "batch128": batch(128)}
print(tag, " ".join(f"{k}={v * 1e3:.1f}us" for k, v in out.items()), flush=True)
