"""Times Image.convolve on 4096^2 frames (3x3 / 5x5 / 7x7; ZIGNAL_HIP_CONV2D_INT selects the integer accumulators). usage: python tools/exp_conv2d.py [tag]"""
import sys

sys.path.insert(0, ".")
import numpy as np
import torch

import bench
import zignal_amd as zg

tag = sys.argv[1] if len(sys.argv) > 1 else ""
R = 4096
x = torch.rand((4096, 4096), device="cuda")
for _ in range(200):
    x = x * 1.0001
torch.cuda.synchronize()
out = {}
for name, shape in (("rgba", (R, R, 4)), ("grey", (R, R))):
    im = [(zg.Image(torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda")), zg.Image(torch.empty(shape, dtype=torch.uint8, device="cuda"))) for _ in range(4)]
    for n in (3, 5, 7):
        k = np.full((n, n), 1.0 / (n * n), np.float32)
        k[n // 2, n // 2] += 0.25
        k[0, 0] -= 0.25
        out[f"{name}_{n}x{n}"] = bench._time_kernel(torch, lambda i: im[i % 4][0].convolve(k, out=im[i % 4][1]), n=16, warm=4)
print(tag, " ".join(f"{k}={v * 1e3:.1f}us" for k, v in out.items()), flush=True)
