"""Times the long-tap u8 Gaussians ImagePyramid asks for and the pyramid itself. usage: python tools/exp_pyramid.py [tag]"""
import sys

sys.path.insert(0, ".")
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
for name, shape, sigma in (("grey_s1.06", (R, R), 1.06), ("grey_s2.25", (R, R), 2.25), ("grey_s5.5", (R, R), 5.5), ("rgba_s2.5", (R, R, 4), 2.5)):
    im = [(zg.Image(torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda")), zg.Image(torch.empty(shape, dtype=torch.uint8, device="cuda"))) for _ in range(4)]
    out[name] = bench._time_kernel(torch, lambda i: im[i % 4][0].gaussian_blur(sigma, out=im[i % 4][1]), n=24, warm=4)
src = zg.Image(torch.randint(0, 256, (R, R), dtype=torch.uint8, device="cuda"))
out["pyramid"] = bench._time_kernel(torch, lambda i: zg.ImagePyramid.build_default(src), n=6, warm=2)
print(tag, " ".join(f"{k}={v * 1e3:.1f}us" for k, v in out.items()), flush=True)
