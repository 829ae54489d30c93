"""Times Image.sobel on 4096^2 frames (ZIGNAL_HIP_NO_SOBEL_STREAM selects the LDS-tiled kernel). usage: python tools/exp_sobel.py [tag]"""
import sys

sys.path.insert(0, ".")
import torch

import bench
import zignal_amd as zg

tag = sys.argv[1] if len(sys.argv) > 1 else ""
R = 4096
x = torch.rand((4096, 4096), device="cuda")
for _ in range(300):
    x = x * 1.0001
torch.cuda.synchronize()
out = {}
for name, shape in (("rgba", (R, R, 4)), ("grey", (R, R))):
    im = [(zg.Image(torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda")), zg.Image(torch.empty((R, R), dtype=torch.uint8, device="cuda"))) for _ in range(4)]
    out[name] = bench._time_kernel(torch, lambda i: im[i % 4][0].sobel(out=im[i % 4][1]), n=16, warm=4)
print(tag, " ".join(f"sobel_{k}={v * 1e3:.1f}us" for k, v in out.items()), flush=True)
