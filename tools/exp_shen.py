"""Times Image.shenCastan on a 4096^2 Rgba(u8) frame (bench.py's leg) and the ISEF smoothing inside it through rocprofv3-free event timing of the whole
call. usage: python tools/exp_shen.py [tag]"""
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
src = [zg.Image(torch.randint(0, 256, (R, R, 4), dtype=torch.uint8, device="cuda")) for _ in range(2)]
dst = [zg.Image(torch.empty((R, R), dtype=torch.uint8, device="cuda")) for _ in range(2)]
t = bench._time_kernel(torch, lambda i: src[i % 2].shen_castan(out=dst[i % 2]), n=8, warm=2)
print(tag, f"shen_castan_rgba_u8_4096={t * 1e3:.1f}us", flush=True)
