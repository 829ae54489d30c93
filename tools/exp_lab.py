"""Times Rgba(u8) -> Lab(f32) and Lab(f32) -> Rgba(u8) on 4096^2 frames (ZIGNAL_HIP_NO_LAB4 selects the route walker). usage: python tools/exp_lab.py [tag]"""
import sys

sys.path.insert(0, ".")
import numpy as np
import torch

import bench
import zignal_amd as zg

tag = sys.argv[1] if len(sys.argv) > 1 else ""
R = 4096
x = torch.rand((4096, 4096), device="cuda")
for _ in range(300):
    x = x * 1.0001
torch.cuda.synchronize()
src = [zg.Image(torch.randint(0, 256, (R, R, 4), dtype=torch.uint8, device="cuda")) for _ in range(3)]
lab = [zg.Image(torch.empty((R, R, 3), dtype=torch.float32, device="cuda")) for _ in range(3)]
back = [zg.Image(torch.empty((R, R, 4), dtype=torch.uint8, device="cuda")) for _ in range(3)]
out = {"rgba_u8_to_lab": bench._time_kernel(torch, lambda i: src[i % 3].convert(zg.CS_LAB, np.float32, out=lab[i % 3]), n=12, warm=3)}
out["lab_to_rgba_u8"] = bench._time_kernel(torch, lambda i: lab[i % 3].convert(zg.CS_RGBA, np.uint8, src_space=zg.CS_LAB, out=back[i % 3]), n=12, warm=3)
print(tag, " ".join(f"{k}={v * 1e3:.1f}us" for k, v in out.items()), flush=True)
