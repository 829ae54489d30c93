"""Times the two kernels VERDICT r03 found regressed (Rgba(u8) bicubic warp of config 4, the 8192^2 -> 4096^2 bilinear resize) and the batched
forms that must not move (16 frames per launch). usage: python tools/exp_regress.py [tag]"""
import sys

sys.path.insert(0, ".")
import torch

import bench
import zignal_amd as zg

tag = sys.argv[1] if len(sys.argv) > 1 else ""
I = zg.Interpolation
R = 4096
x = torch.rand((4096, 4096), device="cuda")
for _ in range(300):
    x = x * 1.0001
torch.cuda.synchronize()
src = [zg.Image(torch.randint(0, 256, (R, R, 4), dtype=torch.uint8, device="cuda")) for _ in range(4)]
dst = [zg.Image(torch.empty((R, R, 4), dtype=torch.uint8, device="cuda")) for _ in range(4)]
big = [zg.Image(torch.randint(0, 256, (2 * R, 2 * R, 4), dtype=torch.uint8, device="cuda")) for _ in range(4)]
tr = zg.ProjectiveTransform.from_points([(0, 0), (4095, 0), (0, 4095), (4095, 4095)], [(200, 120), (3900, 60), (90, 3980), (4000, 4050)])
out = {
    "warp_u8_config4": bench._time_kernel(torch, lambda i: src[i % 4].warp(tr, dst[i % 4], I.bicubic), n=16, warm=4),
    "resize_8192_to_4096": bench._time_kernel(torch, lambda i: big[i % 4].resize(dst[i % 4], I.bilinear), n=16, warm=4),
}
del big
n = 16
srcs = [torch.randint(0, 256, (n, R, R, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
outs = [torch.empty((n, 1024, 1024, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
p = zg.Pipeline([zg.Step.resize(1024, 1024)])
out["resize_4096_to_1024_x16_per_frame"] = bench._time_kernel(torch, lambda i: p.run(srcs[i % 2], out=outs[i % 2]), n=6, warm=2) / n
print(tag, " ".join(f"{k}={v * 1e3:.2f}us" for k, v in out.items()), flush=True)
