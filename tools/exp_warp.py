"""Times the Rgba(f32) radius-2 resamplers with and without the wave-staged kernel (ZIGNAL_HIP_NO_WARP_STAGE). usage: python tools/exp_warp.py [tag]"""
import sys

sys.path.insert(0, ".")
import torch

import bench
import zignal_amd as zg

tag = sys.argv[1] if len(sys.argv) > 1 else ""
I = zg.Interpolation
R = 4096
x = torch.rand((4096, 4096), device="cuda")
for _ in range(200):
    x = x * 1.0001
torch.cuda.synchronize()
src = [zg.Image(torch.rand((R, R, 4), dtype=torch.float32, device="cuda")) for _ in range(3)]
dst = [zg.Image(torch.empty((R, R, 4), dtype=torch.float32, device="cuda")) for _ in range(3)]
half = [zg.Image(torch.empty((R // 2, R // 2, 4), dtype=torch.float32, device="cuda")) for _ in range(3)]
small = [zg.Image(torch.rand((R // 2, R // 2, 4), dtype=torch.float32, device="cuda")) for _ in range(3)]
tr = zg.ProjectiveTransform.from_points([(0, 0), (4095, 0), (0, 4095), (4095, 4095)], [(200, 120), (3900, 60), (90, 3980), (4000, 4050)])
out = {
    "warp_config4": bench._time_kernel(torch, lambda i: src[i % 3].warp(tr, dst[i % 3], I.bicubic), n=12, warm=3),
    "resize_up_2x": bench._time_kernel(torch, lambda i: small[i % 3].resize(dst[i % 3], I.bicubic), n=12, warm=3),
    "resize_down_2x": bench._time_kernel(torch, lambda i: src[i % 3].resize(half[i % 3], I.bicubic), n=12, warm=3),
    "resize_0.8x_catmull": bench._time_kernel(torch, lambda i: src[i % 3].resize(zg.Image(dst[i % 3].data[:3277, :3277]), I.catmull_rom), n=12, warm=3),
    "rotate_10deg": bench._time_kernel(torch, lambda i: src[i % 3].rotate(0.1745, I.bicubic), n=6, warm=2),
}
print(tag, " ".join(f"{k}={v * 1e3:.1f}us" for k, v in out.items()), flush=True)
