"""Times Image(u8).resize(.bilinear) of a 4096^2 plane to a few sizes (the pyramid's first levels and a thumbnail). usage: python tools/exp_resize_u8.py [tag]"""
import sys
sys.path.insert(0, ".")
import torch
import bench
import zignal_amd as zg

tag = sys.argv[1] if len(sys.argv) > 1 else ""
R = 4096
srcs = [zg.Image(torch.randint(0, 256, (R, R), dtype=torch.uint8, device="cuda")) for _ in range(8)]
out = {}
for size in ((3413, 3413), (2844, 2844), (2370, 2370), (2049, 2049), (1024, 1024)):
    dsts = [zg.Image(torch.empty(size, dtype=torch.uint8, device="cuda")) for _ in range(8)]
    out[f"{size[0]}"] = bench._time_kernel(torch, lambda i: srcs[i % 8].resize(dsts[i % 8], zg.Interpolation.bilinear), n=24, warm=4)
print(tag, " ".join(f"{k}={v * 1e3:.1f}us" for k, v in out.items()), flush=True)
