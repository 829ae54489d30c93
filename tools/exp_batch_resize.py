"""Bilinear 4096^2 -> 1024^2 Rgba(u8) resize, one frame per launch against n frames per launch (zg_batch_pipeline). usage: [n ...]"""
import sys
sys.path.insert(0, ".")
import torch
import bench
import zignal_amd as zg

ns = [int(a) for a in sys.argv[1:]] or [1, 4, 16]
x = torch.rand((4096, 4096), device="cuda")
for _ in range(200):
    x = x * 1.0001
torch.cuda.synchronize()
for n in ns:
    ring = max(2, 32 // n)  # >= 2 GiB of distinct sources
    srcs = [torch.randint(0, 256, (n, 4096, 4096, 4), dtype=torch.uint8, device="cuda") for _ in range(ring)]
    outs = [torch.empty((n, 1024, 1024, 4), dtype=torch.uint8, device="cuda") for _ in range(ring)]
    for label, steps in (("resize", [zg.Step.resize(1024, 1024)]), ("resize+oklab", [zg.Step.resize(1024, 1024), zg.Step.convert(zg.CS_OKLAB)])):
        p = zg.Pipeline(steps)
        o = outs if label == "resize" else [torch.empty((n, 1024, 1024, 3), dtype=torch.float32, device="cuda") for _ in range(ring)]
        ms = bench._time_kernel(torch, lambda i: p.run(srcs[i % ring], out=o[i % ring]), n=max(4, 64 // n), warm=2)
        per = ms / n * 1e3
        alg = (20 if label == "resize" else 28) * 1024 * 1024
        dram = 4096 * 4096 * 4 // 2 + (4 if label == "resize" else 12) * 1024 * 1024
        print(f"{label:13s} n={n:2d}: {per:6.2f} us per frame  strict {alg / per / 1e3 / 8000:.3f}  on DRAM-granular bytes {dram / per / 1e3 / 8000:.3f}", flush=True)
