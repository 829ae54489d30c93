import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import zignal_amd as zg
a = np.random.default_rng(1).random((4096, 4096, 4), dtype=np.float32)
out = np.empty_like(a)
img, o = zg.Image(a), zg.Image(out)
img.gaussian_blur(0.6, out=o)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); img.gaussian_blur(0.6, out=o); ts.append(time.perf_counter() - t0)
best = min(ts)
print(f"host-pointer gaussianBlur 4096^2 rgba f32 (pageable numpy): {best*1e3:.2f} ms  = {16.777216/best:.1f} Mpixels/s, {2*a.nbytes/best/1e9:.1f} GB/s over PCIe incl. alloc")
u = (a[..., :] * 255).astype(np.uint8); ou = np.empty_like(u)
iu, oo = zg.Image(u), zg.Image(ou)
iu.gaussian_blur(0.6, out=oo)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); iu.gaussian_blur(0.6, out=oo); ts.append(time.perf_counter() - t0)
best = min(ts)
print(f"host-pointer gaussianBlur 4096^2 rgba u8: {best*1e3:.2f} ms = {16.777216/best:.1f} Mpixels/s")
