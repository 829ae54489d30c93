"""Secondary layouts / paths: ms per 4096^2 frame (graph-timed)."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import zignal_amd as zg
from bench import _time_kernel
R = 4096
def frames(shape, dtype):
    if dtype == torch.uint8:
        return [torch.randint(0, 256, shape, dtype=dtype, device="cuda") for _ in range(4)]
    return [torch.rand(shape, dtype=dtype, device="cuda") for _ in range(4)]
def run(name, shape, dtype, fn, bytes_px):
    src = frames(shape, dtype)
    im = [(zg.Image(s), zg.Image(torch.empty_like(s))) for s in src]
    ms = _time_kernel(torch, lambda i: fn(im[i % 4][0], im[i % 4][1]), n=20, warm=3)
    print(f"{name:44s} {ms*1e3:9.1f} us  {bytes_px*R*R/ms/1e6:8.1f} GB/s")
blur = lambda s: (lambda a, b: a.gaussian_blur(s, out=b))
run("blur 0.6 u8 plane", (R, R), torch.uint8, blur(0.6), 2)
run("blur 0.6 rgb_u8", (R, R, 3), torch.uint8, blur(0.6), 6)
run("blur 0.6 rgb_f32", (R, R, 3), torch.float32, blur(0.6), 24)
run("blur 0.6 rgba_u8", (R, R, 4), torch.uint8, blur(0.6), 8)
run("blur 1.0 rgba_u8 (7 taps)", (R, R, 4), torch.uint8, blur(1.0), 8)
run("blur 1.5 rgba_u8 (11 taps)", (R, R, 4), torch.uint8, blur(1.5), 8)
run("blur 2.5 rgba_u8 (17 taps)", (R, R, 4), torch.uint8, blur(2.5), 8)
run("blur 2.5 u8 plane (17 taps)", (R, R), torch.uint8, blur(2.5), 2)
run("blur 5.5 u8 plane (35 taps, ORB level 7)", (R, R), torch.uint8, blur(5.5), 2)
run("blur 2.5 f32 plane (17 taps)", (R, R), torch.float32, blur(2.5), 8)
run("blur 2.5 rgba_f32 (17 taps)", (R, R, 4), torch.float32, blur(2.5), 32)
run("blur 1.0 rgba_f32 (7 taps)", (R, R, 4), torch.float32, blur(1.0), 32)
k3 = np.full((3, 3), 1 / 9, np.float32)
run("convolve 3x3 rgba_u8", (R, R, 4), torch.uint8, lambda a, b: a.convolve(k3, 1, out=b), 8)
run("convolve 3x3 rgba_f32", (R, R, 4), torch.float32, lambda a, b: a.convolve(k3, 1, out=b), 32)
run("box_blur r=2 rgba_u8", (R, R, 4), torch.uint8, lambda a, b: a.box_blur(2, out=b), 8)
run("box_blur r=2 u8", (R, R), torch.uint8, lambda a, b: a.box_blur(2, out=b), 2)
torch.cuda.synchronize()
x = torch.randint(0, 256, (R, R, 4), dtype=torch.uint8, device="cuda"); y = torch.empty_like(x)
ms = _time_kernel(torch, lambda i: y.copy_(x), n=20, warm=3); print(f"{'torch copy 64 MB':44s} {ms*1e3:9.1f} us  {2*x.numel()/ms/1e6:8.1f} GB/s")
x = torch.rand((R, R, 4), device="cuda"); y = torch.empty_like(x)
ms = _time_kernel(torch, lambda i: y.copy_(x), n=20, warm=3); print(f"{'torch copy 256 MB':44s} {ms*1e3:9.1f} us  {2*x.numel()*4/ms/1e6:8.1f} GB/s")
