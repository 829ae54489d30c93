"""Run one op a few times on the GPU (for rocprofv3 passes). usage: python tools/run_op.py blur_u8|blur_f32|resize|warp_u8|warp_f32|oklab|box_u8|box_rgba8|blur17_u8|blur17_f32|blur11_u8|conv3_u8|conv3_f32 [n]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import zignal_amd as zg

op = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
I = zg.Interpolation
R = 4096
if op == "blur_u8":
    s = zg.Image(torch.randint(0, 256, (R, R, 4), dtype=torch.uint8, device="cuda")); d = zg.Image(torch.empty_like(s.data))
    f = lambda: s.gaussian_blur(0.6, out=d)
elif op.startswith("greyblur_"):  # greyblur_2.25: gaussianBlur(sigma) of a 4096^2 Image(u8), the pyramid's long-tap two-pass path
    s = zg.Image(torch.randint(0, 256, (R, R), dtype=torch.uint8, device="cuda")); d = zg.Image(torch.empty_like(s.data))
    sg = float(op.split("_")[1])
    f = lambda: s.gaussian_blur(sg, out=d)
elif op in ("blur_f32plane", "blur_f32planes4"):  # Image(f32) planes: k_sep_tile_f32, one launch for one / four planes; eight sets so the sources pass the Infinity Cache
    n_pl = 4 if op.endswith("4") else 1
    sets = [([zg.Image(torch.rand((R, R), dtype=torch.float32, device="cuda")) for _ in range(n_pl)], [zg.Image(torch.empty((R, R), dtype=torch.float32, device="cuda")) for _ in range(n_pl)]) for _ in range(8 // n_pl * 2)]
    it = [0]
    def f():
        it[0] += 1
        a, b = sets[it[0] % len(sets)]
        zg.gaussian_blur_planes(a, 0.6, outs=b)
elif op == "resize_lab":  # the fused resize -> Oklab of one frame (config 3's pair of steps), eight sources
    ss = [zg.Image(torch.randint(0, 256, (R, R, 4), dtype=torch.uint8, device="cuda")) for _ in range(8)]; d = zg.Image(torch.empty((1024, 1024, 3), dtype=torch.float32, device="cuda"))
    it = [0]
    def f():
        it[0] += 1
        ss[it[0] % 8].resize_convert(d, zg.CS_OKLAB)
elif op == "box_small":  # boxBlur of a 256 x 256 Image(u8): rows * cols * 255 < 2^24, the direct kernel (no integral image)
    s = zg.Image(torch.randint(0, 256, (256, 256), dtype=torch.uint8, device="cuda")); d = zg.Image(torch.empty_like(s.data))
    f = lambda: s.box_blur(2, out=d)
elif op == "blur_f32":
    s = zg.Image(torch.rand((R, R, 4), dtype=torch.float32, device="cuda")); d = zg.Image(torch.empty_like(s.data))
    f = lambda: s.gaussian_blur(0.6, out=d)
elif op == "resize":  # eight distinct sources (512 MiB): past the 256 MiB Infinity Cache
    ss = [zg.Image(torch.randint(0, 256, (R, R, 4), dtype=torch.uint8, device="cuda")) for _ in range(8)]; d = zg.Image(torch.empty((1024, 1024, 4), dtype=torch.uint8, device="cuda"))
    it = [0]
    def f():
        it[0] += 1
        ss[it[0] % 8].resize(d, I.bilinear)
elif op in ("warp_u8", "warp_f32"):
    tr = zg.ProjectiveTransform.from_points([(0, 0), (4095, 0), (0, 4095), (4095, 4095)], [(200, 120), (3900, 60), (90, 3980), (4000, 4050)])
    t = torch.randint(0, 256, (R, R, 4), dtype=torch.uint8, device="cuda") if op == "warp_u8" else torch.rand((R, R, 4), dtype=torch.float32, device="cuda")
    s = zg.Image(t); d = zg.Image(torch.empty_like(t))
    f = lambda: s.warp(tr, d, I.bicubic)
elif op == "oklab":
    s = zg.Image(torch.randint(0, 256, (R, R, 4), dtype=torch.uint8, device="cuda")); d = zg.Image(torch.empty((R, R, 3), dtype=torch.float32, device="cuda"))
    f = lambda: s.convert(zg.CS_OKLAB, np.float32, out=d)
elif op in ("box_u8", "box_rgba8"):
    t = torch.randint(0, 256, (R, R, 4) if op == "box_rgba8" else (R, R), dtype=torch.uint8, device="cuda")
    s = zg.Image(t); d = zg.Image(torch.empty_like(t))
    f = lambda: s.box_blur(2, out=d)
elif op in ("blur17_u8", "blur17_f32", "blur11_u8"):
    t = torch.rand((R, R, 4), dtype=torch.float32, device="cuda") if op.endswith("f32") else torch.randint(0, 256, (R, R, 4), dtype=torch.uint8, device="cuda")
    s = zg.Image(t); d = zg.Image(torch.empty_like(t))
    f = lambda: s.gaussian_blur(1.5 if "11" in op else 2.5, out=d)
elif op in ("blur17_g8", "blur35_g8", "blur11_g8"):
    t = torch.randint(0, 256, (R, R), dtype=torch.uint8, device="cuda")
    s = zg.Image(t); d = zg.Image(torch.empty_like(t))
    f = lambda: s.gaussian_blur({"17": 2.5, "35": 5.5, "11": 1.5}[op[4:6]], out=d)
elif op in ("shen_photo", "canny_photo"):  # photo-like frame: smooth colour fields, a few hundred hard-edged shapes, a little sensor noise
    g = torch.Generator(device="cuda").manual_seed(7)
    yy, xx = torch.meshgrid(torch.arange(R, device="cuda"), torch.arange(R, device="cuda"), indexing="ij")
    pic = torch.stack([128 + 90 * torch.sin(xx / 310.0) * torch.cos(yy / 270.0), 128 + 80 * torch.cos(xx / 190.0 + yy / 400.0), 128 + 100 * torch.sin((xx + yy) / 520.0)], -1)
    for _ in range(300):
        cx, cy, rad = [int(v) for v in torch.randint(0, R, (3,), generator=g, device="cuda").tolist()]
        rad = 20 + rad % 180
        m = ((xx - cx) ** 2 + (yy - cy) ** 2) < rad * rad
        pic[m] = pic[m] * 0.5 + torch.randint(0, 256, (3,), generator=g, device="cuda").float() * 0.5
    pic = (pic + 2.0 * torch.randn(pic.shape, generator=g, device="cuda")).clamp(0, 255)
    t = torch.cat([pic, torch.full((R, R, 1), 255.0, device="cuda")], -1).to(torch.uint8).contiguous()
    s = zg.Image(t); d = zg.Image(torch.empty((R, R), dtype=torch.uint8, device="cuda"))
    f = (lambda: s.shen_castan(out=d)) if op == "shen_photo" else (lambda: s.canny(1.4, 50, 150, out=d))
elif op == "batch64":  # BASELINE configs[4]: 64 x 1080p Rgba(u8) frames, gaussianBlur(0.6) then resize(.bilinear) to 540 x 960
    import ctypes as C
    src = torch.randint(0, 256, (64, 1080, 1920, 4), dtype=torch.uint8, device="cuda")
    dst = torch.empty((64, 540, 960, 4), dtype=torch.uint8, device="cuda")
    m = I.bilinear._c()
    lib = zg.lib()
    def f():
        rc = lib.zg_batch_blur_resize(C.c_void_p(src.data_ptr()), 64, 1080, 1920, 3, C.c_float(0.6), C.c_void_p(dst.data_ptr()), 540, 960, C.byref(m),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, lib.zg_last_error()
elif op == "pyramid":  # ImagePyramid.build(source, 8, 1.2, 1.6) on a grey frame (ORB's default)
    src = zg.Image(torch.randint(0, 256, (R, R), dtype=torch.uint8, device="cuda"))
    f = lambda: zg.ImagePyramid.build_default(src)
elif op in ("shen", "canny", "sobel"):
    t = torch.randint(0, 256, (R, R, 4), dtype=torch.uint8, device="cuda")
    s = zg.Image(t); d = zg.Image(torch.empty((R, R), dtype=torch.uint8, device="cuda"))
    f = {"shen": lambda: s.shen_castan(out=d), "canny": lambda: s.canny(1.4, 50, 150, out=d), "sobel": lambda: s.sobel(out=d)}[op]
elif op in ("lab", "lab_back"):
    t = torch.randint(0, 256, (R, R, 4), dtype=torch.uint8, device="cuda")
    s = zg.Image(t); d = zg.Image(torch.empty((R, R, 3), dtype=torch.float32, device="cuda"))
    s.convert(zg.CS_LAB, np.float32, out=d)
    b = zg.Image(torch.empty_like(t))
    f = (lambda: s.convert(zg.CS_LAB, np.float32, out=d)) if op == "lab" else (lambda: d.convert(zg.CS_RGBA, np.uint8, src_space=zg.CS_LAB, out=b))
elif op == "resize16":  # sixteen 4096^2 -> 1024^2 frames per launch, two batches alternating (2 GiB of sources)
    srcs = [torch.randint(0, 256, (16, R, R, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    outs = [torch.empty((16, 1024, 1024, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    p = zg.Pipeline([zg.Step.resize(1024, 1024)])
    it = [0]
    def f():
        it[0] += 1
        p.run(srcs[it[0] % 2], out=outs[it[0] % 2])
elif op == "recipe":  # the CLI's example recipe over 64 x 1080p Rgba(u8): resize lanczos -> gaussian sigma 2 -> edges sobel
    src = torch.randint(0, 256, (64, 1080, 1920, 4), dtype=torch.uint8, device="cuda")
    out = torch.empty((64, 450, 800, 4), dtype=torch.uint8, device="cuda")
    p = zg.Pipeline([zg.Step.resize(450, 800, I.lanczos), zg.Step.gaussian_blur(2.0), zg.Step.edges_sobel()])
    f = lambda: p.run(src, out=out)
elif op == "conv5_u8":
    t = torch.randint(0, 256, (R, R, 4), dtype=torch.uint8, device="cuda")
    s = zg.Image(t); d = zg.Image(torch.empty_like(t)); k5 = np.full((5, 5), 1 / 25, np.float32)
    f = lambda: s.convolve(k5, 1, out=d)
elif op in ("conv3_u8", "conv3_f32"):
    t = torch.rand((R, R, 4), dtype=torch.float32, device="cuda") if op.endswith("f32") else torch.randint(0, 256, (R, R, 4), dtype=torch.uint8, device="cuda")
    s = zg.Image(t); d = zg.Image(torch.empty_like(t)); k3 = np.full((3, 3), 1 / 9, np.float32)
    f = lambda: s.convolve(k3, 1, out=d)
elif op in ("png_filter", "png_filter_paeth", "png_decode", "png_encode"):
    import time
    yy, xx = torch.meshgrid(torch.arange(R, device="cuda"), torch.arange(R, device="cuda"), indexing="ij")
    smooth = torch.stack([(xx // 8) % 256, (yy // 8) % 256, ((xx + yy) // 16) % 256, torch.full_like(xx, 255)], -1)
    t = (smooth + torch.randint(0, 4, (R, R, 4), device="cuda")).clamp(0, 255).to(torch.uint8)  # photo-like: smooth + a little noise
    s = zg.Image(t)
    if op.startswith("png_filter"):
        f = lambda: zg.png.filter_scanlines(s, 4 if op.endswith("paeth") else -1)
    else:
        t0 = time.perf_counter(); data = zg.png.encode(s); t1 = time.perf_counter()
        print(f"encode {1e3 * (t1 - t0):.1f} ms, {len(data) / 2**20:.1f} MiB file from {t.numel() / 2**20:.0f} MiB of pixels")
        if op == "png_decode":
            def f():
                t0 = time.perf_counter(); out = zg.png.load_from_bytes(data); torch.cuda.synchronize(); t1 = time.perf_counter()
                print(f"decode {1e3 * (t1 - t0):.1f} ms")
        else:
            def f():
                t0 = time.perf_counter(); zg.png.encode(s); t1 = time.perf_counter()
                print(f"encode {1e3 * (t1 - t0):.1f} ms")
elif op in ("jpeg420", "jpeg444", "jpeg420p"):
    import io, time
    from PIL import Image as PI
    yy, xx = np.mgrid[0:R, 0:R].astype(np.float32)
    pic = np.stack([128 + 100 * np.sin(xx / 170) * np.cos(yy / 230), 128 + 90 * np.cos(xx / 110) * np.sin(yy / 130), 128 + 110 * np.sin((xx + yy) / 290)], -1)
    pic = np.clip(pic + np.random.default_rng(0).normal(0, 4, pic.shape), 0, 255).astype(np.uint8)
    buf = io.BytesIO(); PI.fromarray(pic).save(buf, "JPEG", quality=90, subsampling=0 if op == "jpeg444" else 2, progressive=op.endswith("p")); data = buf.getvalue()
    print(f"{op}: {len(data) / 2**20:.1f} MiB file")
    from concurrent.futures import ThreadPoolExecutor
    def one(_):
        with torch.cuda.stream(torch.cuda.Stream()):
            out = zg.jpeg.load_from_bytes(data); torch.cuda.current_stream().synchronize()
        return out
    dev_pic = zg.Image(torch.from_numpy(pic).cuda())
    def f():
        t0 = time.perf_counter(); out = zg.jpeg.load_from_bytes(data); torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"decode {1e3 * (t1 - t0):.1f} ms")
        t0 = time.perf_counter(); enc = zg.jpeg.encode(dev_pic, zg.jpeg.EncodeOptions(subsampling=0 if op == "jpeg444" else 2)); t1 = time.perf_counter()
        print(f"encode {1e3 * (t1 - t0):.1f} ms, {len(enc) / 2**20:.1f} MiB")
        for nt in (16,):
            with ThreadPoolExecutor(nt) as ex:
                t0 = time.perf_counter(); list(ex.map(one, range(nt * 2))); t1 = time.perf_counter()
            print(f"  {nt} threads: {nt * 2} files in {1e3 * (t1 - t0):.1f} ms = {nt * 2 * R * R / (t1 - t0) / 1e6:.0f} Mpixels/s")
for _ in range(n):
    f()
torch.cuda.synchronize()
