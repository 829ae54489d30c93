"""Round-5 A/B timings on one box, one process: the f32 plane Gaussian (the LDS-tiled kernel of round 4 vs the tile-per-wave kernel, one and
four planes per launch) and the Rgba(u8) Gaussian / config 5 with and without the folded unit-end taps. Rings of >= 1 GiB, graph-replayed
(bench._time_kernel), every variant twice, interleaved. usage: python tools/exp_r05.py [what ...]   (what: f32 u8 all; default all)"""
import os
import sys

sys.path.insert(0, ".")
import ctypes as C

import torch

import bench
import zignal_amd as zg

what = set(sys.argv[1:]) or {"all"}
R = 4096


class knob:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, str(v))

    def __exit__(self, *exc):
        for k, v in self.old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


def warm():
    x = torch.rand((4096, 4096), device="cuda")
    for _ in range(400):
        x = x * 1.0001
    torch.cuda.synchronize()


def report(name, vals):
    print(f"{name:58s} " + "  ".join(f"{v * 1e3:7.2f}" for v in vals) + "   us", flush=True)


warm()
if what & {"all", "f32"}:
    ring = 8
    planes = [(zg.Image(torch.rand((R, R), device="cuda")), zg.Image(torch.empty((R, R), device="cuda"))) for _ in range(ring)]
    one = lambda i: planes[i % ring][0].gaussian_blur(0.6, out=planes[i % ring][1])
    variants = [("LDS-tiled k_sep_f32x4 (round 4)", dict(ZIGNAL_HIP_NO_TILE_F32=1)), ("tile per wave, halo by every lane", dict(ZIGNAL_HIP_F32_TILE_HALO="all")),
                ("tile per wave, halo by the outer lanes", dict(ZIGNAL_HIP_F32_TILE_HALO="outer")), ("default", {})]
    res = {n: [] for n, _ in variants}
    for rep in range(3):
        for n, kv in variants:
            with knob(**kv):
                res[n].append(bench._time_kernel(torch, one, n=48, warm=8))
    for n, _ in variants:
        report("f32 plane 4096^2 gaussianBlur(0.6): " + n, res[n])
    del planes
    ring4 = 4  # four planes per launch (channel-major RGBA f32), 2 GiB of planes
    quads = [([zg.Image(torch.rand((R, R), device="cuda")) for _ in range(4)], [zg.Image(torch.empty((R, R), device="cuda")) for _ in range(4)]) for _ in range(ring4)]
    four = lambda i: zg.gaussian_blur_planes(quads[i % ring4][0], 0.6, outs=quads[i % ring4][1])
    four_calls = lambda i: [quads[i % ring4][0][c].gaussian_blur(0.6, out=quads[i % ring4][1][c]) for c in range(4)]
    variants = [("one launch, default", four, {}), ("one launch, halo by every lane", four, dict(ZIGNAL_HIP_F32_TILE_HALO="all")),
                ("four launches", four_calls, {}), ("four launches, LDS-tiled k_sep_f32x4 (round 4)", four_calls, dict(ZIGNAL_HIP_NO_TILE_F32=1))]
    res = {n: [] for n, _, _ in variants}
    for rep in range(2):
        for n, fn, kv in variants:
            with knob(**kv):
                res[n].append(bench._time_kernel(torch, fn, n=16, warm=4))
    for n, _, _ in variants:
        report("f32 4 planes 4096^2: " + n, res[n])
    del quads

if what & {"resize"}:
    ring = 16
    srcs = [torch.randint(0, 256, (R, R, 4), dtype=torch.uint8, device="cuda") for _ in range(ring)]
    im = [(zg.Image(x), zg.Image(torch.empty((1024, 1024, 4), dtype=torch.uint8, device="cuda"))) for x in srcs]
    one = lambda i: im[i % ring][0].resize(im[i % ring][1], zg.Interpolation.bilinear)
    names = {0: "4 rows per workgroup, XCD-major (round 4)", 1: "one wave per workgroup, 1 row, address order", 2: "one wave per workgroup, 2 rows, address order",
             4: "one wave per workgroup, 4 rows, address order"}
    res = {k: [] for k in names}
    for rep in range(4):
        for k in names:
            with knob(ZIGNAL_HIP_RESIZE_FORM=k):
                res[k].append(bench._time_kernel(torch, one, n=64, warm=8))
    for k, n in names.items():
        report("resize 4096^2 -> 1024^2 Rgba(u8) bilinear: " + n, res[k])
    lab = [(zg.Image(x), zg.Image(torch.empty((1024, 1024, 3), dtype=torch.float32, device="cuda"))) for x in srcs]
    fused = lambda i: lab[i % ring][0].resize_convert(lab[i % ring][1], zg.CS_OKLAB)
    res = {0: [], 1: []}
    for rep in range(4):
        for k in res:
            with knob(ZIGNAL_HIP_RESIZE_FORM=k):
                res[k].append(bench._time_kernel(torch, fused, n=64, warm=8))
    report("fused resize -> Oklab 4096^2 -> 1024^2: 4 rows per workgroup, XCD-major (round 4)", res[0])
    report("fused resize -> Oklab 4096^2 -> 1024^2: one wave per workgroup, address order", res[1])
    del lab
    big = [(zg.Image(torch.randint(0, 256, (2 * R, 2 * R, 4), dtype=torch.uint8, device="cuda")), zg.Image(torch.empty((R, R, 4), dtype=torch.uint8, device="cuda"))) for _ in range(4)]
    two = lambda i: big[i % 4][0].resize(big[i % 4][1], zg.Interpolation.bilinear)
    res = {k: [] for k in names}
    for rep in range(2):
        for k in names:
            with knob(ZIGNAL_HIP_RESIZE_FORM=k):
                res[k].append(bench._time_kernel(torch, two, n=16, warm=4))
    for k, n in names.items():
        report("resize 8192^2 -> 4096^2 Rgba(u8) bilinear: " + n, res[k])
    del big
    up = [(zg.Image(torch.randint(0, 256, (R // 2, R // 2, 4), dtype=torch.uint8, device="cuda")), zg.Image(torch.empty((R, R, 4), dtype=torch.uint8, device="cuda"))) for _ in range(12)]
    upf = lambda i: up[i % 12][0].resize(up[i % 12][1], zg.Interpolation.bilinear)
    res = {k: [] for k in names}
    for rep in range(2):
        for k in names:
            with knob(ZIGNAL_HIP_RESIZE_FORM=k):
                res[k].append(bench._time_kernel(torch, upf, n=24, warm=4))
    for k, n in names.items():
        report("resize 2048^2 -> 4096^2 Rgba(u8) bilinear: " + n, res[k])
    del up, im, srcs

if what & {"all", "u8"}:
    ring = 8
    fr = [(zg.Image(torch.randint(0, 256, (R, R, 4), dtype=torch.uint8, device="cuda")), zg.Image(torch.empty((R, R, 4), dtype=torch.uint8, device="cuda"))) for _ in range(ring)]
    one = lambda i: fr[i % ring][0].gaussian_blur(0.6, out=fr[i % ring][1])
    variants = [("plain taps", dict(ZIGNAL_HIP_STREAM_NO_FOLD=1)), ("folded unit-end taps", {})]
    res = {n: [] for n, _ in variants}
    for rep in range(3):
        for n, kv in variants:
            with knob(**kv):
                res[n].append(bench._time_kernel(torch, one, n=48, warm=8))
    for n, _ in variants:
        report("Rgba(u8) 4096^2 gaussianBlur(0.6): " + n, res[n])
    del fr
    lib = zg.lib()
    m = zg.Interpolation.bilinear._c()
    for nf in (64, 128):
        src = [torch.randint(0, 256, (nf, 1080, 1920, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
        dst = [torch.empty((nf, 540, 960, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]

        def run(i):
            rc = lib.zg_batch_blur_resize(C.c_void_p(src[i % 2].data_ptr()), nf, 1080, 1920, 3, C.c_float(0.6), C.c_void_p(dst[i % 2].data_ptr()), 540, 960, C.byref(m),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, lib.zg_last_error()
        res = {"plain taps": [], "folded unit-end taps": []}
        for rep in range(3):
            for n, kv in (("plain taps", dict(ZIGNAL_HIP_STREAM_NO_FOLD=1)), ("folded unit-end taps", {})):
                with knob(**kv):
                    res[n].append(bench._time_kernel(torch, run, n=10, warm=2))
        for n in res:
            report(f"config 5, {nf} x 1080p [blur, resize 1/2]: " + n, res[n])
        del src, dst
