#!/usr/bin/env python
"""Randomised GPU-vs-oracle parity sweep: random shapes (biased to tile seams), pixel types, parameters, views.
usage: python tests/fuzz_parity.py [seconds] [seed] [max_rows max_cols]   — exits non-zero on the first mismatch, printing the case."""
import math
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

import zignal_amd as zg
from oracle import pyoracle as o

KINDS = ("u8", "f32", "rgb_u8", "rgba_u8", "rgb_f32", "rgba_f32")
SEAMS = (1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 272, 352, 511, 512, 513, 1023, 1024, 1025, 1040,
         2047, 2048, 2049, 2064, 4095, 4096, 4097, 4112)
MAX_ROWS, MAX_COLS = 300, 1100  # overridden by argv[3], argv[4]


def synth(rng, kind, rows, cols):
    ch = {"u8": (), "f32": (), "rgb_u8": (3,), "rgba_u8": (4,), "rgb_f32": (3,), "rgba_f32": (4,)}[kind]
    if kind.endswith("u8"):
        return rng.integers(0, 256, (rows, cols) + ch, dtype=np.uint8)
    return rng.random((rows, cols) + ch, dtype=np.float32)


def dim(rng, cap):
    return int(rng.choice([d for d in SEAMS if d <= cap])) if rng.random() < 0.6 else int(rng.integers(1, cap + 1))


def dev(a):
    return zg.Image(torch.from_numpy(np.ascontiguousarray(a)).cuda())


def dev_view(rng, a):
    """The same pixels as a view into a larger device frame (stride > cols, arbitrary — possibly unaligned — origin)."""
    rows, cols = a.shape[:2]
    top, left, bottom, right = (int(rng.integers(0, 6)) for _ in range(4))
    big = np.zeros((rows + top + bottom, cols + left + right) + a.shape[2:], a.dtype)
    big[...] = 123 if a.dtype == np.uint8 else 0.5
    big[top:top + rows, left:left + cols] = a
    return zg.Image(torch.from_numpy(big).cuda()).view((left, top, left + cols, top + rows))


def same(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def case(rng):
    kind = str(rng.choice(KINDS))
    op = str(rng.choice(["blur", "sep", "conv2d", "box", "resize", "warp", "rotate", "convert", "sobel", "canny", "shen", "isef", "motion", "insert_flip", "letterbox_extract", "misc8", "codec", "pipeline", "pipeline", "pyramid", "planes", "resize_convert"]))
    rows, cols = dim(rng, MAX_ROWS), dim(rng, MAX_COLS)
    img = synth(rng, kind, rows, cols)
    border = int(rng.integers(0, 4))
    use_view = rng.random() < 0.3
    D = (lambda a: dev_view(rng, a)) if use_view else dev
    I = zg.Interpolation
    methods = [(I.nearest, o.NEAREST), (I.bilinear, o.BILINEAR), (I.bicubic, o.BICUBIC), (I.catmull_rom, o.CATMULL_ROM), (I.lanczos, o.LANCZOS)]
    if op == "codec":
        return codec_case(rng)
    if op == "pipeline":
        return pipeline_case(rng, kind)
    flat = lambda arrays: np.concatenate([np.ascontiguousarray(a).reshape(-1).view(np.uint8) for a in arrays]) if arrays else np.zeros(0, np.uint8)
    if op == "pyramid":  # ImagePyramid.build: every level = blur of the source + bilinear resize (fused from a reduction by 2 up on Image(u8))
        kind = str(rng.choice(["u8", "u8", "u8", "f32", "rgba_u8"]))
        cols = max(cols, int(rng.choice([256, 272, 512, 640, 1040]))) if rng.random() < 0.7 else cols  # the packed two-pass path needs >= 256 columns
        img = synth(rng, kind, rows, cols)
        n, sf, sg = int(rng.integers(1, 9)), float(rng.choice([1.2, 1.3, 1.5, 2.0, 2.1, 3.0])), float(rng.choice([0.8, 1.0, 1.6, 2.91, 3.5]))
        want = o.pyramid(img, n, sf, sg)
        pyr = zg.ImagePyramid.build(D(img), n, sf, sg)
        torch.cuda.synchronize()
        if pyr.n_levels != len(want):
            return f"pyramid {kind} {rows}x{cols} ({n},{sf},{sg}) level count", np.zeros(1, np.uint8), np.ones(1, np.uint8)
        return f"pyramid {kind} {rows}x{cols} ({n},{sf},{sg})", flat([l.to_numpy() for l in pyr.levels]), flat(want)
    if op == "planes":  # several Image(f32) planes through one launch (zg_conv_separable_planes / zg_gaussian_blur_planes), shapes and views mixed in
        npl = int(rng.integers(1, 12))
        cols4 = max(4, cols // 4 * 4) if rng.random() < 0.8 else cols
        planes = [synth(rng, "f32", rows, cols4) - np.float32(0.5) for _ in range(npl)]
        if npl > 2 and rng.random() < 0.3:
            planes[1] = synth(rng, "f32", max(1, rows // 2), cols4)  # one plane of another shape breaks the run
        devs = [D(pl) if rng.random() < 0.2 else dev(pl) for pl in planes]
        if rng.random() < 0.5:
            sigma = float(rng.choice([0.3, 0.6, 1.0]))
            outs = zg.gaussian_blur_planes(devs, sigma)
            want = [o.gaussian_blur(pl, sigma) for pl in planes]
            name = f"planes f32 {npl}x{rows}x{cols4} sigma={sigma}"
        else:
            nk = int(rng.choice([1, 3, 5, 7, 9]))
            kx = rng.random(nk).astype(np.float32) - np.float32(0.3); ky = rng.random(nk).astype(np.float32) - np.float32(0.3)
            if rng.random() < 0.3: kx[0] = 0.0
            outs = zg.convolve_separable_planes(devs, kx, ky, border)
            want = [o.conv_separable(pl, kx, ky, border) for pl in planes]
            name = f"planes f32 {npl}x{rows}x{cols4} n={nk} b={border}"
        torch.cuda.synchronize()
        return name, flat([x.to_numpy() for x in outs]), flat(want)
    if op == "resize_convert":  # [resize, convert] as one call: the fused kernels (Rgba(u8), bilinear -> Oklab / Xyz) and the two-step route
        kind = str(rng.choice(["rgba_u8", "rgba_u8", "rgb_u8", "u8"]))
        img = synth(rng, kind, rows, cols)
        dr, dc = dim(rng, 200), dim(rng, 600)
        sp = int(rng.choice([zg.CS_OKLAB, zg.CS_XYZ, zg.CS_LAB]))
        src_space = {"u8": o.CS_GRAY, "rgb_u8": o.CS_RGB, "rgba_u8": o.CS_RGBA}[kind]
        small = o.resize(img, (dr, dc), o.method(o.BILINEAR))
        return f"resize_convert {kind} {rows}x{cols}->{dr}x{dc} space={sp}", D(img).resize_convert((dr, dc), sp), o.convert(small, src_space, sp, np.float32, 3)
    if op == "blur":
        sigma = float(rng.choice([0.3, 0.6, 1.0, 1.4, 2.25, 3.3, 5.5]))
        return f"blur {kind} {rows}x{cols} sigma={sigma}", D(img).gaussian_blur(sigma), o.gaussian_blur(img, sigma)
    if op == "sep":
        nx, ny = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        if rng.random() < 0.35:  # equal odd tap counts: the fused single-launch kernels (one instantiation per count and pixel type)
            nx = ny = int(rng.choice([1, 3, 5, 7, 9, 11, 13]))
        if rng.random() < 0.5:  # non-negative, normalised (the packed u8 paths)
            kx = rng.random(nx).astype(np.float32); kx /= kx.sum()
            ky = rng.random(ny).astype(np.float32); ky /= ky.sum()
        else:
            kx = (rng.random(nx).astype(np.float32) - np.float32(0.3)); ky = (rng.random(ny).astype(np.float32) - np.float32(0.3))
        return f"sep {kind} {rows}x{cols} n=({nx},{ny}) b={border}", D(img).convolve_separable(kx, ky, border), o.conv_separable(img, kx, ky, border)
    if op == "conv2d":
        kh, kw = int(rng.integers(1, 10)), int(rng.integers(1, 10))
        k = (rng.random((kh, kw)).astype(np.float32) - np.float32(0.3)) / np.float32(kh * kw * 0.3)
        return f"conv2d {kind} {rows}x{cols} {kh}x{kw} b={border}", D(img).convolve(k, border), o.convolve(img, k, border)
    if op == "box":
        rad = int(rng.integers(0, 9))
        return f"box {kind} {rows}x{cols} r={rad}", D(img).box_blur(rad), o.box_blur(img, rad)
    if op == "resize":
        m, om = methods[int(rng.integers(0, len(methods)))]
        dr, dc = dim(rng, 200), dim(rng, 600)
        return f"resize {kind} {rows}x{cols}->{dr}x{dc} {om}", D(img).resize((dr, dc), m), o.resize(img, (dr, dc), o.method(om))
    if op == "warp":
        m, om = methods[int(rng.integers(0, 3))]
        pts = [(0, 0), (cols - 1, 0), (0, rows - 1), (cols - 1, rows - 1)]
        to = [(x + float(rng.uniform(-0.1, 0.1)) * cols, y + float(rng.uniform(-0.1, 0.1)) * rows) for x, y in pts]
        if rows < 2 or cols < 2:
            return None
        h = o.homography_from_4pts(pts, to)
        return (f"warp {kind} {rows}x{cols} {om}", D(img).warp(zg.ProjectiveTransform(h), (rows, cols), m),
                o.warp(img, (rows, cols), o.PROJECTIVE, h, o.method(om)))
    if op == "rotate":
        m, om = methods[int(rng.integers(0, 3))]
        ang = float(rng.choice([0.0, math.pi / 2, math.pi, 0.3, -1.2, 2.5]))
        cs = o.cos_sin(ang)  # both sides get the same @cos / @sin values (a Zig caller passes Zig's)
        return f"rotate {kind} {rows}x{cols} a={ang} {om}", D(img).rotate(ang, m, border, cos_sin=cs), o.rotate(img, ang, o.method(om), border)
    if op == "convert":
        if kind in ("u8", "f32"):
            return None
        spaces = ["OKLAB", "XYZ", "LAB", "LCH", "OKLCH", "XYB", "HSL", "HSV", "LMS", "YCBCR"]
        sp = getattr(zg, "CS_" + str(rng.choice(spaces)))
        ss = zg.CS_RGBA if kind.startswith("rgba") else zg.CS_RGB
        return f"convert {kind} -> {sp}", D(img).convert(sp, np.float32), o.convert(img, ss, sp, np.float32, 3)
    if op == "sobel":
        return f"sobel {kind} {rows}x{cols}", D(img).sobel(), o.sobel(img)
    if op == "canny":
        sg = float(rng.choice([0.0, 1.0, 1.4])); lo = float(rng.uniform(1, 40)); hi = lo + float(rng.uniform(1, 80))
        return f"canny {kind} {rows}x{cols} {sg} {lo} {hi}", D(img).canny(sg, lo, hi), o.canny(img, sg, lo, hi)
    if op == "shen":
        kw = dict(smooth=float(rng.uniform(0.5, 0.95)), window_size=int(rng.choice([3, 5, 7, 11])), high_ratio=float(rng.uniform(0.5, 0.99)),
                  low_rel=float(rng.uniform(0.1, 0.9)), hysteresis=bool(rng.integers(0, 2)), use_nms=bool(rng.integers(0, 2)))
        return f"shen {kind} {rows}x{cols} {kw}", D(img).shen_castan(**kw), o.shen_castan(img, **kw)
    if op == "isef":  # shenCastan's smoothing stage alone (the segmented recursions and their repair launch), bytes or f32 in, the f32 plane out
        plane = rng.integers(0, 256, (rows, cols), dtype=np.uint8) if rng.random() < 0.5 else ((rng.random((rows, cols), dtype=np.float32) - 0.3) * 300).astype(np.float32)
        smooth = float(rng.choice([0.95, 0.9, 0.8, 0.7, 0.6, 0.45, 0.2]))
        return f"isef {plane.dtype} {rows}x{cols} {smooth}", dev(plane).isef_smooth(smooth), o.isef_plane(plane.astype(np.float32), smooth)
    if op == "motion":
        if rng.random() < 0.5:
            ang, d = float(rng.choice([0.0, math.pi / 2, 0.4, 2.2, -0.9])), int(rng.integers(0, 25))
            return f"motion linear {kind} {rows}x{cols} {ang} {d}", D(img).motion_blur_linear(ang, d), o.motion_blur_linear(img, ang, d)
        cx, cy, st, spin = float(rng.uniform(-0.2, 1.2)), float(rng.uniform(-0.2, 1.2)), float(rng.uniform(0, 1.3)), bool(rng.integers(0, 2))
        return f"motion radial {kind} {rows}x{cols} {cx} {cy} {st} {spin}", D(img).motion_blur_radial(cx, cy, st, spin), o.motion_blur_radial(img, cx, cy, st, spin)
    if op == "insert_flip":
        if rng.random() < 0.3:
            d = dev(img.copy())
            d.flip_left_right()
            return f"flip_lr {kind} {rows}x{cols}", d, img[:, ::-1]
        skind = str(rng.choice(KINDS))
        source = synth(rng, skind, dim(rng, 80), dim(rng, 120))
        m, om = methods[int(rng.integers(0, 3))]
        ang = float(rng.choice([0.0, 0.0, 0.35, -1.1]))
        l, t = float(rng.uniform(-20, cols)), float(rng.uniform(-20, rows))
        if ang == 0.0 and rng.random() < 0.5:
            rect = (round(l), round(t), round(l) + source.shape[1], round(t) + source.shape[0])  # the 1:1 fast path
        else:
            rect = (l, t, l + float(rng.uniform(2, 150)), t + float(rng.uniform(2, 100)))
        blend = int(rng.integers(0, 13))
        cs = o.cos_sin(ang)
        want = o.insert(img.copy(), source, rect, ang, o.method(om), blend)
        got = D(img.copy()).insert(dev(source), rect, ang, m, blend, cos_sin=cs)
        return f"insert {skind}->{kind} {rows}x{cols} rect={rect} a={ang} blend={blend} view={use_view}", got, want
    if op == "misc8":  # the u8-family filters: sharpen / integral / invert, thresholds and morphology, enhancement, order statistics
        which = int(rng.integers(0, 9))
        u8kind = kind.endswith("u8")
        if which == 0:
            rad = int(rng.integers(0, 7))
            return f"sharpen {kind} {rows}x{cols} r={rad}", D(img).sharpen(rad), o.sharpen(img, rad)
        if which == 1:
            got = D(img).integral()
            return f"integral {kind} {rows}x{cols}", got.cpu().numpy(), o.integral(img)
        if which == 2 and kind != "f32":
            return f"invert {kind} {rows}x{cols}", D(img.copy()).invert(), o.invert(img.copy())
        if not u8kind:
            return None
        if which == 3 and kind == "u8":
            got, gt = D(img).threshold_otsu()
            want, wt = o.threshold_otsu(img)
            assert gt == wt, f"otsu threshold {gt} != {wt}"
            return f"otsu {rows}x{cols}", got, want
        if which == 4 and kind == "u8":
            rad, cc = int(rng.integers(1, 9)), float(rng.uniform(-10, 10))
            return f"adaptive {rows}x{cols} r={rad} c={cc}", D(img).threshold_adaptive_mean(rad, cc), o.threshold_adaptive_mean(img, rad, cc)
        if which == 5 and kind == "u8":
            k = (rng.random((int(rng.choice([1, 3, 5])), int(rng.choice([1, 3, 7])))) > 0.3).astype(np.uint8)
            mop, it = int(rng.integers(0, 4)), int(rng.integers(0, 4))
            mask = (img > 128).astype(np.uint8) * 255
            return f"morph {rows}x{cols} op={mop} it={it} k={k.shape}", D(mask)._morph(k, it, mop, None), o.morph(mask, k, it, mop)
        if which == 6:
            if rng.random() < 0.5:
                cut = float(rng.choice([0.0, 0.01, 0.2, 0.45]))
                return f"autocontrast {kind} {rows}x{cols} {cut}", D(img.copy()).autocontrast(cut), o.autocontrast(img.copy(), cut)
            return f"equalize {kind} {rows}x{cols}", D(img.copy()).equalize(), o.equalize(img.copy())
        if which == 7 and rows * cols <= 40000:
            rad, oop = int(rng.integers(0, 4)), int(rng.integers(0, 3))
            param = float(rng.uniform(0, 1)) if oop == 0 else float(rng.uniform(0, 0.49))
            return (f"orderstat {kind} {rows}x{cols} r={rad} op={oop} p={param} b={border}", D(img)._order_stat(rad, oop, param, border, None),
                    o.order_statistic_blur(img, rad, oop, param, border))
        return None
    if op == "letterbox_extract":
        m, om = methods[int(rng.integers(0, 3))]
        if rng.random() < 0.5:
            dr, dc = dim(rng, 200), dim(rng, 300)
            want = np.zeros((dr, dc) + img.shape[2:], img.dtype)
            wrect = o.letterbox(img, want, o.method(om))
            got, grect = D(img).letterbox((dr, dc), m)
            assert tuple(grect) == tuple(wrect), f"letterbox rect {grect} != {wrect}"
            return f"letterbox {kind} {rows}x{cols}->{dr}x{dc}", got, want
        dr, dc = dim(rng, 120), dim(rng, 160)
        ang = float(rng.choice([0.0, 0.5, -2.0]))
        rect = (float(rng.uniform(-10, cols)), float(rng.uniform(-10, rows)), float(rng.uniform(0, cols + 20)), float(rng.uniform(0, rows + 20)))
        rect = (min(rect[0], rect[2]), min(rect[1], rect[3]), max(rect[0], rect[2]) + 1, max(rect[1], rect[3]) + 1)
        cs = o.cos_sin(ang)
        out = np.empty((dr, dc) + img.shape[2:], img.dtype)
        want = o.extract(img, out, rect, ang, o.method(om), border)
        got = D(img).extract(rect, ang, (dr, dc), m, border, cos_sin=cs)
        return f"extract {kind} {rows}x{cols} rect={rect} a={ang}", got, want
    return None


def pipeline_case(rng, kind):
    """zg_batch_pipeline over a few frames against the oracle applied step by step to every frame (a recipe of 1-4 random steps)."""
    n = int(rng.integers(1, 6))
    rows, cols = dim(rng, min(MAX_ROWS, 200)), dim(rng, min(MAX_COLS, 600))
    frames = np.stack([synth(rng, kind, rows, cols) for _ in range(n)])
    I = zg.Interpolation
    methods = [(I.nearest, o.NEAREST), (I.bilinear, o.BILINEAR), (I.bicubic, o.BICUBIC), (I.catmull_rom, o.CATMULL_ROM)]
    steps, refs, desc = [], [], []
    cur_kind, r, c = kind, rows, cols
    for _ in range(int(rng.integers(1, 5))):
        what = str(rng.choice(["blur", "blur", "resize", "resize", "box", "convert", "half", "median", "motion", "edges"]))
        u8_kind = cur_kind in ("u8", "rgb_u8", "rgba_u8")
        if what == "median" and u8_kind:
            rad = int(rng.integers(0, 4))
            steps.append(zg.Step.median_blur(rad)); refs.append(lambda a, rad=rad: o.order_statistic_blur(a, rad, 0, 0.5)); desc.append(f"median{rad}")
            continue
        if what == "motion" and u8_kind:
            if rng.random() < 0.5:
                ang, dist = float(rng.uniform(-3.2, 3.2)) if rng.random() < 0.7 else 0.0, int(rng.integers(0, 12))
                cs = o.cos_sin(ang)
                steps.append(zg.Step.motion_blur_linear(ang, dist, cos_sin=cs)); refs.append(lambda a, ang=ang, dist=dist, cs=cs: o.motion_blur_linear(a, ang, dist, cos_sin=cs)); desc.append(f"motionlin{ang:.3f}:{dist}")
            else:
                cx, cy, st, spin = float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), bool(rng.random() < 0.5)
                steps.append(zg.Step.motion_blur_radial(cx, cy, st, spin)); refs.append(lambda a, cx=cx, cy=cy, st=st, spin=spin: o.motion_blur_radial(a, cx, cy, st, spin)); desc.append(f"motionrad{cx:.2f}:{cy:.2f}:{st:.2f}:{int(spin)}")
            continue
        if what == "edges" and u8_kind:
            space, ch = {"u8": (o.CS_GRAY, 1), "rgb_u8": (o.CS_RGB, 3), "rgba_u8": (o.CS_RGBA, 4)}[cur_kind]
            which = int(rng.integers(0, 3))
            if which == 0:
                step, det, name = zg.Step.edges_sobel(), o.sobel, "sobel"
            elif which == 1:
                sg, lo = float(rng.choice([0.8, 1.0, 1.4])), float(rng.uniform(10, 80))
                hi = lo + float(rng.uniform(0, 120))
                step, det, name = zg.Step.edges_canny(sg, lo, hi), (lambda g, sg=sg, lo=lo, hi=hi: o.canny(g, sg, lo, hi)), f"canny{sg}:{lo:.1f}:{hi:.1f}"
            else:
                nms = bool(rng.random() < 0.5)
                step, det, name = zg.Step.edges_shen_castan(use_nms=nms), (lambda g, nms=nms: o.shen_castan(g, use_nms=nms)), f"shen{int(nms)}"
            steps.append(step)
            refs.append(lambda a, det=det, space=space, ch=ch: o.convert(det(a if ch == 1 else o.convert(a, space, o.CS_GRAY, np.uint8, 1)), o.CS_GRAY, space, np.uint8, ch) if ch != 1 else det(a))
            desc.append(name)
            continue
        if what == "blur":
            sigma = float(rng.choice([0.0, 0.3, 0.6, 1.0, 1.4, 2.25]))
            steps.append(zg.Step.gaussian_blur(sigma)); refs.append(lambda a, sigma=sigma: o.gaussian_blur(a, sigma) if sigma > 0 else a.copy()); desc.append(f"blur{sigma}")
        elif what == "box":
            rad = int(rng.integers(0, 4))
            steps.append(zg.Step.box_blur(rad)); refs.append(lambda a, rad=rad: o.box_blur(a, rad)); desc.append(f"box{rad}")
        elif what in ("resize", "half"):
            if what == "half" and r % 2 == 0 and c % 4 == 0 and r >= 2:
                nr, nc, (zm, om) = r // 2, c // 2, methods[1]
            else:
                nr, nc, (zm, om) = dim(rng, min(MAX_ROWS, 200)), dim(rng, min(MAX_COLS, 600)), methods[int(rng.integers(0, len(methods)))]
            steps.append(zg.Step.resize(nr, nc, zm)); refs.append(lambda a, nr=nr, nc=nc, om=om: o.resize(a, (nr, nc), o.method(om))); desc.append(f"resize{nr}x{nc}:{om}")
            r, c = nr, nc
        elif what == "convert" and cur_kind in ("rgb_u8", "rgba_u8"):
            space, ospace = (zg.CS_OKLAB, o.CS_OKLAB) if rng.random() < 0.5 else (zg.CS_XYZ, o.CS_XYZ)
            src_space = o.CS_RGB if cur_kind == "rgb_u8" else o.CS_RGBA
            steps.append(zg.Step.convert(space)); refs.append(lambda a, src_space=src_space, ospace=ospace: o.convert(a, src_space, ospace, np.float32, 3)); desc.append(f"convert{space}")
            cur_kind = "lab_f32"
            break  # colour types past Rgb / Rgba are not inputs of the other steps here
    if not steps:
        return None
    got = zg.Pipeline(steps).run(torch.from_numpy(frames).cuda()).cpu().numpy()
    want = []
    for f in frames:
        a = f
        for ref in refs:
            a = ref(a)
        want.append(a)
    return f"pipeline {kind} {n}x{rows}x{cols} " + ",".join(desc), got, np.stack(want)


def codec_case(rng):
    """PNG / JPEG files in, PNG filter streams out: random formats, sizes, and — half the time — a cut or a flipped byte.
    Both sides must then fail with the same error name or produce the same pixels."""
    from tests import jpeg_util as J
    from tests import png_util as P
    which = int(rng.integers(0, 5))
    h, w = int(rng.integers(1, 90)), int(rng.integers(1, 140))
    if which == 4:  # jpeg.encode: the whole file, byte for byte
        grey = rng.random() < 0.3
        pic = J.test_image(h, w, seed=int(rng.integers(0, 1000)), smooth=bool(rng.integers(0, 2)))
        pic = pic[..., 0].copy() if grey else pic
        q, sub = int(rng.integers(1, 101)), int(rng.integers(0, 3))
        got = zg.jpeg.encode(dev(pic), zg.jpeg.EncodeOptions(quality=q, subsampling=sub))
        return f"jpegenc {pic.shape} q={q} sub={sub}", np.frombuffer(got, np.uint8), np.frombuffer(o.jpeg_encode(pic, q, sub), np.uint8)
    if which == 0:  # PNG filter stream (the device half of png.encode)
        ch = int(rng.choice([1, 3, 4]))
        rows = int(rng.choice([h, 513 + h]))
        img = rng.integers(0, 256, (rows, w, ch), dtype=np.uint8)
        if rng.random() < 0.5:
            img[rows // 4:] = img[rows // 4]  # long runs where one filter keeps winning
        img = img if ch > 1 else img[..., 0]
        mode = int(rng.integers(-1, 5))
        return f"pngfilter {img.shape} mode={mode}", zg.png.filter_scanlines(dev(img), mode).cpu().numpy(), o.png_filter(img, mode)
    if which == 1:
        ct, bd = [(P.GRAY, 1), (P.GRAY, 2), (P.GRAY, 4), (P.GRAY, 8), (P.GRAY, 16), (P.RGB, 8), (P.RGB, 16), (P.PALETTE, 1), (P.PALETTE, 2), (P.PALETTE, 4),
                  (P.PALETTE, 8), (P.GRAY_ALPHA, 8), (P.GRAY_ALPHA, 16), (P.RGBA, 8), (P.RGBA, 16)][int(rng.integers(0, 15))]
        plen = min(1 << bd, int(rng.integers(1, 257))) if ct == P.PALETTE else None
        samples = P.random_samples(rng, h, w, bd, ct, plen)
        trns = None
        if rng.random() < 0.4 and ct in (P.GRAY, P.RGB, P.PALETTE):
            trns = ([int(samples[0, 0, 0]) >> 8, int(samples[0, 0, 0]) & 255] if ct == P.GRAY else
                    [b for c in samples[0, 0] for b in (int(c) >> 8, int(c) & 255)] if ct == P.RGB else rng.integers(0, 256, int(rng.integers(1, plen + 1))).tolist())
        data = P.make_png(samples, bd, ct, int(rng.integers(0, 2)), filters=lambda y: int(rng.integers(0, 5)),
                          palette=rng.integers(0, 256, (plen, 3)).tolist() if plen else None, trns=trns, idat_split=int(rng.integers(0, 2)) * 211)
        load_g, load_o, tag = zg.png.load_from_bytes, o.png_decode_native, f"png ct={ct} bd={bd} {h}x{w}"
    else:
        pic = J.test_image(h, w, seed=int(rng.integers(0, 1000)), smooth=bool(rng.integers(0, 2)))
        kw = dict(quality=int(rng.integers(20, 100)), subsampling=int(rng.integers(0, 3)), progressive=bool(rng.integers(0, 2)), optimize=bool(rng.integers(0, 2)))
        if rng.random() < 0.2:
            kw["restart_marker_blocks"] = int(rng.integers(1, 9))
        grey = rng.random() < 0.25
        if grey:
            kw.pop("subsampling")
        data = J.pil_jpeg(pic[..., 0] if grey else pic, **kw)
        load_g, load_o, tag = zg.jpeg.load_from_bytes, o.jpeg_decode_native, f"jpeg {kw} {h}x{w}"
    damage = rng.random()
    if damage < 0.25:
        data = data[:int(rng.integers(0, len(data) + 1))]
    elif damage < 0.5:
        data = bytearray(data)
        data[int(rng.integers(0, len(data)))] = int(rng.integers(0, 256))
        data = bytes(data)
    try:
        want = load_o(data)[0]
        want_err = None
    except (o.PngError, o.JpegError) as e:
        want, want_err = None, e.name
    try:
        got = load_g(data)
        got_err = None
    except zg.CodecError as e:
        got, got_err = None, e.name
    if want_err != got_err:
        print(f"MISMATCH: {tag} (damage {damage:.2f}): oracle error {want_err}, product error {got_err}")
        sys.exit(1)
    if want_err is not None:
        return f"codec-error {tag}", np.zeros(1, np.uint8), np.zeros(1, np.uint8)
    return tag, got, want


def main():
    global MAX_ROWS, MAX_COLS
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
    if len(sys.argv) > 4:
        MAX_ROWS, MAX_COLS = int(sys.argv[3]), int(sys.argv[4])
    rng = np.random.default_rng(seed)
    t0, n, by_op = time.time(), 0, {}
    while time.time() - t0 < budget:
        try:
            c = case(rng)
        except (zg.ZignalError, RuntimeError) as e:  # both sides must agree that a case is invalid: re-raise if only one did
            print("skipped (error raised):", type(e).__name__, str(e)[:100])
            continue
        if c is None:
            continue
        name, got, want = c
        torch.cuda.synchronize()
        g = got.to_numpy() if hasattr(got, "to_numpy") else got
        if not same(g, want):
            diff = np.argwhere(np.ascontiguousarray(g).view(np.uint8).reshape(-1) != np.ascontiguousarray(want).view(np.uint8).reshape(-1))
            print(f"MISMATCH after {n} cases (seed {seed}): {name}; first differing byte {diff[0] if len(diff) else '?'} of {g.nbytes}")
            sys.exit(1)
        n += 1
        by_op[name.split()[0]] = by_op.get(name.split()[0], 0) + 1
    print(f"fuzz parity: {n} cases bit-identical in {time.time() - t0:.0f} s (seed {seed}); per op {by_op}")


if __name__ == "__main__":
    main()
