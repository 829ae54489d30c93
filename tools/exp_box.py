"""Round 6: parity + timing of the fused box blur (box_fused.hip) against the oracle and against the integral-image route.
usage: python tools/exp_box.py [check] [time]   (ZIGNAL_HIP_BOX_UNFUSED=1 selects the old route for the A/B)"""
import os
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

import zignal_amd as zg
from oracle import pyoracle as oracle  # checker only

what = sys.argv[1:] or ["check", "time"]
dev = lambda a: zg.Image(torch.from_numpy(a).cuda())


def sync(img):
    torch.cuda.synchronize()
    return img.to_numpy()


def same(got, want, name):
    if np.array_equal(got, want):
        return True
    bad = np.argwhere(got != want)
    rows = np.unique(bad[:, 0])
    cols = np.unique(bad[:, 1])
    print(f"FAIL {name}: {len(bad)} of {got.size} differ; rows {rows[:8]}..{rows[-3:]} cols {cols[:8]}..{cols[-3:]} first {tuple(bad[0])}: got {got[tuple(bad[0])]} want {want[tuple(bad[0])]}")
    return False


if "check" in what:
    ok = True
    t0 = time.time()
    for kind, tail in (("u8", ()), ("rgba_u8", (4,))):
        for rows, cols in ((64, 64), (65, 67), (100, 130), (257, 1031), (300, 64), (131, 200), (1080, 1920), (4096, 4096), (700, 4100)):
            src = oracle.synth_u8(rows * 7 + cols, (rows, cols) + tail)
            for radius in (1, 2, 3):
                ok &= same(sync(dev(src).box_blur(radius)), oracle.box_blur(src, radius), f"box {kind} {rows}x{cols} r={radius}")
            ok &= same(sync(dev(src).sharpen(2)), oracle.sharpen(src, 2), f"sharpen {kind} {rows}x{cols} r=2")
        # all white (the largest SAT values), all black
        for fill in (255, 0):
            src = np.full((2048, 3000) + tail, fill, np.uint8)
            ok &= same(sync(dev(src).box_blur(2)), oracle.box_blur(src, 2), f"box {kind} fill {fill}")
        # in place
        src = oracle.synth_u8(77, (500, 333) + tail)
        t = dev(src)
        t.box_blur(3, out=t)
        ok &= same(sync(t), oracle.box_blur(src, 3), f"in place {kind}")
        t = dev(src)
        t.sharpen(1, out=t)
        ok &= same(sync(t), oracle.sharpen(src, 1), f"sharpen in place {kind}")
        # views: source and destination inside larger allocations
        big = oracle.synth_u8(78, (400, 520) + tail)
        td = torch.full((420, 560) + tail, 0x5A, dtype=torch.uint8, device="cuda")
        rect = (8, 3, 8 + 300, 3 + 350)  # l, t, r, b
        zg.Image(torch.from_numpy(big).cuda()).view((4, 2, 304, 352)).box_blur(2, out=zg.Image(td).view(rect))
        got = sync(zg.Image(td))
        ok &= same(got[3:353, 8:308], oracle.box_blur(np.ascontiguousarray(big[2:352, 4:304]), 2), f"views {kind}")
        got[3:353, 8:308] = 0x5A
        ok &= bool(np.all(got == 0x5A)) or print(f"FAIL views {kind}: wrote outside the destination view")
    # determinism
    src = oracle.synth_u8(5, (4096, 4096, 4))
    d = dev(src)
    first = sync(d.box_blur(2))
    for _ in range(5):
        ok &= same(sync(d.box_blur(2)), first, "determinism")
    print(f"check {'OK' if ok else 'FAILED'} in {time.time() - t0:.1f} s")
    if not ok:
        sys.exit(1)

if "time" in what:
    R = 4096
    rr = tuple(int(x) for x in os.environ.get("BOX_RADII", "1,2,3").split(","))
    kinds = os.environ.get("BOX_KINDS", "u8,rgba_u8").split(",")
    for kind, tail, radii in (("u8", (), rr), ("rgba_u8", (4,), rr)):
        if kind not in kinds:
            continue
        n_src = 16 if not tail else 8  # past the Infinity Cache
        srcs = [zg.Image(torch.randint(0, 256, (R, R) + tail, dtype=torch.uint8, device="cuda")) for _ in range(n_src)]
        dsts = [zg.Image(torch.empty((R, R) + tail, dtype=torch.uint8, device="cuda")) for _ in range(2)]
        for radius in radii:
            for i in range(30):
                srcs[i % n_src].box_blur(radius, out=dsts[i % 2])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 100
            e0.record()
            for i in range(n):
                srcs[i % n_src].box_blur(radius, out=dsts[i % 2])
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / n
            bytes_alg = 2 * R * R * (4 if tail else 1)
            print(f"time {kind} r={radius}: {us:.1f} us  {bytes_alg / us / 1e6:.2f} TB/s algorithmic = {bytes_alg / us / 1e6 / 8:.3f} of 8 TB/s  (unfused={os.environ.get('ZIGNAL_HIP_BOX_UNFUSED') is not None} debug={os.environ.get('ZIGNAL_HIP_BOX_DEBUG')})")
