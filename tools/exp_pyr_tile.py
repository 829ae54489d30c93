"""Round 6: parity + timing of the tile kernel of ImagePyramid.build (pyramid_tile.hip) against the oracle. usage: python tools/exp_pyr_tile.py [check] [time]
(ZIGNAL_HIP_NO_PYRAMID_TILE=1 selects round 5's temp-plane route for the A/B)"""
import os
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

import bench
import zignal_amd as zg
from oracle import pyoracle as oracle  # checker only

what = sys.argv[1:] or ["check", "time"]

if "check" in what:
    ok, n_cases, t0 = True, 0, time.time()
    rng = np.random.default_rng(7)
    cases = [((4096, 4096), 8, 1.2, 1.6), ((1080, 1920), 8, 1.2, 1.6), ((700, 517), 6, 1.5, 1.0), ((513, 1031), 5, 1.9, 0.8), ((300, 260), 7, 1.3, 2.0),
             ((64, 64), 4, 1.2, 1.6), ((97, 131), 5, 1.1, 3.0), ((2000, 130), 6, 1.25, 1.2), ((130, 2000), 6, 1.25, 1.2), ((1000, 1000), 3, 3.5, 0.7),
             ((1024, 1024), 8, 1.2, 0.9), ((511, 513), 4, 2.0, 1.7)]
    for shape, n, sf, sg in cases:
        for fill in ("random", "white"):
            img = rng.integers(0, 256, shape, dtype=np.uint8) if fill == "random" else np.full(shape, 255, np.uint8)
            want = oracle.pyramid(img, n, sf, sg)
            pyr = zg.ImagePyramid.build(zg.Image(torch.from_numpy(img).cuda()), n, sf, sg)
            torch.cuda.synchronize()
            got = [l.to_numpy() for l in pyr.levels]
            n_cases += 1
            if len(got) != len(want):
                print(f"FAIL {shape} ({n},{sf},{sg}) {fill}: {len(got)} levels, want {len(want)}"); ok = False; continue
            for li, (g, w) in enumerate(zip(got, want)):
                if g.shape != w.shape or not np.array_equal(g, w):
                    bad = np.argwhere(g != w) if g.shape == w.shape else []
                    print(f"FAIL {shape} ({n},{sf},{sg}) {fill} level {li} {g.shape}: {len(bad)} differ; first {tuple(bad[0]) if len(bad) else None}"
                          + (f" got {g[tuple(bad[0])]} want {w[tuple(bad[0])]} rows {np.unique(bad[:,0])[:6]} cols {np.unique(bad[:,1])[:6]}" if len(bad) else ""))
                    ok = False
    # a view as the source (stride > cols) and levels that are views
    big = rng.integers(0, 256, (600, 800), dtype=np.uint8)
    view = zg.Image(torch.from_numpy(big).cuda()).view((8, 4, 8 + 512, 4 + 500))
    want = oracle.pyramid(np.ascontiguousarray(big[4:504, 8:520]), 6, 1.2, 1.6)
    pyr = zg.ImagePyramid.build(view, 6, 1.2, 1.6)
    torch.cuda.synchronize()
    for li, (g, w) in enumerate(zip([l.to_numpy() for l in pyr.levels], want)):
        if not np.array_equal(g, w):
            print(f"FAIL view source level {li}"); ok = False
    print(f"check {'OK' if ok else 'FAILED'}: {n_cases} pyramids in {time.time() - t0:.1f} s")
    if not ok:
        sys.exit(1)

if "time" in what:
    R = 4096
    src = [zg.Image(torch.randint(0, 256, (R, R), dtype=torch.uint8, device="cuda")) for _ in range(4)]
    for _ in range(3):
        ms = bench._time_kernel(torch, lambda i: zg.ImagePyramid.build_default(src[i % 4]), n=8, warm=3)
        eager = bench._time_kernel(torch, lambda i: zg.ImagePyramid.build_default(src[i % 4]), n=8, warm=3, capture=False)
        print(f"time pyramid build_default 4096^2 u8: graph {ms * 1e3:.1f} us, eager {eager * 1e3:.1f} us  (tile kernel off: {os.environ.get('ZIGNAL_HIP_NO_PYRAMID_TILE') is not None})")
