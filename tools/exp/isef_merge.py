"""How many steps two runs of the ISEF recursion temp[i] = b * x[i] + a * temp[i - 1] (edges.zig:283-305; separate f32 multiplies and an addition)
need before they hold the SAME f32 value, one of them started from the sequential history and the other from zero. a = 1 - b < 1 shrinks the
difference by a per step; once the values are equal they stay equal. isef.hip's k_isef_spec starts every segment W steps early from zero and relies
on this (and proves it per segment at run time); this script is where W's formula was read off. CPU only (numpy).

usage: python tools/exp/isef_merge.py > profiles/r04_isef_merge.txt
"""
import numpy as np

rng = np.random.default_rng(1)
N = 2_000_000
print("# chains per case:", N, "- columns: P(still unequal after k steps), as counts")
for smooth in (0.95, 0.9, 0.7, 0.5):
    b = np.float32(smooth)
    a = np.float32(1) - b
    for kind in ("noise", "smooth"):
        L = 96
        if kind == "noise":
            x = rng.integers(0, 256, (L + 40, N)).astype(np.float32)
        else:
            x = (128 + 60 * np.sin(np.arange(L + 40)[:, None] * 0.05 + rng.random(N)[None, :] * 6)).astype(np.float32)
        t = np.zeros(N, np.float32)
        for i in range(40):  # the sequential history
            t = (b * x[i]).astype(np.float32) + (a * t).astype(np.float32)
        s = np.zeros(N, np.float32)  # the segment's run, from zero
        last = np.zeros(N, int)
        for k in range(L):
            t = (b * x[40 + k]).astype(np.float32) + (a * t).astype(np.float32)
            s = (b * x[40 + k]).astype(np.float32) + (a * s).astype(np.float32)
            last[t != s] = k + 1
        h = np.bincount(last, minlength=L + 2)
        tail = np.cumsum(h[::-1])[::-1]
        ks = (8, 12, 16, 20, 24, 32, 40, 48, 64)
        print(f"smooth {smooth:4.2f} {kind:6s} last unequal step {last.max():3d} | " + " ".join(f"k={k}:{int(tail[k + 1])}" for k in ks))
