#!/usr/bin/env python
"""CPU-only fuzz of the codecs' host layers: zg_png_scan_hash / zg_jpeg_coefficient_hash / the probes against the oracle on random
files, cuts, byte flips, duplicated and dropped segments. usage: python tests/fuzz_host_layers.py [seconds] [seed]"""
import struct
import sys
import time

sys.path.insert(0, ".")
import numpy as np

import zignal_amd as zg
from oracle import pyoracle as o
from tests import jpeg_util as J
from tests import png_util as P
from tests.test_oracle_png import FORMATS


def outcome(fn, *a):
    try:
        return "ok", fn(*a)
    except Exception as e:
        if not hasattr(e, "name"):
            raise
        return "err", e.name


def damage(rng, data):
    data = bytearray(data)
    k = int(rng.integers(0, 5))
    if k == 0:
        return bytes(data)
    if k == 1:
        return bytes(data[:int(rng.integers(0, len(data) + 1))])
    if k == 2:
        for _ in range(int(rng.integers(1, 5))):
            data[int(rng.integers(0, len(data)))] = int(rng.integers(0, 256))
        return bytes(data)
    if k == 3:  # drop a slice
        a = int(rng.integers(0, len(data)))
        return bytes(data[:a] + data[a + int(rng.integers(1, 40)):])
    a, b = sorted(int(x) for x in rng.integers(0, len(data), 2))  # duplicate a slice
    return bytes(data[:b] + data[a:b] + data[b:])


TRACE = "--trace" in sys.argv  # keep the input of the call in flight on disk (for crashes)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0, n, names = time.time(), 0, {}
    while time.time() - t0 < budget:
        h, w = int(rng.integers(1, 70)), int(rng.integers(1, 90))
        if rng.random() < 0.5:
            ct, bd = FORMATS[int(rng.integers(0, len(FORMATS)))]
            plen = min(1 << bd, int(rng.integers(1, 257))) if ct == P.PALETTE else None
            s = P.random_samples(rng, h, w, bd, ct, plen)
            base = P.make_png(s, bd, ct, int(rng.integers(0, 2)), filters=lambda y: int(rng.integers(0, 5)),
                              palette=rng.integers(0, 256, (plen, 3)).tolist() if plen else None, idat_split=int(rng.integers(0, 2)) * 53, level=int(rng.integers(0, 10)))
            pairs = [(o.png_scan_hash, zg.png.scan_hash), (lambda d: tuple(o.png_decode_chunks(d)[1:2]), lambda d: tuple(zg.png.decode(d)[2:3]))]
        else:
            pic = J.test_image(h, w, seed=int(rng.integers(0, 999)), smooth=bool(rng.integers(0, 2)))
            kw = dict(quality=int(rng.integers(5, 100)), subsampling=int(rng.integers(0, 3)), progressive=bool(rng.integers(0, 2)), optimize=bool(rng.integers(0, 2)))
            if rng.random() < 0.3:
                kw["restart_marker_blocks"] = int(rng.integers(1, 6))
            if rng.random() < 0.2:
                kw.pop("subsampling")
                pic = pic[..., 0]
            base = J.pil_jpeg(pic, **kw)
            pairs = [(o.jpeg_coefficient_hash, zg.jpeg.coefficient_hash), (lambda d: o.jpeg_decode_state(d)[1], lambda d: zg.jpeg.decode(d)[1])]
        for _ in range(6):
            data = damage(rng, base)
            for want_fn, got_fn in pairs:
                if TRACE:
                    open("/tmp/fuzz_host_last.bin", "wb").write(data)
                want, got = outcome(want_fn, data), outcome(got_fn, data)
                if want != got:
                    open("/tmp/fuzz_host_mismatch.bin", "wb").write(data)
                    print(f"MISMATCH after {n} cases: oracle {want}, product {got}; input saved to /tmp/fuzz_host_mismatch.bin")
                    sys.exit(1)
                names[want[1] if want[0] == "err" else "ok"] = names.get(want[1] if want[0] == "err" else "ok", 0) + 1
                n += 1
    print(f"host-layer fuzz: {n} comparisons identical in {time.time() - t0:.0f} s; outcomes {dict(sorted(names.items(), key=lambda kv: -kv[1]))}")


if __name__ == "__main__":
    main()
