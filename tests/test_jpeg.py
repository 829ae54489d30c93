"""JPEG edge of the path (zg_jpeg_*, zignal_amd/csrc/jpeg_codec.hip) against the oracle (oracle/jpeg.c).

CPU part: marker parsing, limits and the progressive entropy decoder are host code (zg_jpeg_probe / zg_jpeg_info): they must
agree with the oracle's decode(), error name for error name, on hand-built streams and under random corruption. GPU part:
decoded pixels, bit for bit, for every layout Pillow and our own coefficient-level writer can produce."""
import struct

import numpy as np
import pytest

import zignal_amd as zg
from tests import jpeg_util as J
from tests.test_oracle_jpeg import DHT, DQT, EOI, PROGRESSIVE, SCAN1, SIG, SOF2


def outcome(fn, *a, **kw):
    try:
        return "ok", fn(*a, **kw)
    except Exception as e:
        if not hasattr(e, "name"):
            raise
        return "err", e.name


def head(h):
    return (h.width, h.height, h.precision, h.num_components, h.progressive)


def sof(marker=0xC0, precision=8, h=8, w=8, comps=((1, 0x11, 0),)):
    body = struct.pack(">BHHB", precision, h, w, len(comps)) + b"".join(bytes(c) for c in comps)
    return bytes([0xFF, marker]) + struct.pack(">H", 2 + len(body)) + body


def structural_cases():
    sof0 = bytes([0xFF, 0xC0]) + SOF2[2:]
    cases = [
        (b"", {}), (b"\x89PNG\r\n", {}), (SIG, {}), (SIG + EOI, {}), (SIG + sof() + EOI, {}), (SIG + b"\x00\x00\x00", {}),
        (SIG + sof(0xC1), {}), (SIG + sof(0xC3), {}), (SIG + bytes([0xFF, 0xCC, 0, 2]), {}), (SIG + bytes([0xFF, 0xDE, 0, 2]), {}),
        (SIG + bytes([0xFF, 0xDC, 0, 2]), {}), (SIG + sof(precision=12), {}), (SIG + sof(precision=16), {}), (SIG + sof(precision=9), {}),
        (SIG + sof(h=0), {}), (SIG + bytes([0xFF, 0xC0, 0, 5, 8, 0, 8]), {}), (SIG + sof(h=9000), {}),
        (SIG + sof(comps=((1, 0x11, 0),) * 4), {}), (SIG + sof(comps=((1, 0x11, 0),) * 2), {}), (SIG + sof(comps=()), {}),
        (SIG + sof(comps=((1, 0x22, 0), (2, 0x11, 1), (3, 0x12, 1))), {}), (SIG + sof(comps=((1, 0x12, 0), (2, 0x11, 1), (3, 0x11, 1))), {}),
        (SIG + sof(comps=((1, 0x51, 0),)), {}), (SIG + sof(comps=((1, 0x41, 0), (2, 0x11, 1), (3, 0x11, 1))) + EOI, {}),
        (SIG + sof(h=100, w=100), dict(max_pixels=9999)), (SIG + sof(h=100, w=100), dict(max_pixels=0, max_width=99)),
        (SIG + bytes([0xFF, 0xC4, 0, 2]), {}), (SIG + bytes([0xFF, 0xC4, 0, 5, 0, 1, 2]), {}), (SIG + bytes([0xFF, 0xC4, 0, 19, 0]) + bytes([255] * 16), {}),
        (SIG + bytes([0xFF, 0xC4, 0, 22, 0, 3]) + bytes(15) + bytes([1, 2, 3]), {}), (SIG + bytes([0xFF, 0xDB, 0, 2]), {}),
        (SIG + bytes([0xFF, 0xDB, 0, 10, 0]) + bytes(7), {}), (SIG + bytes([0xFF, 0xDD, 0, 5, 0, 0, 0]), {}), (SIG + bytes([0xFF, 0xDD, 0, 4, 0, 7]) + EOI, {}),
        (SIG + sof() + bytes([0xFF, 0xDA, 0, 8, 1, 1, 0, 1, 63, 0]), {}), (SIG + sof() + bytes([0xFF, 0xDA, 0, 8, 2, 1, 0, 0, 63, 0]), {}),
        (SIG + sof(0xC2) + bytes([0xFF, 0xDA, 0, 8, 1, 1, 0, 0, 5, 0]), {}), (SIG + sof(0xC2) + bytes([0xFF, 0xDA, 0, 8, 1, 1, 0, 9, 5, 0]), {}),
        (SIG + sof(0xC2) + bytes([0xFF, 0xDA, 0, 8, 0, 1, 0, 0, 0, 0]), {}), (SIG + sof(0xC2) + bytes([0xFF, 0xDA, 0, 4, 1, 1]), {}),
        (SIG + bytes([0xFF, 0xDB, 0, 1]), {}), (SIG + bytes([0xFF, 0xDB, 0, 200, 0]), {}), (SIG + bytes([0xFF, 0xDB, 0]), {}),
        (SIG + bytes([0xFF, 0x02, 0, 1]), {}), (SIG + bytes([0xFF, 0x02, 0, 4, 9, 9]) + sof() + EOI, {}), (SIG + bytes([0xFF, 0xF0]), {}),
        (SIG + bytes([0xFF, 0xE0, 0, 0]) + EOI, {}), (SIG + bytes([0xFF, 0xD3, 0, 2]) + EOI, {}), (SIG + bytes([0xFF, 0xFE, 0, 9]) + EOI, {}),
        (SIG, dict(max_jpeg_bytes=1)), (bytes([0xFF, 0xD8, 0xFF, 0xE0, 0x00, 0x04, 0x00, 0x00, 0xFF, 0xD9]), dict(max_jpeg_bytes=0, max_marker_bytes=2)),
        (SIG + bytes([0xFF, 0xC0, 0x00, 0x0B, 0x08, 0x00, 0x10, 0x00, 0x10, 0x01, 0x01, 0x11, 0x00]) + EOI, dict(max_blocks=1)),
        (PROGRESSIVE, {}), (PROGRESSIVE, dict(max_scans=2)), (PROGRESSIVE, dict(max_scans=0)), (PROGRESSIVE, dict(max_marker_bytes=100)),
        (SIG + sof0 + sof0 + EOI, {}), (PROGRESSIVE[:-4], {}), (SIG + DQT + SOF2 + DHT + SCAN1[:-1], {}), (SIG + DQT + SOF2 + DHT + SCAN1 + EOI, {}),
        (SIG + DQT + SOF2 + SCAN1 + EOI, {}),  # no Huffman table: MissingHuffmanTable from inside the scan
        (SIG + DQT + DHT + SCAN1 + EOI, {}),   # SOS before SOF
    ]
    return cases


def probe_both(oracle, data, limits):
    want = outcome(oracle.jpeg_decode_state, data, oracle.jpeg_limits(**limits) if limits else None)
    got = outcome(zg.jpeg.decode, data, zg.jpeg.decode_limits(**limits) if limits else None)
    return want, got


def same(want, got):
    if want[0] != got[0]:
        return False
    if want[0] == "err":
        return want[1] == got[1]
    return head(want[1][0]) == head(got[1][0]) and want[1][1] == got[1][1]


def test_marker_layer_matches_oracle_case_by_case(oracle):
    seen = set()
    for i, (data, limits) in enumerate(structural_cases()):
        want, got = probe_both(oracle, data, limits)
        assert same(want, got), (i, want, got)
        seen.add(want[1] if want[0] == "err" else "ok")
    assert len(seen) >= 25, seen  # the cases really do reach that many different outcomes


def test_get_info_matches_oracle(oracle):
    def info_tuple(h):
        return head(h) + (h.subsampling,)
    files = [J.pil_jpeg(J.test_image(20, 30), subsampling=s) for s in (0, 1, 2)] + [J.pil_jpeg(J.test_image(9, 5)[..., 0], progressive=True)]
    files += [d for d, _ in structural_cases()]
    for data in files:
        for lim in ({}, dict(max_jpeg_bytes=40), dict(max_jpeg_bytes=0)):
            want = outcome(oracle.jpeg_info, data, oracle.jpeg_limits(**lim) if lim else None)
            got = outcome(zg.jpeg.get_info, data, zg.jpeg.decode_limits(**lim) if lim else None)
            assert want[0] == got[0] and (want[1] == got[1] if want[0] == "err" else info_tuple(want[1]) == info_tuple(got[1])), (want, got)
    base = files[2]
    for cut in range(0, min(len(base), 700), 5):
        want, got = outcome(oracle.jpeg_info, base[:cut]), outcome(zg.jpeg.get_info, base[:cut])
        assert want[0] == got[0] and (want[1] == got[1] if want[0] == "err" else info_tuple(want[1]) == info_tuple(got[1])), cut


def test_host_decoder_under_random_corruption(oracle):
    """Byte flips and cuts of valid baseline and progressive files. For a progressive file jpeg.decode runs every scan's
    entropy decoder, so this compares the product's host Huffman / refinement code with the oracle's on damaged streams too."""
    rng = np.random.default_rng(31)
    img = J.test_image(40, 56, seed=1)
    bases = [J.pil_jpeg(img, quality=85, subsampling=2, progressive=True), J.pil_jpeg(img, quality=70, subsampling=0, progressive=True),
             J.pil_jpeg(img[..., 0], quality=85, progressive=True), J.pil_jpeg(img, quality=85, subsampling=1),
             J.pil_jpeg(img, quality=85, subsampling=2, progressive=True, restart_marker_blocks=2)]
    n_err = n_ok = 0
    for base in bases:
        for _ in range(250):
            data = bytearray(base)
            if rng.integers(0, 3) == 0:
                data = data[:int(rng.integers(0, len(data) + 1))]
            else:
                for _ in range(int(rng.integers(1, 4))):
                    data[int(rng.integers(0, len(data)))] = int(rng.integers(0, 256))
            want, got = probe_both(oracle, bytes(data), {})
            assert same(want, got), (want, got)
            n_err += want[0] == "err"
            n_ok += want[0] == "ok"
    assert n_err > 100 and n_ok > 100


def test_entropy_decoders_match_oracle_on_the_cpu(oracle):
    """zg_jpeg_coefficient_hash: decode + performBlockScan on the host, hashed. Baseline and progressive files from libjpeg, our
    own coefficient-level files (4:1:1, odd ids, restart intervals, 16-bit DQT, large values), cuts and flipped bytes: the same
    hash or the same error name as the oracle — the whole entropy layer compared without a GPU."""
    rng = np.random.default_rng(77)
    files = []
    for (h, w) in ((8, 8), (33, 47), (100, 37), (130, 258)):
        for smooth in (True, False):
            img = J.test_image(h, w, seed=h, smooth=smooth)
            for kw in (dict(subsampling=0), dict(subsampling=1), dict(subsampling=2), dict(subsampling=2, progressive=True), dict(subsampling=0, progressive=True, quality=40),
                       dict(subsampling=2, optimize=True, quality=97), dict(subsampling=1, restart_marker_blocks=2), dict(subsampling=2, progressive=True, restart_marker_rows=1)):
                files.append(J.pil_jpeg(img, **{"quality": 88, **kw}))
            files.append(J.pil_jpeg(img[..., 0], quality=75))
    for (lh, lv) in ((1, 1), (2, 1), (2, 2), (4, 1)):
        for ids in ((1, 2, 3), (0, 1, 2), (9, 1, 7)):
            comps = J.layout(lh, lv, ids)
            for ri in (0, 1, 3):
                files.append(J.write_baseline(50, 35, comps, J.FLAT_Q, J.random_coefficients(rng, comps, 50, 35), restart_interval=ri, dqt16=(ri == 3)))
    big = J.random_coefficients(rng, J.YCC, 48, 32, density=0.7, dc_range=1023, ac_range=1023)
    files.append(J.write_baseline(48, 32, J.YCC, {0: [255] * 64, 1: [65535] * 64}, big, dqt16=True))
    seen = set()
    n = 0
    for data in files:
        variants = [data]
        sos = data.index(b"\xFF\xDA")
        for _ in range(6):
            if rng.random() < 0.5:
                variants.append(data[:int(rng.integers(sos, len(data)))])
            else:
                bad = bytearray(data)
                bad[int(rng.integers(sos + 10, len(data) - 1))] = int(rng.integers(0, 256))
                variants.append(bytes(bad))
        for v in variants:
            want, got = outcome(oracle.jpeg_coefficient_hash, v), outcome(zg.jpeg.coefficient_hash, v)
            assert want == got, (len(v), want, got)
            seen.add(want[1] if want[0] == "err" else "ok")
            n += 1
    assert n > 700 and "ok" in seen and len(seen) >= 3, seen


def test_damaged_tables_found_by_the_host_layer_fuzz(oracle):
    """Two inputs tests/fuzz_host_layers.py turned up where the reference itself has no defined result (it asserts / divides by
    zero); both sides now return the same error. A Huffman table holding symbol 255 — the fast table's own "empty" mark — sends
    short codes down the slow path below min_code (jpeg.zig:1211, :1238); a zero sampling nibble makes the MCU size zero."""
    img = J.test_image(40, 56, seed=3, smooth=False)
    data = bytearray(J.pil_jpeg(img, quality=90, subsampling=2))
    at, hits = 2, 0
    while data[at + 1] != 0xDA:
        n = (data[at + 2] << 8) | data[at + 3]
        if data[at + 1] == 0xC4 and data[at + 4] == 0x10:  # AC table 0: give its second-shortest code the symbol 255
            data[at + 4 + 17 + 1] = 255
            hits += 1
        at += 2 + n
    assert hits == 1
    want, got = outcome(oracle.jpeg_coefficient_hash, bytes(data)), outcome(zg.jpeg.coefficient_hash, bytes(data))
    assert want == got == ("err", "InvalidHuffmanCode"), (want, got)

    gray = bytearray(J.pil_jpeg(img[..., 0], quality=75))
    sof = gray.index(b"\xFF\xC0")
    for nibbles in (0x01, 0x10, 0x00):
        gray[sof + 11] = nibbles
        want, got = outcome(oracle.jpeg_coefficient_hash, bytes(gray)), outcome(zg.jpeg.coefficient_hash, bytes(gray))
        assert want == got == ("err", "UnsupportedSamplingFactor"), (nibbles, want, got)


def test_storage_shortcuts_keep_the_reference_result(oracle):
    """The decoder stores subsampled chroma on its own grid and skips the clears and zero stores the reference repeats on
    memory that is already zero. The cases where that would show: a scan that names one component twice (its blocks are
    decoded twice: the second pass must start from a cleared block), a chroma component whose id is 1 (a scan of it alone
    walks the full grid), padded MCUs at the right / bottom edge, and files cut inside a block."""
    rng = np.random.default_rng(91)
    files = []
    for (h, w) in ((8, 8), (17, 50), (70, 33)):
        img = J.test_image(h, w, seed=w, smooth=False)
        for sub in (0, 1, 2):
            data = bytearray(J.pil_jpeg(img, quality=93, subsampling=sub))
            sos = data.index(b"\xFF\xDA")
            assert data[sos + 4] == 3
            for a, b in ((1, 0), (2, 0), (2, 1)):  # component b's id copied over component a's
                twice = bytearray(data)
                twice[sos + 5 + 2 * a] = twice[sos + 5 + 2 * b]
                files.append(bytes(twice))
            files.append(bytes(data))
    for (lh, lv) in ((2, 2), (2, 1), (4, 1), (1, 1)):
        for ids in ((9, 1, 7), (5, 6, 1), (1, 1, 1), (2, 2, 3)):
            comps = J.layout(lh, lv, ids)
            files.append(J.write_baseline(45, 52, comps, J.FLAT_Q, J.random_coefficients(rng, comps, 45, 52, density=0.5)))
    n = 0
    for data in files:
        sos = data.index(b"\xFF\xDA")
        for v in [data] + [data[:int(c)] for c in rng.integers(sos + 12, len(data), 5)]:
            want, got = outcome(oracle.jpeg_coefficient_hash, v), outcome(zg.jpeg.coefficient_hash, v)
            assert want == got, (len(v), want, got)
            n += want[0] == "ok"
    assert n > 150


def test_host_decoder_from_many_threads(oracle):
    """The coefficient arrays of the last decode stay with the calling thread and are cleared, not reallocated, for the next
    one: frames of different sizes and layouts, decoded in a different order by each of eight threads, must hash as they do alone."""
    from concurrent.futures import ThreadPoolExecutor
    files = []
    for i, (h, w) in enumerate(((16, 16), (200, 300), (90, 41), (301, 199), (64, 512))):
        img = J.test_image(h, w, seed=i, smooth=bool(i % 2))
        files += [J.pil_jpeg(img, quality=90, subsampling=i % 3), J.pil_jpeg(img, quality=60, subsampling=2, progressive=True), J.pil_jpeg(img[..., 0], quality=80)]
        files.append(files[-3][:len(files[-3]) * 2 // 3])  # a cut file: the tail of a reused array must read as zero
    want = [outcome(oracle.jpeg_coefficient_hash, f) for f in files]
    assert sum(w[0] == "ok" for w in want) >= 18

    def worker(seed):
        order = np.random.default_rng(seed).permutation(len(files) * 3) % len(files)
        return all(outcome(zg.jpeg.coefficient_hash, files[i]) == want[i] for i in order)
    with ThreadPoolExecutor(8) as pool:
        assert all(pool.map(worker, range(8)))


def random_blocks(rng, rows, cols, gray, subsampling, density):
    hm = 1 if gray or subsampling == 0 else 2
    vm = 2 if not gray and subsampling == 2 else 1
    n = -(-cols // (8 * hm)) * -(-rows // (8 * vm)) * (hm * vm + (0 if gray else 2))
    b = np.zeros((n, 64), np.int16)
    mask = rng.random((n, 64)) < density
    b[mask] = rng.integers(-1023, 1024, int(mask.sum()))
    b[:, 0] = rng.integers(-1024, 1024, n)
    b[rng.random(n) < 0.1, 1:] = 0  # DC-only blocks; runs past sixteen zeros come with the low densities
    return b


def test_encode_blocks_does_not_depend_on_the_thread_count(oracle, monkeypatch):
    """zg_jpeg_encode_blocks, the host half of encode: frames of 16 384 blocks and more are entropy-coded in bands of MCU rows
    on several threads and spliced at bit granularity. One thread is the plain serial coder (the one the GPU tests compare
    with the oracle's files byte for byte); every other thread count must give the same bytes, and files both decoders read."""
    rng = np.random.default_rng(12)
    n_banded = 0
    for (rows, cols) in ((1, 1), (8, 8), (640, 480), (1000, 1500), (777, 2049), (1536, 1536)):
        for gray in (False, True):
            for sub in ((0,) if gray else (0, 1, 2)):
                for density in (0.02, 0.4, 1.0):
                    blocks = random_blocks(rng, rows, cols, gray, sub, density)
                    opt = zg.jpeg.EncodeOptions(quality=80, subsampling=sub)
                    monkeypatch.setenv("ZIGNAL_HIP_HOST_THREADS", "1")
                    want = zg.jpeg.encode_blocks(blocks, rows, cols, gray, opt)
                    for threads in ("2", "5", "16", "64"):
                        monkeypatch.setenv("ZIGNAL_HIP_HOST_THREADS", threads)
                        assert zg.jpeg.encode_blocks(blocks, rows, cols, gray, opt) == want, (rows, cols, gray, sub, density, threads)
                    assert outcome(oracle.jpeg_coefficient_hash, want) == outcome(zg.jpeg.coefficient_hash, want)
                    n_banded += len(blocks) >= 16384
    assert n_banded >= 20
    with pytest.raises(ValueError):
        zg.jpeg.encode_blocks(np.zeros((3, 64), np.int16), 8, 8, True)
    with pytest.raises(zg.CodecError):
        zg.jpeg.encode_blocks(np.zeros((0, 64), np.int16), 0, 8, True)


def test_encode_blocks_writes_the_oracle_file(oracle, monkeypatch):
    """The same entry point against the oracle's encoder, byte for byte, on grey frames: the blocks are made here from the
    oracle's forward DCT (jpeg.zig:631-746), the quantiser restated in numpy (quantizeWithRecip :763-770) and the table the
    oracle's own file carries. 1024 x 1024 is 16 384 blocks: two bands on two threads; the small frame goes the serial way."""
    for (rows, cols, threads) in ((61, 83, "4"), (1024, 1024, "2"), (1024, 1030, "16")):
        img = J.test_image(rows, cols, seed=rows, smooth=False)[..., 0]
        for quality in (35, 90):
            want = oracle.jpeg_encode(img, quality=quality)
            dqt = want.index(b"\xFF\xDB")
            q = np.zeros(64, np.int64)
            q[J.ZIGZAG] = np.frombuffer(want[dqt + 5:dqt + 69], np.uint8)
            recip = np.round(16777216.0 / (q * 8.0)).astype(np.int64)
            by, bx = -(-rows // 8), -(-cols // 8)
            padded = np.pad(img.astype(np.int32), ((0, by * 8 - rows), (0, bx * 8 - cols)), mode="edge") - 128
            blocks = np.zeros((by * bx, 64), np.int16)
            for y in range(by):
                for x in range(bx):
                    d = oracle.jpeg_fdct8x8(padded[y * 8:y * 8 + 8, x * 8:x * 8 + 8]).reshape(64).astype(np.int64)
                    blocks[y * bx + x] = np.sign(d) * ((np.abs(d) * recip + (1 << 23)) >> 24)
            monkeypatch.setenv("ZIGNAL_HIP_HOST_THREADS", threads)
            got = zg.jpeg.encode_blocks(blocks, rows, cols, True, zg.jpeg.EncodeOptions(quality=quality))
            assert got == want, (rows, cols, quality, len(got), len(want))
    # colour: everything up to the first entropy-coded byte is a function of the options alone
    rng = np.random.default_rng(2)
    rgb = J.test_image(37, 50, seed=5)
    for sub in (0, 1, 2):
        for kw in (dict(quality=77), dict(quality=12, density_dpi=300, comment=b"made by the test")):
            want = oracle.jpeg_encode(rgb, subsampling=sub, **kw)
            got = zg.jpeg.encode_blocks(random_blocks(rng, 37, 50, False, sub, 0.2), 37, 50, False, zg.jpeg.EncodeOptions(subsampling=sub, **kw))
            head = want.index(b"\xFF\xDA") + 14
            assert got[:head] == want[:head] and got[-2:] == b"\xFF\xD9", (sub, kw)


# ---- GPU ---------------------------------------------------------------------------------------------------------------------

def decode_both(oracle, data, kind=None, limits=None):
    want = outcome(lambda: oracle.jpeg_decode_native(data, oracle.jpeg_limits(**limits) if limits else None))
    got = outcome(lambda: zg.jpeg.load_from_bytes(data, kind, zg.jpeg.decode_limits(**limits) if limits else None, return_scan_limit_reached=True))
    return want, got


def assert_same_decode(oracle, data, what, limits=None):
    want, got = decode_both(oracle, data, limits=limits)
    assert want[0] == got[0], (what, want[0], got[0], want[1] if want[0] == "err" else "", got[1] if got[0] == "err" else "")
    if want[0] == "err":
        assert want[1] == got[1], (what, want, got)
        return want[1]
    (wimg, _, whit), (gimg, ghit) = want[1], got[1]
    g = gimg.to_numpy()
    assert g.shape == wimg.shape and whit == ghit, what
    if not np.array_equal(g, wimg):
        bad = np.argwhere(g != wimg)
        raise AssertionError(f"{what}: {len(bad)} differing samples, first at {bad[0].tolist()}: got {g[tuple(bad[0])]} want {wimg[tuple(bad[0])]}")
    return "ok"


PIL_CASES = [dict(subsampling=0), dict(subsampling=1), dict(subsampling=2), dict(subsampling=0, progressive=True), dict(subsampling=1, progressive=True),
             dict(subsampling=2, progressive=True), dict(subsampling=2, optimize=True), dict(subsampling=2, quality=35), dict(subsampling=0, quality=100),
             dict(subsampling=2, restart_marker_blocks=3), dict(subsampling=0, restart_marker_rows=1), dict(subsampling=1, progressive=True, restart_marker_blocks=5),
             dict(subsampling=2, progressive=True, quality=60)]


@pytest.mark.gpu
@pytest.mark.parametrize("kw", PIL_CASES, ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
def test_decode_parity_pillow_files(oracle, kw):
    for (h, w) in ((1, 1), (8, 8), (9, 7), (16, 16), (33, 47), (64, 80), (100, 37), (130, 258)):
        for smooth in (True, False):
            img = J.test_image(h, w, seed=h + w, smooth=smooth)
            assert_same_decode(oracle, J.pil_jpeg(img, **{"quality": 90, **kw}), (kw, h, w, smooth))
        gkw = {k: v for k, v in {"quality": 90, **kw}.items() if k != "subsampling"}
        assert_same_decode(oracle, J.pil_jpeg(J.test_image(h, w, seed=3)[..., 1], **gkw), ("grey", kw, h, w))


@pytest.mark.gpu
def test_decode_parity_coefficient_level_files(oracle):
    """4:1:1, odd component ids (which change what the reference takes for a non-interleaved scan), 16-bit DQT, restart
    intervals of every length, large coefficients (i32 wrap-around in the IDCT is part of the contract)."""
    rng = np.random.default_rng(14)
    outcomes = set()
    for (lh, lv) in ((1, 1), (2, 1), (2, 2), (4, 1)):
        for (h, w) in ((8, 8), (17, 50), (40, 33), (70, 130)):
            for ids in ((1, 2, 3), (0, 1, 2), (7, 9, 200)):
                comps = J.layout(lh, lv, ids)
                co = J.random_coefficients(rng, comps, w, h)
                for ri in (0, 1, 2, 5):
                    data = J.write_baseline(w, h, comps, J.FLAT_Q, co, restart_interval=ri, dqt16=(ri == 5))
                    outcomes.add(assert_same_decode(oracle, data, (lh, lv, h, w, ids, ri)))
    big = J.random_coefficients(rng, J.YCC, 48, 32, density=0.6, dc_range=1023, ac_range=1023)
    wide_q = {0: [255] * 64, 1: [65535] * 64}
    outcomes.add(assert_same_decode(oracle, J.write_baseline(48, 32, J.YCC, wide_q, big, dqt16=True), "wrap-around"))
    g = [(1, 1, 1, 0, 0, 0)]
    for ri in (0, 4):
        outcomes.add(assert_same_decode(oracle, J.write_baseline(30, 20, g, J.FLAT_Q, J.random_coefficients(rng, g, 30, 20), restart_interval=ri), ("grey", ri)))
    assert "ok" in outcomes


@pytest.mark.gpu
def test_decode_known_answers_limits_and_cuts(oracle):
    for data, limits in structural_cases():
        assert_same_decode(oracle, data, "structural", limits or None)
    img, hit = zg.jpeg.load_from_bytes(PROGRESSIVE, return_scan_limit_reached=True)  # jpeg.zig:3069-3116
    assert (img.to_numpy() == 143).all() and not hit
    img, hit = zg.jpeg.load_from_bytes(PROGRESSIVE, limits=zg.jpeg.decode_limits(max_scans=2), return_scan_limit_reached=True)
    assert (img.to_numpy() == 142).all() and hit
    assert (zg.jpeg.load_from_bytes(PROGRESSIVE[:-4]).to_numpy() == 142).all()
    assert (zg.jpeg.load_from_bytes(SIG + DQT + SOF2 + DHT + SCAN1[:-1]).to_numpy() == 128).all()
    assert (zg.jpeg.load_from_bytes(SIG + DQT + SOF2 + DHT + SCAN1 + EOI).to_numpy() == 140).all()
    rng = np.random.default_rng(6)
    pic = J.test_image(48, 64, seed=5)
    seen = set()
    for kw in (dict(subsampling=2), dict(subsampling=2, progressive=True), dict(subsampling=0, progressive=True, quality=60), dict(subsampling=1, optimize=True)):
        data = J.pil_jpeg(pic, **{"quality": 88, **kw})
        for cut in [len(data) - 1, len(data) - 2, len(data) - 3] + [int(c) for c in rng.integers(20, len(data), 25)]:
            seen.add(assert_same_decode(oracle, data[:cut], (kw, cut)))
        for _ in range(25):  # damage inside the entropy data
            bad = bytearray(data)
            at = int(rng.integers(data.index(b"\xFF\xDA") + 14, len(data) - 2))
            bad[at] = int(rng.integers(0, 256))
            seen.add(assert_same_decode(oracle, bytes(bad), (kw, "flip", at)))
    assert "ok" in seen and len(seen) >= 2


@pytest.mark.gpu
def test_decode_targets(oracle):
    import torch
    data = J.pil_jpeg(J.test_image(37, 53, seed=2), quality=90, subsampling=2)
    native = oracle.jpeg_decode_native(data)[0]
    for kind in ("u8", "rgb_u8", "rgba_u8"):  # loadFromBytes(T): Image.convert of the native image
        assert np.array_equal(zg.jpeg.load_from_bytes(data, kind).to_numpy(), oracle.jpeg_load(data, kind)), kind
    gdata = J.pil_jpeg(J.test_image(37, 53, seed=2)[..., 0], quality=90)
    for kind in ("u8", "rgb_u8", "rgba_u8"):
        assert np.array_equal(zg.jpeg.load_from_bytes(gdata, kind).to_numpy(), oracle.jpeg_load(gdata, kind)), kind
    host = zg.jpeg.load_from_bytes(data, device=None)  # zg_jpeg_decode_host
    assert not host.on_device and np.array_equal(host.data, native)
    big = zg.Image(torch.zeros((60, 80, 3), dtype=torch.uint8, device="cuda"))  # a strided destination
    view = big.view((9, 5, 9 + 53, 5 + 37))
    buf = (zg._lib.C.c_uint8 * len(data)).from_buffer_copy(data)
    d = view._desc()
    zg._lib.check(zg.lib().zg_jpeg_decode(buf, len(data), None, zg._lib.C.byref(d), zg.CS_RGB, None, view._stream()))
    torch.cuda.synchronize()
    out = big.to_numpy()
    assert np.array_equal(out[5:42, 9:62], native)
    out[5:42, 9:62] = 0
    assert not out.any()
    f32 = zg.Image(torch.zeros((37, 53, 3), dtype=torch.float32, device="cuda"))
    d = f32._desc()
    zg._lib.check(zg.lib().zg_jpeg_decode(buf, len(data), None, zg._lib.C.byref(d), zg.CS_RGB, None, f32._stream()))
    torch.cuda.synchronize()
    assert np.array_equal(f32.to_numpy().view(np.uint32), oracle.convert(native, oracle.CS_RGB, oracle.CS_RGB, np.float32, 3).view(np.uint32))
    with pytest.raises(zg.DimensionMismatch):
        bad = zg.Image(torch.zeros((37, 52, 3), dtype=torch.uint8, device="cuda"))
        d = bad._desc()
        zg._lib.check(zg.lib().zg_jpeg_decode(buf, len(data), None, zg._lib.C.byref(d), zg.CS_RGB, None, bad._stream()))
    assert np.array_equal(zg.Image.load_from_bytes(data).to_numpy(), native)  # format detection by signature


@pytest.mark.gpu
def test_decode_large_frame(oracle):
    img = J.test_image(1080, 1920, seed=9)
    for kw in (dict(subsampling=2), dict(subsampling=0, progressive=True)):
        assert_same_decode(oracle, J.pil_jpeg(img, **{"quality": 85, **kw}), ("1080p", kw))


# ---- encoder ---------------------------------------------------------------------------------------------------------------

def test_oracle_encoder_reference_roundtrips(oracle):
    """The reference's own encode -> decode tests (jpeg.zig:2860-3026) through the oracle: same images, options and PSNR bars."""
    def psnr(a, b):
        return 10 * np.log10(255.0 ** 2 / ((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean())

    def grad(rows, cols, blue):
        y, x = np.mgrid[0:rows, 0:cols]
        return np.stack([(x * 255) // (cols - 1), (y * 255) // (rows - 1), blue(x, y)], -1).astype(np.uint8)
    cases = [(grad(16, 16, lambda x, y: ((x + y) * 255) // 30), dict(quality=85), 40.0),
             (grad(19, 25, lambda x, y: ((x * y) * 255) // (24 * 18)), dict(quality=85, subsampling=1), 40.0),
             (grad(64, 48, lambda x, y: ((x + 2 * y) * 255) // (47 + 2 * 63)), dict(quality=92, subsampling=2), 45.0),
             (grad(37, 53, lambda x, y: ((2 * x + 3 * y) * 255) // (2 * 52 + 3 * 36)), dict(quality=85, subsampling=2), 35.0)]
    for img, kw, bar in cases:
        data = oracle.jpeg_encode(img, **kw)
        assert psnr(img, oracle.jpeg_decode_native(data)[0]) > bar
        assert psnr(img, J.pil_decode(data)) > bar - 3  # and libjpeg reads the file
    yy, xx = np.mgrid[0:16, 0:16]
    g = (((xx + yy) * 255) // 30).astype(np.uint8)
    data = oracle.jpeg_encode(g, quality=85)
    assert psnr(g, oracle.jpeg_decode_native(data)[0]) > 45 and oracle.jpeg_info(data).num_components == 1
    with pytest.raises(oracle.JpegError) as e:
        oracle.jpeg_encode(np.zeros((0, 4), np.uint8))
    assert e.value.name == "InvalidImageDimensions"
    # the forward DCT is the scaled-by-8 DCT-II to within rounding
    rng = np.random.default_rng(1)
    k = np.arange(8)
    c = np.where(k == 0, np.sqrt(0.5), 1.0)
    basis = 0.5 * c[None, :] * np.cos((2 * k[:, None] + 1) * k[None, :] * np.pi / 16)
    blk = rng.integers(-128, 128, (8, 8))
    assert np.abs(oracle.jpeg_fdct8x8(blk) - 8 * (basis.T @ blk @ basis)).max() <= 2.0


@pytest.mark.gpu
def test_encode_files_byte_for_byte(oracle):
    import torch
    rng = np.random.default_rng(12)
    for (h, w) in ((1, 1), (7, 9), (8, 8), (16, 16), (33, 47), (64, 80), (100, 37), (130, 258)):
        for smooth in (True, False):
            img = J.test_image(h, w, seed=h * w, smooth=smooth)
            for sub in (0, 1, 2):
                for q in (int(rng.integers(1, 101)), 90):
                    want = oracle.jpeg_encode(img, q, sub)
                    got = zg.jpeg.encode(zg.Image(torch.from_numpy(img).cuda()), zg.jpeg.EncodeOptions(quality=q, subsampling=sub))
                    assert got == want, (h, w, smooth, sub, q, len(got), len(want))
            want = oracle.jpeg_encode(img[..., 1], 77)
            got = zg.jpeg.encode(zg.Image(torch.from_numpy(img[..., 1].copy()).cuda()), zg.jpeg.EncodeOptions(quality=77))
            assert got == want, ("grey", h, w, smooth)
    img = J.test_image(45, 61, seed=8)
    dev = zg.Image(torch.from_numpy(img).cuda())
    opts = zg.jpeg.EncodeOptions(quality=60, subsampling=1, density_dpi=300, comment=b"made on an MI355X")
    want = oracle.jpeg_encode(img, 60, 1, 300, b"made on an MI355X")
    assert zg.jpeg.encode(dev, opts) == want and b"made on an MI355X" in want
    assert zg.jpeg.encode(zg.Image(img), opts) == want                      # zg_jpeg_encode_host
    big = zg.Image(torch.zeros((70, 90, 3), dtype=torch.uint8, device="cuda"))  # a view: stride != cols
    big.data[11:56, 13:74] = torch.from_numpy(img).cuda()
    assert zg.jpeg.encode(big.view((13, 11, 74, 56)), opts) == want
    # any other T goes through Image.convert(Rgb): Rgba(u8), Rgb(f32), f32
    rgba = rng.integers(0, 256, (20, 31, 4), dtype=np.uint8)
    assert zg.jpeg.encode(zg.Image(torch.from_numpy(rgba).cuda())) == oracle.jpeg_encode(rgba)
    f = rng.random((20, 31, 3), dtype=np.float32)
    assert zg.jpeg.encode(zg.Image(torch.from_numpy(f).cuda())) == oracle.jpeg_encode(f)
    g = rng.random((20, 31), dtype=np.float32)
    assert zg.jpeg.encode(zg.Image(torch.from_numpy(g).cuda())) == oracle.jpeg_encode(g)
    # round trip through our own decoder and a file on disk
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "x.JPG")
        dev.save(path)
        assert open(path, "rb").read() == oracle.jpeg_encode(img, 90, 2)
        assert np.array_equal(zg.Image.load(path).to_numpy(), oracle.jpeg_decode_native(oracle.jpeg_encode(img, 90, 2))[0])
    with pytest.raises(zg.CodecError) as e:
        zg.jpeg.encode(zg.Image(torch.zeros((0, 5, 3), dtype=torch.uint8, device="cuda")))
    assert e.value.name == "InvalidImageDimensions"


def test_header_beyond_the_limits_sizes_no_allocation():
    """A 20-byte header declaring 65535 x 65535 x 3 used to force a 12.9 GB zero-filled image before the decode refused it
    (jpeg.zig's limits run on the header, :19-33): the mirror now looks at the effective limits before it allocates."""
    import struct
    import tracemalloc

    data = bytearray(J.pil_jpeg(J.test_image(16, 16)))
    at = data.index(b"\xff\xc0")
    data[at + 5:at + 9] = struct.pack(">HH", 65535, 65535)
    tracemalloc.start()
    try:
        with pytest.raises(Exception):
            zg.jpeg.load_from_bytes(bytes(data), device=None)  # ImageTooLarge on a GPU box; no device at all here
        peak = tracemalloc.get_traced_memory()[1]
    finally:
        tracemalloc.stop()
    assert peak < 64 << 20, peak
