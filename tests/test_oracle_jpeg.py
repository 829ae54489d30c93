"""The JPEG oracle (oracle/jpeg.c) pinned to the reference's own tests (src/codecs/jpeg.zig:181-258, 3028-3174) and
cross-checked against an independent decoder (Pillow / libjpeg) and an f64 IDCT. CPU only."""
import struct

import numpy as np
import pytest

from tests import jpeg_util as J

SIG = bytes([0xFF, 0xD8])
DQT = bytes([0xFF, 0xDB, 0x00, 0x43, 0x00]) + bytes([8] * 64)                                  # jpeg.zig:3055
SOF2 = bytes([0xFF, 0xC2, 0x00, 0x0B, 0x08, 0x00, 0x08, 0x00, 0x08, 0x01, 0x01, 0x11, 0x00])  # :3056
DHT = bytes([0xFF, 0xC4, 0x00, 0x14, 0x00, 0x01]) + bytes(15) + bytes([0x02])                  # :3057
SCAN1 = bytes([0xFF, 0xDA, 0x00, 0x08, 0x01, 0x01, 0x00, 0x00, 0x00, 0x02, 0x7F])              # :3059
SCAN2 = bytes([0xFF, 0xDA, 0x00, 0x08, 0x01, 0x01, 0x00, 0x00, 0x00, 0x21, 0xFF, 0x00])        # :3061
SCAN3 = bytes([0xFF, 0xDA, 0x00, 0x08, 0x01, 0x01, 0x00, 0x00, 0x00, 0x10, 0xFF, 0x00])        # :3063
EOI = bytes([0xFF, 0xD9])
PROGRESSIVE = SIG + DQT + SOF2 + DHT + SCAN1 + SCAN2 + SCAN3 + EOI


def expect_error(oracle, name, fn, *a, **kw):
    with pytest.raises(oracle.JpegError) as e:
        fn(*a, **kw)
    assert e.value.name == name, f"expected error.{name}, got error.{e.value.name}"


def test_reference_known_answers(oracle):  # jpeg.zig:3028-3116
    L = oracle.jpeg_limits
    expect_error(oracle, "JpegDataTooLarge", oracle.jpeg_decode_state, SIG, L(max_jpeg_bytes=1))
    expect_error(oracle, "MarkerDataLimitExceeded", oracle.jpeg_decode_state,
                 bytes([0xFF, 0xD8, 0xFF, 0xE0, 0x00, 0x04, 0x00, 0x00, 0xFF, 0xD9]), L(max_jpeg_bytes=0, max_marker_bytes=2))
    sof0_16 = bytes([0xFF, 0xC0, 0x00, 0x0B, 0x08, 0x00, 0x10, 0x00, 0x10, 0x01, 0x01, 0x11, 0x00])
    expect_error(oracle, "BlockMemoryLimitExceeded", oracle.jpeg_decode_state, SIG + sof0_16 + EOI, L(max_blocks=1))
    img, h, hit = oracle.jpeg_decode_native(PROGRESSIVE)
    assert img.shape == (8, 8) and (img == 143).all() and h.progressive and not hit
    img, _, hit = oracle.jpeg_decode_native(PROGRESSIVE, L(max_scans=2))
    assert (img == 142).all() and hit
    sof0 = bytes([0xFF, 0xC0]) + SOF2[2:]
    expect_error(oracle, "DuplicateSOF", oracle.jpeg_decode_state, SIG + sof0 + sof0 + EOI)
    assert (oracle.jpeg_decode_native(PROGRESSIVE[:-4])[0] == 142).all()
    assert (oracle.jpeg_decode_native(SIG + DQT + SOF2 + DHT + SCAN1[:-1])[0] == 128).all()
    assert (oracle.jpeg_decode_native(SIG + DQT + SOF2 + DHT + SCAN1 + EOI)[0] == 140).all()


def test_get_info_known_answers(oracle):  # jpeg.zig:181-258
    for sub, code in ((0, 0), (1, 1), (2, 2)):
        h = oracle.jpeg_info(J.pil_jpeg(J.test_image(20, 30), subsampling=sub))
        assert (h.width, h.height, h.num_components, h.precision, h.progressive, h.subsampling) == (30, 20, 3, 8, 0, code)
    h = oracle.jpeg_info(J.pil_jpeg(J.test_image(9, 5)[..., 0], progressive=True))
    assert (h.width, h.height, h.num_components, h.progressive, h.subsampling) == (5, 9, 1, 1, -1)
    expect_error(oracle, "InvalidJpegFile", oracle.jpeg_info, b"\x89PNG")
    expect_error(oracle, "MissingSOF", oracle.jpeg_info, SIG + EOI)
    expect_error(oracle, "EndOfStream", oracle.jpeg_info, SIG + DQT)
    expect_error(oracle, "InvalidMarker", oracle.jpeg_info, SIG + bytes([0xFF, 0xE0, 0x00, 0x01]))
    expect_error(oracle, "InvalidSOF", oracle.jpeg_info, SIG + bytes([0xFF, 0xC0, 0x00, 0x05, 8, 0, 1]))


def test_structural_errors(oracle):  # decode (:2035-2151), parseSOF / DHT / DQT / SOS / DRI (:1314-1645)
    dec = oracle.jpeg_decode_state

    def sof(marker=0xC0, precision=8, h=8, w=8, comps=((1, 0x11, 0),)):
        body = struct.pack(">BHHB", precision, h, w, len(comps)) + b"".join(bytes(c) for c in comps)
        return bytes([0xFF, marker]) + struct.pack(">H", 2 + len(body)) + body

    expect_error(oracle, "InvalidJpegFile", dec, b"")
    expect_error(oracle, "InvalidJpegFile", dec, b"\x89PNG\r\n")
    expect_error(oracle, "NoScanData", dec, SIG + EOI)
    expect_error(oracle, "NoScanData", dec, SIG + sof() + EOI)
    expect_error(oracle, "InvalidMarker", dec, SIG + b"\x00\x00\x00")
    expect_error(oracle, "UnsupportedExtendedSequential", dec, SIG + sof(0xC1))
    expect_error(oracle, "UnsupportedLosslessJpeg", dec, SIG + sof(0xC3))
    expect_error(oracle, "UnsupportedArithmeticCoding", dec, SIG + bytes([0xFF, 0xCC, 0, 2]))
    expect_error(oracle, "UnsupportedHierarchicalJpeg", dec, SIG + bytes([0xFF, 0xDE, 0, 2]))
    expect_error(oracle, "UnsupportedJpegVariant", dec, SIG + bytes([0xFF, 0xDC, 0, 2]))
    expect_error(oracle, "Unsupported12BitPrecision", dec, SIG + sof(precision=12))
    expect_error(oracle, "Unsupported16BitPrecision", dec, SIG + sof(precision=16))
    expect_error(oracle, "UnsupportedPrecision", dec, SIG + sof(precision=9))
    expect_error(oracle, "InvalidSOF", dec, SIG + sof(h=0))
    expect_error(oracle, "InvalidSOF", dec, SIG + bytes([0xFF, 0xC0, 0, 5, 8, 0, 8]))
    expect_error(oracle, "ImageTooLarge", dec, SIG + sof(h=9000))
    expect_error(oracle, "UnsupportedComponentCount", dec, SIG + sof(comps=((1, 0x11, 0),) * 4))
    expect_error(oracle, "InvalidComponentCount", dec, SIG + sof(comps=((1, 0x11, 0),) * 2))
    expect_error(oracle, "InvalidComponentCount", dec, SIG + sof(comps=()))
    expect_error(oracle, "InvalidComponentCount", dec, SIG + sof(comps=((1, 0x22, 0), (2, 0x11, 1), (3, 0x12, 1))))
    expect_error(oracle, "UnsupportedSamplingFactor", dec, SIG + sof(comps=((1, 0x12, 0), (2, 0x11, 1), (3, 0x11, 1))))
    expect_error(oracle, "UnsupportedSamplingFactor", dec, SIG + sof(comps=((1, 0x51, 0),)))
    expect_error(oracle, "ImageTooLarge", dec, SIG + sof(h=100, w=100), oracle.jpeg_limits(max_pixels=9999))
    expect_error(oracle, "InvalidDHT", dec, SIG + bytes([0xFF, 0xC4, 0, 2]))
    expect_error(oracle, "InvalidDHT", dec, SIG + bytes([0xFF, 0xC4, 0, 5, 0, 1, 2]))
    expect_error(oracle, "InvalidHuffmanTable", dec, SIG + bytes([0xFF, 0xC4, 0, 19, 0]) + bytes([255] * 16))
    expect_error(oracle, "InvalidHuffmanTable", dec, SIG + bytes([0xFF, 0xC4, 0, 22, 0, 3]) + bytes(15) + bytes([1, 2, 3]))  # a code of all ones
    expect_error(oracle, "InvalidDQT", dec, SIG + bytes([0xFF, 0xDB, 0, 2]))
    expect_error(oracle, "InvalidDQT", dec, SIG + bytes([0xFF, 0xDB, 0, 10, 0]) + bytes(7))
    expect_error(oracle, "InvalidDRI", dec, SIG + bytes([0xFF, 0xDD, 0, 5, 0, 0, 0]))
    expect_error(oracle, "InvalidSOS", dec, SIG + sof() + bytes([0xFF, 0xDA, 0, 8, 1, 1, 0, 1, 63, 0]))     # baseline must be 0..63, 0
    expect_error(oracle, "InvalidSOS", dec, SIG + sof() + bytes([0xFF, 0xDA, 0, 8, 2, 1, 0, 0, 63, 0]))     # component count
    expect_error(oracle, "InvalidSOS", dec, SIG + sof(0xC2) + bytes([0xFF, 0xDA, 0, 8, 1, 1, 0, 0, 5, 0]))  # DC mixed with AC
    expect_error(oracle, "InvalidSOS", dec, SIG + sof(0xC2) + bytes([0xFF, 0xDA, 0, 8, 1, 1, 0, 9, 5, 0]))
    expect_error(oracle, "InvalidMarker", dec, SIG + bytes([0xFF, 0xDB, 0, 1]))
    expect_error(oracle, "InvalidMarker", dec, SIG + bytes([0xFF, 0xDB, 0, 200, 0]))
    expect_error(oracle, "UnexpectedEndOfData", dec, SIG + bytes([0xFF, 0xDB, 0]))
    # a baseline stream stops being parsed at its first SOS; what follows is only looked at by the block scan
    base = J.pil_jpeg(J.test_image(16, 16)[..., 0], quality=80)
    h, hit = dec(base)
    assert (h.width, h.height, h.num_components, h.progressive) == (16, 16, 1, 0) and not hit
    no_dht = b"".join(bytes([0xFF, m]) + struct.pack(">H", len(p) + 2) + p for m, p in J.segments(base) if m != 0xC4)
    ent = base[base.index(b"\xFF\xDA"):]
    expect_error(oracle, "MissingHuffmanTable", oracle.jpeg_decode_native, SIG + no_dht + ent[2 + struct.unpack(">H", ent[2:4])[0]:])
    no_dqt = b"".join(bytes([0xFF, m]) + struct.pack(">H", len(p) + 2) + p for m, p in J.segments(base) if m != 0xDB)
    expect_error(oracle, "MissingQuantTable", oracle.jpeg_decode_native, SIG + no_dqt + ent[2 + struct.unpack(">H", ent[2:4])[0]:])


def test_idct_against_f64(oracle):
    """The integer IDCT is within one level of the exact transform (and exact for flat blocks)."""
    rng = np.random.default_rng(2)
    k = np.arange(8)
    c = np.where(k == 0, np.sqrt(0.5), 1.0)
    basis = 0.5 * c[None, :] * np.cos((2 * k[:, None] + 1) * k[None, :] * np.pi / 16)  # [x, u]
    worst = 0.0
    for _ in range(300):
        blk = np.zeros((8, 8))
        n = int(rng.integers(1, 12))
        blk.flat[rng.integers(0, 64, n)] = rng.integers(-300, 301, n)
        blk[0, 0] = rng.integers(-1000, 1001)
        exact = basis @ blk @ basis.T
        got = oracle.jpeg_idct8x8(blk.astype(np.int32))
        worst = max(worst, float(np.abs(got - exact).max()))
    assert worst <= 1.0 + 1e-9
    flat = np.zeros((8, 8), np.int32)
    flat[0, 0] = 123
    assert (oracle.jpeg_idct8x8(flat) == (123 + 4) >> 3).all()


PIL_CASES = [dict(subsampling=0), dict(subsampling=1), dict(subsampling=2), dict(subsampling=0, progressive=True), dict(subsampling=1, progressive=True),
             dict(subsampling=2, progressive=True), dict(subsampling=2, optimize=True), dict(subsampling=2, quality=35), dict(subsampling=0, quality=100),
             dict(subsampling=2, progressive=True, quality=60)]  # restart intervals: see test_restart_intervals


@pytest.mark.parametrize("kw", PIL_CASES, ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
def test_against_libjpeg(oracle, kw):
    """Same files through Pillow: 4:4:4 and grey agree to the IDCT's last level; with subsampled chroma the reference's
    upsampler (taps confined to the MCU's own chroma block) differs from libjpeg's only at block seams, so the mean stays small."""
    for (h, w) in ((8, 8), (16, 16), (33, 47), (64, 80), (100, 37)):
        img = J.test_image(h, w, seed=h)
        data = J.pil_jpeg(img, **{"quality": 90, **kw})
        got, header, _ = oracle.jpeg_decode_native(data)
        ref = J.pil_decode(data).astype(np.int32)
        assert got.shape == ref.shape == (h, w, 3) and header.progressive == int(bool(kw.get("progressive")))
        diff = np.abs(got.astype(np.int32) - ref)
        if kw["subsampling"] == 0:
            assert diff.max() <= 3, (kw, h, w, diff.max())
        else:
            assert diff.mean() < 2.5 and np.percentile(diff, 99) <= 16, (kw, h, w, diff.mean(), diff.max())
        gdata = J.pil_jpeg(img[..., 1], **{k: v for k, v in {"quality": 90, **kw}.items() if k != "subsampling"})
        ggot, gh, _ = oracle.jpeg_decode_native(gdata)
        assert gh.num_components == 1 and np.abs(ggot.astype(np.int32) - J.pil_decode(gdata)).max() <= 1


def test_progressive_equals_baseline_coefficients(oracle):
    """libjpeg writes the same quantised coefficients whether the file is baseline or progressive: decoding must agree exactly."""
    for sub in (0, 1, 2):
        for (h, w) in ((24, 40), (61, 35)):
            img = J.test_image(h, w, seed=sub)
            a, _, _ = oracle.jpeg_decode_native(J.pil_jpeg(img, quality=85, subsampling=sub))
            b, _, _ = oracle.jpeg_decode_native(J.pil_jpeg(img, quality=85, subsampling=sub, progressive=True))
            assert np.array_equal(a, b), (sub, h, w)


def test_coefficient_level_files(oracle):
    """Our own baseline writer: 4:1:1, component ids other than 1 2 3, 16-bit DQT, restart intervals, grey — against Pillow."""
    rng = np.random.default_rng(4)
    for (lh, lv) in ((1, 1), (2, 1), (2, 2), (4, 1)):
        for (h, w) in ((8, 8), (17, 50), (40, 33)):
            comps = J.layout(lh, lv)
            co = J.random_coefficients(rng, comps, w, h)
            for ri in (0, 1, 3):
                data = J.write_baseline(w, h, comps, J.FLAT_Q, co, restart_interval=ri, dqt16=(ri == 3))
                ref = J.pil_decode(data).astype(np.int32)
                if ri == 0:
                    got, hd, _ = oracle.jpeg_decode_native(data)
                    diff = np.abs(got.astype(np.int32) - ref)
                    assert got.shape == (h, w, 3) and (diff.max() <= 3 if (lh, lv) == (1, 1) else diff.mean() < 6), (lh, lv, h, w, diff.mean(), diff.max())
                else:  # see test_restart_intervals for what the reference makes of these: an image, or an entropy error
                    try:
                        assert oracle.jpeg_decode_native(data)[0].shape == (h, w, 3)
                    except oracle.JpegError as e:
                        assert e.name in ("InvalidHuffmanCode", "InvalidACCoefficient", "InvalidDCCoefficient")
    g = [(1, 1, 1, 0, 0, 0)]
    co = J.random_coefficients(rng, g, 30, 20)
    data = J.write_baseline(30, 20, g, J.FLAT_Q, co)
    assert np.abs(oracle.jpeg_decode_native(data)[0].astype(np.int32) - J.pil_decode(data)).max() <= 1


def test_restart_intervals(oracle):
    """Restart markers are swallowed by the reference's bit filler (jpeg.zig:1694-1711) and a restart boundary then discards
    the pre-fetched bits (:2428-2436): when the filler has already read past the marker, bytes of the next interval are lost.
    The oracle restates exactly that; files whose intervals end on a long enough run of padding decode like libjpeg, others
    do not, and both behaviours are the reference's."""
    outcomes = set()
    for seed in range(6):
        img = J.test_image(32, 48, seed=seed)
        clean = oracle.jpeg_decode_native(J.pil_jpeg(img[..., 0], quality=90))[0]
        try:
            with_rst = oracle.jpeg_decode_native(J.pil_jpeg(img[..., 0], quality=90, restart_marker_blocks=2))[0]
        except oracle.JpegError as e:
            assert e.name in ("InvalidHuffmanCode", "InvalidACCoefficient", "InvalidDCCoefficient")
            outcomes.add("error")
            continue
        outcomes.add("image")
        # the first interval never crosses a marker: its blocks are identical with and without restart markers
        assert clean.shape == with_rst.shape and np.array_equal(clean[:8, :16], with_rst[:8, :16])
    assert outcomes


def test_truncation_keeps_decoded_blocks(oracle):
    img = J.test_image(48, 64, seed=5)
    for kw in (dict(), dict(progressive=True)):
        data = J.pil_jpeg(img, quality=88, subsampling=2, **kw)
        full, _, _ = oracle.jpeg_decode_native(data)
        sos = data.index(b"\xFF\xDA")
        for cut in (len(data) - 2, len(data) - 40, sos + (len(data) - sos) // 2, sos + 30):
            part, h, _ = oracle.jpeg_decode_native(data[:cut])
            assert part.shape == full.shape
            if not kw:
                rows_ok = (part == full).all(axis=(1, 2))
                assert rows_ok[:8].all() or cut < sos + 200  # the head of the image survives a cut in the tail


def test_load_conversions(oracle):  # loadFromBytes(T) (:2825-2851)
    data = J.pil_jpeg(J.test_image(20, 28), quality=90)
    native = oracle.jpeg_decode_native(data)[0]
    assert np.array_equal(oracle.jpeg_load(data, "rgb_u8"), native)
    rgba = oracle.jpeg_load(data, "rgba_u8")
    assert np.array_equal(rgba[..., :3], native) and (rgba[..., 3] == 255).all()
    assert np.array_equal(oracle.jpeg_load(data, "u8"), oracle.convert(native, oracle.CS_RGB, oracle.CS_GRAY, np.uint8, 1))
    gdata = J.pil_jpeg(J.test_image(20, 28)[..., 0])
    g = oracle.jpeg_decode_native(gdata)[0]
    assert np.array_equal(oracle.jpeg_load(gdata, "rgb_u8"), np.stack([g, g, g], -1))


def test_bit_reader_and_ycbcr_known_answers(oracle):  # jpeg.zig:3132-3149, :3151-3174; markers :3119-3130
    assert oracle.jpeg_get_bits(bytes([0b10110011, 0b01010101]), [4, 4, 8]) == [0b1011, 0b0011, 0b01010101]
    assert oracle.jpeg_get_bits(bytes([0xFF, 0x00, 0x0F]), [8, 8]) == [0xFF, 0x0F]          # a stuffed 0xFF is one data byte
    assert oracle.jpeg_get_bits(bytes([0xAB, 0xFF, 0xD3, 0xCD]), [8, 8]) == [0xAB, 0xCD]    # an RSTn marker is swallowed
    assert oracle.jpeg_get_bits(bytes([0xAB, 0xFF, 0xD9]), [8, 8]) == [0xAB]                # any other marker ends the data
    assert oracle.jpeg_get_bits(bytes([0xAB]), [4, 8]) == [0xA]                             # end of data
    for ycc, rgb in (((128, 128, 128), (128, 128, 128)), ((255, 128, 128), (255, 255, 255)), ((0, 128, 128), (0, 0, 0))):
        got = oracle.convert(np.array([[ycc]], np.uint8), oracle.CS_YCBCR, oracle.CS_RGB, np.uint8, 3)
        assert tuple(got[0, 0]) == rgb
    # Marker.fromBytes: SOI and SOF0 are markers the decoder knows (a stream of just those two is "no scan data", not "invalid marker")
    expect_error(oracle, "NoScanData", oracle.jpeg_decode_state, SIG + bytes([0xFF, 0xC0, 0x00, 0x0B, 8, 0, 8, 0, 8, 1, 1, 0x11, 0]) + EOI)
