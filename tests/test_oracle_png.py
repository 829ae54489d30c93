"""The PNG oracle (oracle/png.c) pinned to the reference's own tests (src/codecs/png.zig:2073-3493) and to a numpy model
built from the PNG specification (tests/png_util.py). CPU only."""
import struct
import zlib

import numpy as np
import pytest

from tests import png_util as P


def rgb_test_image(rows=8, cols=16):  # makeTruncationTestPng, png.zig:2226-2233
    i = np.arange(rows * cols, dtype=np.uint32)
    return np.stack([i * 3, i * 5 + 1, i * 7 + 2], -1).astype(np.uint8).reshape(rows, cols, 3)


def expect_error(oracle, name, fn, *args, **kw):
    with pytest.raises(oracle.PngError) as e:
        fn(*args, **kw)
    assert e.value.name == name, f"expected error.{name}, got error.{e.value.name}"


# ---- known answers of the reference's unit tests -------------------------------------------------------------------------

def test_crc_and_paeth_known_answers(oracle):
    assert oracle.png_crc(b"IEND") == 0xAE426082  # the CRC every PNG file ends with
    assert oracle.png_crc(b"IHDR" + bytes([0, 0, 0, 4, 0, 0, 0, 4, 8, 2, 0, 0, 0])) == zlib.crc32(b"IHDR" + bytes([0, 0, 0, 4, 0, 0, 0, 4, 8, 2, 0, 0, 0]))
    assert oracle.png_paeth(10, 20, 15) == 15  # png.zig:2579-2584
    assert oracle.png_paeth(5, 20, 15) == 5
    assert oracle.png_paeth(10, 5, 6) == 10


def test_chunk_ordering_errors(oracle):  # png.zig:2073-2198
    dec = oracle.png_decode_chunks
    expect_error(oracle, "InvalidPngSignature", dec, bytes([1, 2, 3, 4, 5, 6, 7, 8]))
    expect_error(oracle, "ChunkBeforeHeader", dec, P.SIGNATURE + P.chunk(b"PLTE", bytes(3)) + P.chunk(b"IEND"))
    expect_error(oracle, "MissingPalette", dec, P.SIGNATURE + P.ihdr(1, 1, 8, P.PALETTE) + P.chunk(b"IDAT") + P.chunk(b"IEND"))
    expect_error(oracle, "TransparencyBeforePalette", dec,
                 P.SIGNATURE + P.ihdr(1, 1, 8, P.PALETTE) + P.chunk(b"tRNS", b"\0") + P.chunk(b"PLTE", bytes(3)) + P.chunk(b"IDAT") + P.chunk(b"IEND"))
    expect_error(oracle, "PaletteForbiddenForColorType", dec, P.SIGNATURE + P.ihdr(1, 1, 8, P.GRAY) + P.chunk(b"PLTE", bytes(3)) + P.chunk(b"IEND"))
    expect_error(oracle, "NonConsecutiveIdatChunks", dec,
                 P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB) + P.chunk(b"IDAT", P.EMPTY_ZLIB) + P.chunk(b"tEXt", b"key\0val") + P.chunk(b"IDAT", P.EMPTY_ZLIB) + P.chunk(b"IEND"))
    expect_error(oracle, "GammaAfterPalette", dec,
                 P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB) + P.chunk(b"PLTE", bytes(3)) + P.chunk(b"gAMA", bytes([0, 0, 0, 1])) + P.chunk(b"IEND"))
    expect_error(oracle, "SrgbAfterImageData", dec,
                 P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB) + P.chunk(b"IDAT", P.EMPTY_ZLIB) + P.chunk(b"sRGB", b"\0") + P.chunk(b"IEND"))


def test_more_structural_errors(oracle):  # the remaining branches of decode (png.zig:629-794) and parseHeader (:561-625)
    dec = oracle.png_decode_chunks
    idat, iend = P.chunk(b"IDAT", P.EMPTY_ZLIB), P.chunk(b"IEND")
    expect_error(oracle, "MissingHeader", dec, P.SIGNATURE)
    expect_error(oracle, "MissingImageData", dec, P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB) + iend)
    expect_error(oracle, "MultipleHeaders", dec, P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB) + P.ihdr(1, 1, 8, P.RGB) + idat + iend)
    expect_error(oracle, "InvalidDimensions", dec, P.SIGNATURE + P.ihdr(0, 1, 8, P.RGB) + idat + iend)
    expect_error(oracle, "InvalidColorType", dec, P.SIGNATURE + P.ihdr(1, 1, 8, 5) + idat + iend)
    expect_error(oracle, "InvalidBitDepth", dec, P.SIGNATURE + P.ihdr(1, 1, 4, P.RGB) + idat + iend)
    expect_error(oracle, "InvalidBitDepth", dec, P.SIGNATURE + P.ihdr(1, 1, 16, P.PALETTE) + idat + iend)
    expect_error(oracle, "UnsupportedCompressionMethod", dec, P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB, compression=1) + idat + iend)
    expect_error(oracle, "UnsupportedFilterMethod", dec, P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB, filter_method=1) + idat + iend)
    expect_error(oracle, "UnsupportedInterlaceMethod", dec, P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB, interlace=2) + idat + iend)
    expect_error(oracle, "InvalidHeaderLength", dec, P.SIGNATURE + P.chunk(b"IHDR", bytes(12)) + idat + iend)
    bad_crc = bytearray(P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB) + idat + iend)
    bad_crc[8 + 8 + 13] ^= 1
    expect_error(oracle, "InvalidCrc", dec, bytes(bad_crc))
    head = P.SIGNATURE + P.ihdr(1, 1, 8, P.PALETTE)
    expect_error(oracle, "InvalidPaletteLength", dec, head + P.chunk(b"PLTE", bytes(4)) + idat + iend)
    expect_error(oracle, "PaletteTooLarge", dec, head + P.chunk(b"PLTE", bytes(3 * 257)) + idat + iend)
    expect_error(oracle, "DuplicatePalette", dec, head + P.chunk(b"PLTE", bytes(3)) + P.chunk(b"PLTE", bytes(3)) + idat + iend)
    expect_error(oracle, "PaletteAfterImageData", dec, head + P.chunk(b"PLTE", bytes(3)) + idat + P.chunk(b"PLTE", bytes(3)) + iend)
    expect_error(oracle, "InvalidTransparencyLength", dec, head + P.chunk(b"PLTE", bytes(3)) + P.chunk(b"tRNS", bytes(2)) + idat + iend)
    expect_error(oracle, "MultipleTransparencyChunks", dec, head + P.chunk(b"PLTE", bytes(6)) + P.chunk(b"tRNS", b"\1") + P.chunk(b"tRNS", b"\1") + idat + iend)
    expect_error(oracle, "TransparencyAfterImageData", dec, head + P.chunk(b"PLTE", bytes(6)) + idat + P.chunk(b"tRNS", b"\1") + iend)
    rgb = P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB)
    expect_error(oracle, "InvalidTransparencyLength", dec, rgb + P.chunk(b"tRNS", bytes(2)) + idat + iend)
    expect_error(oracle, "InvalidTransparencyLength", dec, P.SIGNATURE + P.ihdr(1, 1, 8, P.GRAY) + P.chunk(b"tRNS", bytes(6)) + idat + iend)
    expect_error(oracle, "InvalidTransparencyForColorType", dec, P.SIGNATURE + P.ihdr(1, 1, 8, P.RGBA) + P.chunk(b"tRNS", bytes(2)) + idat + iend)
    expect_error(oracle, "InvalidGammaLength", dec, rgb + P.chunk(b"gAMA", bytes(3)) + idat + iend)
    expect_error(oracle, "GammaAfterImageData", dec, rgb + idat + P.chunk(b"gAMA", bytes(4)) + iend)
    expect_error(oracle, "InvalidSrgbLength", dec, rgb + P.chunk(b"sRGB", bytes(2)) + idat + iend)
    expect_error(oracle, "InvalidSrgbIntent", dec, rgb + P.chunk(b"sRGB", b"\4") + idat + iend)
    expect_error(oracle, "SrgbAfterPalette", dec, rgb + P.chunk(b"PLTE", bytes(3)) + P.chunk(b"sRGB", b"\0") + idat + iend)
    expect_error(oracle, "ColorProfileConflict", dec, rgb + P.chunk(b"iCCP", b"x") + P.chunk(b"sRGB", b"\0") + idat + iend)
    expect_error(oracle, "ColorProfileConflict", dec, rgb + P.chunk(b"sRGB", b"\0") + P.chunk(b"iCCP", b"x") + idat + iend)
    expect_error(oracle, "IccpAfterPalette", dec, rgb + P.chunk(b"PLTE", bytes(3)) + P.chunk(b"iCCP", b"x") + idat + iend)
    expect_error(oracle, "IccpAfterImageData", dec, rgb + idat + P.chunk(b"iCCP", b"x") + iend)
    # gAMA / sRGB values land in the header (png.zig:3367-3432)
    h, _, _, _ = dec(rgb + P.chunk(b"gAMA", bytes([0, 0, 0xB1, 0x8F])) + idat + iend)
    assert h.has_gamma and abs(h.gamma - 1 / 2.2) < 1e-3
    h, _, _, _ = dec(rgb + P.chunk(b"sRGB", b"\0") + idat + iend)
    assert h.has_srgb and h.srgb_intent == 0


def test_limits(oracle):  # png.zig:2472-2569, :2952-2989
    dec, L = oracle.png_decode_chunks, oracle.png_limits
    expect_error(oracle, "PngDataTooLarge", dec, P.SIGNATURE + b"\0", L(max_png_bytes=8))
    expect_error(oracle, "ChunkDataLimitExceeded", dec, P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB),
                 L(max_png_bytes=1024, max_chunk_bytes=8, max_idat_bytes=1024, max_chunks=16))
    body = P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB) + P.chunk(b"IDAT", P.EMPTY_ZLIB) + P.chunk(b"IEND")
    expect_error(oracle, "ImageDataLimitExceeded", dec, body, L(max_png_bytes=1024, max_chunk_bytes=1024, max_idat_bytes=4, max_chunks=16))
    expect_error(oracle, "TooManyChunks", dec, P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB) + P.chunk(b"IEND"),
                 L(max_png_bytes=1024, max_chunk_bytes=1024, max_chunks=1))
    gray = P.SIGNATURE + P.ihdr(1, 1, 8, P.GRAY) + P.chunk(b"IDAT", P.EMPTY_ZLIB) + P.chunk(b"IEND")
    expect_error(oracle, "ImageTooLarge", dec, gray, L(max_png_bytes=1024, max_chunk_bytes=1024, max_idat_bytes=1024, max_chunks=16, max_decompressed_bytes=1))
    expect_error(oracle, "ImageTooLarge", dec, P.SIGNATURE + P.ihdr(50000, 50000, 8, P.RGB))
    # the default inflate limit covers 8K x 8K RGBA 16-bit Adam7 (png.zig:2556-2569): the chunk layer accepts the header
    big = P.SIGNATURE + P.ihdr(8192, 8192, 16, P.RGBA, interlace=1) + P.chunk(b"IDAT", P.EMPTY_ZLIB) + P.chunk(b"IEND")
    h, _, _, _ = dec(big)
    assert (h.width, h.height) == (8192, 8192)
    # a zero limit disables it
    dec(P.SIGNATURE + P.ihdr(50000, 10, 8, P.GRAY) + P.chunk(b"IDAT", P.EMPTY_ZLIB) + P.chunk(b"IEND"), L(max_width=0))


def test_missing_iend_decodes_as_truncated(oracle):  # png.zig:2200-2224
    data = P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB) + P.chunk(b"IDAT", P.EMPTY_ZLIB)
    _, truncated, _, _ = oracle.png_decode_chunks(data)
    assert truncated
    img, truncated, _ = oracle.png_decode_native(data)
    assert truncated and img.shape == (1, 1, 3) and not img.any()


def test_truncation_recovery(oracle):  # png.zig:2280-2351
    full_img = rgb_test_image()
    png = P.make_png(full_img, 8, P.RGB, filters=lambda y: y % 5)
    full, t, _ = oracle.png_decode_native(png)
    assert not t and np.array_equal(full, full_img)
    idat_at = png.index(b"IDAT")
    idat_len = struct.unpack(">I", png[idat_at - 4:idat_at])[0]
    # cut in the middle of the IDAT payload: a prefix of rows survives, the rest is zero
    part, t, _ = oracle.png_decode_native(png[:idat_at + 4 + idat_len // 2])
    assert t and part.shape == full.shape and np.array_equal(part[0, 0], full[0, 0])
    same = (part == full).all(-1)
    assert (same | (part == 0).all(-1)).all()
    # IEND missing, cut four bytes into the IEND header, or replaced by a cut ancillary chunk: the image is complete
    for cut in (png[:-12], png[:-8], png[:-12] + bytes([0, 0, 0, 0x20]) + b"tEXt" + b"AB"):
        part, t, _ = oracle.png_decode_native(cut)
        assert t and np.array_equal(part, full)


def stored_zlib_cut(raw: bytes, declared: int) -> bytes:  # appendTruncatedStoredZlib, png.zig:2260-2268
    return bytes([0x78, 0x01, 0x01, declared & 0xff, declared >> 8, (~declared) & 0xff, ((~declared) >> 8) & 0xff]) + raw


def test_truncated_zlib_stream_drops_partial_row(oracle):  # png.zig:2353-2408
    raw = bytearray(32)
    for r in range(4):
        for i in range(13):
            idx = r * 13 + i
            if idx < 32:
                raw[idx] = 0 if i == 0 else (r * 16 + i) & 0xff
    data = P.SIGNATURE + P.ihdr(4, 4, 8, P.RGB) + P.chunk(b"IDAT", stored_zlib_cut(bytes(raw), 52)) + P.chunk(b"IEND")
    _, truncated, _, _ = oracle.png_decode_chunks(data)
    assert not truncated  # the chunk layer is intact
    img, truncated, _ = oracle.png_decode_native(data)
    assert truncated
    for r in range(2):
        for c in range(4):
            assert tuple(img[r, c]) == (r * 16 + c * 3 + 1, r * 16 + c * 3 + 2, r * 16 + c * 3 + 3)
    assert not img[2:].any()


def test_truncated_adam7_keeps_complete_passes(oracle):  # png.zig:2410-2447
    raw = bytearray([0xAB] * 68)
    for off in (0, 4, 8, 15, 22, 29, 42, 55):
        raw[off] = 0
    data = P.SIGNATURE + P.ihdr(8, 8, 8, P.RGB, interlace=1) + P.chunk(b"IDAT", stored_zlib_cut(bytes(raw), 207)) + P.chunk(b"IEND")
    img, truncated, _ = oracle.png_decode_native(data)
    assert truncated
    filled, zero = (0xAB,) * 3, (0,) * 3
    assert tuple(img[0, 0]) == filled and tuple(img[0, 1]) == filled
    assert tuple(img[1, 1]) == zero and tuple(img[2, 1]) == zero
    flat = img.reshape(-1, 3)
    assert ((flat == 0xAB).all(-1) | (flat == 0).all(-1)).all()


def test_structural_corruption_still_errors(oracle):  # png.zig:2449-2470
    expect_error(oracle, "InvalidChunkLength", oracle.png_decode_chunks, P.SIGNATURE + bytes([0, 0, 0, 0x0D]) + b"IHDR" + bytes(2))
    corrupt = P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB) + P.chunk(b"IDAT", bytes([0xFF] * 6)) + P.chunk(b"IEND")
    oracle.png_decode_chunks(corrupt)  # the chunk layer accepts it
    expect_error(oracle, "ReadFailed", oracle.png_decode_native, corrupt)
    # a wrong Adler-32 is corruption too; more data than the header allows is ImageTooLarge (:829-842)
    z = bytearray(zlib.compress(bytes(4)))
    z[-1] ^= 1
    expect_error(oracle, "ReadFailed", oracle.png_decode_native, P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB) + P.chunk(b"IDAT", bytes(z)) + P.chunk(b"IEND"))
    expect_error(oracle, "ImageTooLarge", oracle.png_decode_native,
                 P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB) + P.chunk(b"IDAT", zlib.compress(bytes(5))) + P.chunk(b"IEND"))
    expect_error(oracle, "InvalidFilterType", oracle.png_decode_native,
                 P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB) + P.chunk(b"IDAT", zlib.compress(bytes([5, 0, 0, 0]))) + P.chunk(b"IEND"))


def test_bit_unpacking_and_16_bit_known_answers(oracle):  # png.zig:2728-2822, :2911-2921
    def gray(bits, byte, width):
        data = P.SIGNATURE + P.ihdr(width, 1, bits, P.GRAY) + P.chunk(b"IDAT", zlib.compress(bytes([0, byte]))) + P.chunk(b"IEND")
        return oracle.png_decode_native(data)[0][0].tolist()
    assert gray(1, 0b10110010, 8) == [255, 0, 255, 255, 0, 0, 255, 0]
    assert gray(2, 0b11011000, 4) == [255, 85, 170, 0]
    assert gray(4, 0xF5, 2) == [255, 85]
    vals = [0x0000, 0x00FF, 0xFF00, 0xFFFF, 0x8080, 0x1234]
    data = P.SIGNATURE + P.ihdr(6, 1, 16, P.GRAY) + P.chunk(b"IDAT", zlib.compress(b"\0" + struct.pack(">6H", *vals))) + P.chunk(b"IEND")
    assert oracle.png_decode_native(data)[0][0].tolist() == [0, 0, 255, 255, 128, 18]


def test_transparency_known_answers(oracle):  # png.zig:3151-3365, :3434-3452
    def one_row(samples, bits, ct, **kw):
        return oracle.png_decode_native(P.make_png(np.asarray(samples, np.uint32)[None], bits, ct, **kw))[0][0].tolist()
    # grey tRNS key 0x0080: 128 becomes transparent
    assert one_row([[0], [128], [255], [64]], 8, P.GRAY, trns=[0x00, 0x80]) == [[0, 0, 0, 255], [128, 128, 128, 0], [255, 255, 255, 255], [64, 64, 64, 255]]
    # RGB tRNS "white in 16-bit format" 00FF 00FF 00FF
    assert one_row([[255, 0, 0], [255, 255, 255], [0, 0, 255]], 8, P.RGB, trns=[0, 0xFF] * 3) == [[255, 0, 0, 255], [255, 255, 255, 0], [0, 0, 255, 255]]
    assert one_row([[255, 0, 0], [0, 255, 0]], 8, P.RGB, trns=[0, 0xFF, 0, 0, 0, 0]) == [[255, 0, 0, 0], [0, 255, 0, 255]]
    # 16-bit grey, key 0x8000
    assert one_row([[0x8000], [0x4000]], 16, P.GRAY, trns=[0x80, 0x00]) == [[128, 128, 128, 0], [64, 64, 64, 255]]
    # palette + tRNS (Adam7, 1 x 1): index 1 -> green with alpha 64 (:3099-3124)
    assert one_row([[1]], 8, P.PALETTE, interlace=1, palette=[[255, 0, 0], [0, 255, 0]], trns=[255, 64]) == [[0, 255, 0, 64]]
    # 4-bit palette indices 1, 2 (:3126-3149)
    assert one_row([[1], [2]], 4, P.PALETTE, interlace=1, palette=[[0, 0, 0], [10, 20, 30], [40, 50, 60]]) == [[10, 20, 30], [40, 50, 60]]
    # Adam7 extraction (:3069-3097)
    assert one_row([[255, 0, 0, 255], [0, 255, 0, 128]], 8, P.RGBA, interlace=1) == [[255, 0, 0, 255], [0, 255, 0, 128]]


def test_palette_index_out_of_range(oracle):  # :1080 / :1119 error, :2038-2045 fallback when interlaced
    s = np.array([[[0], [2]]], np.uint32)
    pal = [[1, 2, 3], [4, 5, 6]]
    expect_error(oracle, "InvalidPaletteIndex", oracle.png_decode_native, P.make_png(s, 8, P.PALETTE, palette=pal))
    img, _, _ = oracle.png_decode_native(P.make_png(s, 8, P.PALETTE, palette=pal, interlace=1))
    assert img[0].tolist() == [[1, 2, 3], [0, 0, 0]]
    img, _, _ = oracle.png_decode_native(P.make_png(s, 8, P.PALETTE, palette=pal, interlace=1, trns=[9]))
    assert img[0].tolist() == [[1, 2, 3, 9], [0, 0, 0, 255]]


def test_adaptive_filter_selection_known_answer(oracle):  # png.zig:2644-2682
    raw = np.full((2, 8, 3), 128, np.uint8)
    f = oracle.png_filter(raw)
    assert f[0, 0] == 1 and f[1, 0] == 2  # sub on the constant first row, up on the identical second one
    assert f[0, 1:4].tolist() == [128] * 3 and not f[0, 4:].any() and not f[1, 1:].any()


# ---- against the specification model --------------------------------------------------------------------------------------

FORMATS = [(P.GRAY, b) for b in (1, 2, 4, 8, 16)] + [(P.RGB, 8), (P.RGB, 16), (P.PALETTE, 1), (P.PALETTE, 2), (P.PALETTE, 4), (P.PALETTE, 8),
                                                      (P.GRAY_ALPHA, 8), (P.GRAY_ALPHA, 16), (P.RGBA, 8), (P.RGBA, 16)]


@pytest.mark.parametrize("color_type,bit_depth", FORMATS)
@pytest.mark.parametrize("interlace", [0, 1])
def test_every_format_against_the_model(oracle, color_type, bit_depth, interlace):
    rng = np.random.default_rng(color_type * 100 + bit_depth * 2 + interlace)
    for (h, w) in ((1, 1), (3, 5), (9, 17), (16, 33)):
        plen = min(1 << bit_depth, 200) if color_type == P.PALETTE else None
        palette = rng.integers(0, 256, (plen, 3)).tolist() if plen else None
        s = P.random_samples(rng, h, w, bit_depth, color_type, plen)
        for with_trns in (False, True):
            trns = None
            if with_trns:
                if color_type == P.GRAY:
                    key = int(s[0, 0, 0])
                    trns = [key >> 8, key & 0xff]
                elif color_type == P.RGB:
                    trns = [b for c in s[h // 2, w // 2] for b in (int(c) >> 8, int(c) & 0xff)]
                elif color_type == P.PALETTE:
                    trns = rng.integers(0, 256, max(1, plen // 2)).tolist()
                else:
                    continue
            png = P.make_png(s, bit_depth, color_type, interlace, filters=lambda y: int(rng.integers(0, 5)), palette=palette, trns=trns,
                             idat_split=7 if w > 4 else 0)
            got, truncated, header = oracle.png_decode_native(png)
            want = P.native_model(s, bit_depth, color_type, interlace, palette, trns)
            assert not truncated and (header.width, header.height, header.bit_depth, header.color_type) == (w, h, bit_depth, color_type)
            assert got.shape == want.shape and np.array_equal(got, want), (color_type, bit_depth, interlace, h, w, with_trns)
            info = oracle.png_info(png)
            assert (info.width, info.height, info.bit_depth, info.color_type, info.interlace_method) == (w, h, bit_depth, color_type, interlace)


@pytest.mark.parametrize("kind,ch", [("u8", 1), ("rgb_u8", 3), ("rgba_u8", 4)])
def test_filters_against_the_specification(oracle, kind, ch):
    rng = np.random.default_rng(ch)
    img = rng.integers(0, 256, (13, 21, ch), dtype=np.uint8)
    img[4:9] = (img[4:9] // 64) * 64  # flat-ish rows so the predictors matter
    for mode in range(5):
        want = np.frombuffer(P.scan_data(img.astype(np.uint32), 8, {1: P.GRAY, 3: P.RGB, 4: P.RGBA}[ch], 0, mode), np.uint8).reshape(13, -1)
        got = oracle.png_filter(img if ch > 1 else img[..., 0], mode)
        assert np.array_equal(got, want), mode
    # adaptive: per row the cheapest of the allowed filters (sum of |signed byte|), ties to the lowest ordinal; small images analyse every row
    got = oracle.png_filter(img if ch > 1 else img[..., 0])
    for y in range(13):
        costs = []
        for f in range(5):
            if y == 0 and f >= 2:
                continue
            row = P.filter_row(f, img[y].tobytes(), img[y - 1].tobytes() if y else None, ch)
            costs.append((int(np.abs(np.frombuffer(row, np.int8).astype(np.int32)).sum()), f))
        assert got[y, 0] == min(costs)[1], y


def test_adaptive_filter_sampling_on_tall_images(oracle):
    """Above 512 rows only every 8th row, the first / last three and the rows after a change are analysed (png.zig:1675-1706)."""
    rng = np.random.default_rng(9)
    img = np.zeros((600, 12, 3), np.uint8)
    img[:] = rng.integers(0, 256, (1, 12, 3))  # identical rows: 'up' wins everywhere after row 0
    img[300:310] = rng.integers(0, 256, (10, 12, 3))
    f = oracle.png_filter(img)[:, 0]
    # model of the state machine
    last, streak, want = 0, 0, []
    for y in range(600):
        if y % 8 == 0 or streak == 0 or y < 3 or y >= 597:
            costs = []
            for k in range(5):
                if y == 0 and k >= 2:
                    continue
                row = P.filter_row(k, img[y].tobytes(), img[y - 1].tobytes() if y else None, 3)
                costs.append((int(np.abs(np.frombuffer(row, np.int8).astype(np.int32)).sum()), k))
            best = min(costs)[1]
            if best == last:
                streak = min(streak + 1, 8)
            else:
                streak, last = 0, best
            want.append(best)
        else:
            want.append(last)
    assert f.tolist() == want
    assert len(set(want[301:309])) >= 1


def test_round_trip_and_load_conversions(oracle):  # png.zig:2586-2642 (round trip), :1151-1186 (loadFromBytes)
    img = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [255, 0, 255], [0, 255, 255], [128, 128, 128], [255, 255, 255],
                    [0, 0, 0], [64, 64, 64], [192, 192, 192], [128, 0, 128], [128, 128, 0], [0, 128, 128], [255, 128, 64], [64, 255, 128]], np.uint8).reshape(4, 4, 3)
    for mode in (-1, 0, 1, 2, 3, 4):
        png = oracle.png_encode_stored(img, mode)
        assert png[:8] == P.SIGNATURE
        back, t, h = oracle.png_decode_native(png)
        assert not t and np.array_equal(back, img) and (h.color_type, h.bit_depth) == (2, 8)
    # Image(T).convert of the native image
    png = oracle.png_encode_stored(img)
    rgba = oracle.png_load(png, "rgba_u8")
    assert rgba.shape == (4, 4, 4) and np.array_equal(rgba[..., :3], img) and (rgba[..., 3] == 255).all()
    gray = oracle.png_load(png, "u8")
    assert np.array_equal(gray, oracle.convert(img, oracle.CS_RGB, oracle.CS_GRAY, np.uint8, 1))
    # python's own zlib decodes the oracle's stored stream
    idat = png[png.index(b"IDAT") + 4:png.index(b"IEND") - 8]
    assert zlib.decompress(idat) == oracle.png_filter(img).tobytes()
