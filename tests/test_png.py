"""PNG edge of the path (zg_png_*, zignal_amd/csrc/png_codec.hip) against the oracle (oracle/png.c).

CPU part: the chunk layer (zg_png_probe / zg_png_info) is host code and runs without a GPU; it must agree with the oracle
on every structural case, error name for error name, including random corruption. GPU part: decoded pixels, conversions,
the filtered stream and whole files, bit for bit."""
import struct
import zlib

import numpy as np
import pytest

import zignal_amd as zg
from tests import png_util as P
from tests.test_oracle_png import FORMATS, rgb_test_image, stored_zlib_cut


def outcome(fn, *args, **kw):
    """('ok', value) or ('err', ZigErrorName) for either side."""
    try:
        return "ok", fn(*args, **kw)
    except Exception as e:  # oracle.PngError or zg.CodecError: both carry .name
        if not hasattr(e, "name"):
            raise
        return "err", e.name


def header_tuple(h):
    return (h.width, h.height, h.bit_depth, h.color_type, h.interlace_method, h.has_gamma, h.has_srgb, h.srgb_intent, round(h.gamma, 6))


def structural_cases():
    idat, iend = P.chunk(b"IDAT", P.EMPTY_ZLIB), P.chunk(b"IEND")
    rgb, pal = P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB), P.SIGNATURE + P.ihdr(1, 1, 8, P.PALETTE)
    bad_crc = bytearray(rgb + idat + iend)
    bad_crc[8 + 8 + 13] ^= 1
    cases = [
        bytes([1, 2, 3, 4, 5, 6, 7, 8]), b"", P.SIGNATURE, P.SIGNATURE[:5],
        P.SIGNATURE + P.chunk(b"PLTE", bytes(3)) + iend,
        pal + P.chunk(b"IDAT") + iend,
        pal + P.chunk(b"tRNS", b"\0") + P.chunk(b"PLTE", bytes(3)) + P.chunk(b"IDAT") + iend,
        P.SIGNATURE + P.ihdr(1, 1, 8, P.GRAY) + P.chunk(b"PLTE", bytes(3)) + iend,
        rgb + idat + P.chunk(b"tEXt", b"key\0val") + idat + iend,
        rgb + P.chunk(b"PLTE", bytes(3)) + P.chunk(b"gAMA", bytes([0, 0, 0, 1])) + iend,
        rgb + idat + P.chunk(b"sRGB", b"\0") + iend,
        rgb + idat, rgb + iend, rgb + idat + idat + iend, rgb + P.chunk(b"tEXt", b"a\0b") + idat + iend,
        P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB) + P.ihdr(1, 1, 8, P.RGB) + idat + iend,
        P.SIGNATURE + P.ihdr(0, 1, 8, P.RGB) + idat + iend, P.SIGNATURE + P.ihdr(1, 0, 8, P.RGB) + idat + iend,
        P.SIGNATURE + P.ihdr(1, 1, 8, 5) + idat + iend, P.SIGNATURE + P.ihdr(1, 1, 4, P.RGB) + idat + iend,
        P.SIGNATURE + P.ihdr(1, 1, 16, P.PALETTE) + idat + iend, P.SIGNATURE + P.ihdr(1, 1, 3, P.GRAY) + idat + iend,
        P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB, compression=1) + idat + iend, P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB, filter_method=1) + idat + iend,
        P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB, interlace=2) + idat + iend, P.SIGNATURE + P.chunk(b"IHDR", bytes(12)) + idat + iend,
        bytes(bad_crc),
        pal + P.chunk(b"PLTE", bytes(4)) + idat + iend, pal + P.chunk(b"PLTE", bytes(3 * 257)) + idat + iend, pal + P.chunk(b"PLTE") + idat + iend,
        pal + P.chunk(b"PLTE", bytes(3)) + P.chunk(b"PLTE", bytes(3)) + idat + iend, pal + P.chunk(b"PLTE", bytes(3)) + idat + P.chunk(b"PLTE", bytes(3)) + iend,
        pal + P.chunk(b"PLTE", bytes(3)) + P.chunk(b"tRNS", bytes(2)) + idat + iend,
        pal + P.chunk(b"PLTE", bytes(6)) + P.chunk(b"tRNS", b"\1") + P.chunk(b"tRNS", b"\1") + idat + iend,
        pal + P.chunk(b"PLTE", bytes(6)) + idat + P.chunk(b"tRNS", b"\1") + iend, pal + P.chunk(b"PLTE", bytes(6)) + P.chunk(b"tRNS", b"\1\2") + idat + iend,
        rgb + P.chunk(b"tRNS", bytes(2)) + idat + iend, rgb + P.chunk(b"tRNS", bytes(6)) + idat + iend,
        P.SIGNATURE + P.ihdr(1, 1, 8, P.GRAY) + P.chunk(b"tRNS", bytes(6)) + idat + iend, P.SIGNATURE + P.ihdr(1, 1, 8, P.GRAY) + P.chunk(b"tRNS", bytes(2)) + idat + iend,
        P.SIGNATURE + P.ihdr(1, 1, 8, P.RGBA) + P.chunk(b"tRNS", bytes(2)) + idat + iend, P.SIGNATURE + P.ihdr(1, 1, 8, P.GRAY_ALPHA) + P.chunk(b"tRNS", bytes(2)) + idat + iend,
        rgb + P.chunk(b"gAMA", bytes(3)) + idat + iend, rgb + idat + P.chunk(b"gAMA", bytes(4)) + iend, rgb + P.chunk(b"gAMA", bytes([0, 0, 0xB1, 0x8F])) + idat + iend,
        rgb + P.chunk(b"sRGB", bytes(2)) + idat + iend, rgb + P.chunk(b"sRGB", b"\4") + idat + iend, rgb + P.chunk(b"sRGB", b"\2") + idat + iend,
        rgb + P.chunk(b"PLTE", bytes(3)) + P.chunk(b"sRGB", b"\0") + idat + iend,
        rgb + P.chunk(b"iCCP", b"x") + P.chunk(b"sRGB", b"\0") + idat + iend, rgb + P.chunk(b"sRGB", b"\0") + P.chunk(b"iCCP", b"x") + idat + iend,
        rgb + P.chunk(b"PLTE", bytes(3)) + P.chunk(b"iCCP", b"x") + idat + iend, rgb + idat + P.chunk(b"iCCP", b"x") + iend,
        P.SIGNATURE + bytes([0, 0, 0, 0x0D]) + b"IHDR" + bytes(2),
        P.SIGNATURE + P.ihdr(50000, 50000, 8, P.RGB), P.SIGNATURE + P.ihdr(8192, 8192, 16, P.RGBA, interlace=1) + idat + iend,
        P.SIGNATURE + P.ihdr(8193, 1, 8, P.GRAY) + idat + iend, P.SIGNATURE + P.ihdr(8192, 8192, 8, P.GRAY) + idat + iend,
    ]
    limited = [
        (P.SIGNATURE + b"\0", dict(max_png_bytes=8)),
        (rgb, dict(max_png_bytes=1024, max_chunk_bytes=8, max_idat_bytes=1024, max_chunks=16)),
        (rgb + idat + iend, dict(max_png_bytes=1024, max_chunk_bytes=1024, max_idat_bytes=4, max_chunks=16)),
        (rgb + iend, dict(max_png_bytes=1024, max_chunk_bytes=1024, max_chunks=1)),
        (P.SIGNATURE + P.ihdr(1, 1, 8, P.GRAY) + idat + iend, dict(max_png_bytes=1024, max_chunk_bytes=1024, max_idat_bytes=1024, max_chunks=16, max_decompressed_bytes=1)),
        (P.SIGNATURE + P.ihdr(50000, 10, 8, P.GRAY) + idat + iend, dict(max_width=0)),
        (P.SIGNATURE + P.ihdr(100, 100, 8, P.GRAY) + idat + iend, dict(max_pixels=9999)),
        (P.SIGNATURE + P.ihdr(100, 100, 8, P.GRAY) + idat + iend, dict(max_pixels=0, max_height=99)),
    ]
    return [(c, {}) for c in cases] + limited


def probe_both(oracle, data, limits):
    want = outcome(oracle.png_decode_chunks, data, oracle.png_limits(**limits) if limits else None)
    got = outcome(zg.png.decode, data, zg.png.decode_limits(**limits) if limits else None)
    return want, got


NATIVE_KIND = {0: "u8", 2: "rgb_u8", 3: "rgba_u8"}


def test_chunk_layer_matches_oracle_case_by_case(oracle):
    for i, (data, limits) in enumerate(structural_cases()):
        want, got = probe_both(oracle, data, limits)
        assert want[0] == got[0], (i, want, got)
        if want[0] == "err":
            assert want[1] == got[1], (i, want, got)
        else:
            (wh, wt, _, _), (gh, _gk, gt) = want[1], got[1]
            assert header_tuple(wh) == header_tuple(gh) and wt == gt, i


def test_get_info_matches_oracle(oracle):
    rng = np.random.default_rng(3)
    for i, (data, limits) in enumerate(structural_cases()):
        want = outcome(oracle.png_info, data, oracle.png_limits(**limits) if limits else None)
        got = outcome(zg.png.get_info, data, zg.png.decode_limits(**limits) if limits else None)
        assert want[0] == got[0] and (want[1] == got[1] if want[0] == "err" else header_tuple(want[1]) == header_tuple(got[1])), (i, want, got)
    base = P.make_png(rgb_test_image().astype(np.uint32), 8, P.RGB, pre_idat=P.chunk(b"gAMA", struct.pack(">I", 45455)) + P.chunk(b"tEXt", b"k\0v"))
    for cut in range(0, len(base), 3):  # every kind of early end: quiet stop between chunks, EndOfStream inside one
        want, got = outcome(oracle.png_info, base[:cut]), outcome(zg.png.get_info, base[:cut])
        assert want[0] == got[0] and (want[1] == got[1] if want[0] == "err" else header_tuple(want[1]) == header_tuple(got[1])), cut
    assert rng is not None


def test_chunk_layer_under_random_corruption(oracle):
    """Byte flips, cuts and duplicated / reordered chunks of valid files: the same outcome (error name, or header +
    truncated flag) from the product's chunk layer and the oracle's."""
    rng = np.random.default_rng(11)
    pal = rng.integers(0, 256, (16, 3)).tolist()
    bases = [
        P.make_png(rgb_test_image().astype(np.uint32), 8, P.RGB, filters=4, idat_split=40),
        P.make_png(rng.integers(0, 16, (9, 11, 1)), 4, P.PALETTE, interlace=1, palette=pal, trns=[1, 2, 3],
                   pre_idat=P.chunk(b"gAMA", struct.pack(">I", 45455))),
        P.make_png(rng.integers(0, 65536, (5, 7, 2)), 16, P.GRAY_ALPHA, pre_idat=P.chunk(b"sRGB", b"\1") + P.chunk(b"tEXt", b"a\0b")),
    ]
    n_err = n_ok = 0
    for base in bases:
        for _ in range(400):
            data = bytearray(base)
            kind = rng.integers(0, 4)
            if kind == 0:
                for _ in range(int(rng.integers(1, 4))):
                    data[int(rng.integers(0, len(data)))] = int(rng.integers(0, 256))
            elif kind == 1:
                data = data[:int(rng.integers(0, len(data) + 1))]
            elif kind == 2:  # flip a byte and repair the CRC of the chunk it is in, so the damage reaches the chunk logic
                pos, at = 8, int(rng.integers(8, len(data)))
                while pos + 12 <= len(data):
                    ln = struct.unpack(">I", data[pos:pos + 4])[0]
                    if pos <= at < pos + 12 + ln:
                        if pos + 8 <= at < pos + 8 + ln or pos + 4 <= at < pos + 8:
                            data[at] = int(rng.integers(0, 256))
                            data[pos + 8 + ln:pos + 12 + ln] = struct.pack(">I", zlib.crc32(bytes(data[pos + 4:pos + 8 + ln])) & 0xffffffff)
                        break
                    pos += 12 + ln
            else:  # duplicate a random chunk somewhere else
                chunks, pos = [], 8
                while pos + 12 <= len(data):
                    ln = struct.unpack(">I", data[pos:pos + 4])[0]
                    chunks.append(bytes(data[pos:pos + 12 + ln]))
                    pos += 12 + ln
                chunks.insert(int(rng.integers(0, len(chunks) + 1)), chunks[int(rng.integers(0, len(chunks)))])
                data = bytearray(P.SIGNATURE + b"".join(chunks))
            want, got = probe_both(oracle, bytes(data), {})
            assert want[0] == got[0], (want, got)
            if want[0] == "err":
                assert want[1] == got[1], (want, got)
                n_err += 1
            else:
                assert header_tuple(want[1][0]) == header_tuple(got[1][0]) and want[1][1] == got[1][2]
                n_ok += 1
    assert n_err > 100 and n_ok > 100


def test_host_layers_match_oracle_on_the_cpu(oracle):
    """zg_png_scan_hash: chunk layer + inflate (zlib) with the recovery rules + de-filtering + the palette check, hashed, against
    the oracle's own inflater and de-filter: every format, Adam7, all five filters, split IDATs, cuts and flipped bytes."""
    rng = np.random.default_rng(17)
    n, seen = 0, set()
    for color_type, bit_depth in FORMATS:
        for interlace in (0, 1):
            for (h, w) in ((1, 1), (9, 17), (40, 33)):
                plen = min(1 << bit_depth, 200) if color_type == P.PALETTE else None
                palette = rng.integers(0, 256, (plen, 3)).tolist() if plen else None
                s = P.random_samples(rng, h, w, bit_depth, color_type, plen)
                data = P.make_png(s, bit_depth, color_type, interlace, filters=lambda y: int(rng.integers(0, 5)), palette=palette,
                                  idat_split=int(rng.integers(0, 2)) * 101, level=int(rng.integers(0, 10)))
                variants = [data, data[:-12], data[:-8]]
                idat = data.index(b"IDAT")
                for _ in range(4):
                    if rng.random() < 0.5:
                        variants.append(data[:int(rng.integers(idat, len(data)))])
                    else:
                        bad = bytearray(data)
                        bad[int(rng.integers(idat + 4, len(data) - 12))] = int(rng.integers(0, 256))
                        variants.append(bytes(bad))
                for v in variants:
                    want, got = outcome(oracle.png_scan_hash, v), outcome(zg.png.scan_hash, v)
                    assert want == got, (color_type, bit_depth, interlace, h, w, len(v), want, got)
                    seen.add(want[1] if want[0] == "err" else ("truncated" if want[1][1] else "complete"))
                    n += 1
    s = np.array([[[0], [2]]], np.uint32)  # a palette index past PLTE: an error, but not when interlaced
    for inter in (0, 1):
        v = P.make_png(s, 8, P.PALETTE, palette=[[1, 2, 3], [4, 5, 6]], interlace=inter)
        want, got = outcome(oracle.png_scan_hash, v), outcome(zg.png.scan_hash, v)
        assert want == got and (want == ("err", "InvalidPaletteIndex")) == (inter == 0)
    # more data than the header allows wins over a wrong Adler-32 (png.zig:829-842; found by tests/fuzz_host_layers.py)
    z = bytearray(zlib.compress(bytes(5)))
    z[-1] ^= 1
    v = P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB) + P.chunk(b"IDAT", bytes(z)) + P.chunk(b"IEND")
    want, got = outcome(oracle.png_scan_hash, v), outcome(zg.png.scan_hash, v)
    assert want == got == ("err", "ImageTooLarge"), (want, got)
    assert n > 600 and {"complete", "truncated", "InvalidCrc"} <= seen, seen


def test_compress_scanlines_is_one_zlib_stream(oracle, monkeypatch):
    """zg_png_compress: the IDAT stream, cut into 1 MiB pieces and deflated on several threads for inputs of 4 MiB and more.
    Whatever the piece / thread count, zlib reads it back as one stream, and so does the oracle's own inflater when the
    stream sits in a PNG file (png.zig:829-842 would reject a wrong Adler-32 or a stream that ends early or late)."""
    rng = np.random.default_rng(5)

    def scanlines(h, w):  # filter byte 0 + smooth-ish bytes: compressible, with matches that reach across the cuts
        img = ((np.arange(h)[:, None] * 3 + np.arange(w * 4)[None, :] // 5) % 251 + rng.integers(0, 3, (h, w * 4))).astype(np.uint8)
        return img, np.concatenate([np.zeros((h, 1), np.uint8), img], 1).tobytes()

    for threads in ("1", "3", "16"):
        monkeypatch.setenv("ZIGNAL_HIP_HOST_THREADS", threads)
        for n in (0, 1, (1 << 20) + 1, (4 << 20) - 1, 4 << 20, (5 << 20) + 12345):
            for level in (-1, 0, 9):
                raw = rng.integers(0, 256, n, dtype=np.uint8).tobytes() if level == 0 else bytes(rng.integers(0, 4, n, dtype=np.uint8))
                assert zlib.decompress(zg.png.compress_scanlines(raw, level)) == raw, (threads, n, level)
        img, raw = scanlines(1100, 1200)  # 5.3 MB of scanlines: six pieces
        z = zg.png.compress_scanlines(raw)
        assert zlib.decompress(z) == raw and len(z) < len(raw)
        file = P.SIGNATURE + P.ihdr(1200, 1100, 8, P.RGBA) + P.chunk(b"IDAT", z) + P.chunk(b"IEND")
        out = oracle.png_decode_native(file)
        assert np.array_equal(out[0].reshape(1100, 4800), img) and not out[1]
        want, got = outcome(oracle.png_scan_hash, file), outcome(zg.png.scan_hash, file)
        assert want == got and want[0] == "ok"
    with pytest.raises(zg.ZignalError):
        zg._lib.check(zg._lib.lib().zg_png_compress(None, 5, -1, None, None))


# ---- GPU ---------------------------------------------------------------------------------------------------------------------

def decode_both(oracle, data, kind=None, limits=None):
    want = outcome(lambda: oracle.png_decode_native(data, oracle.png_limits(**limits) if limits else None))
    got = outcome(lambda: zg.png.load_from_bytes(data, kind, zg.png.decode_limits(**limits) if limits else None, return_truncated=True))
    return want, got


@pytest.mark.gpu
@pytest.mark.parametrize("color_type,bit_depth", FORMATS)
@pytest.mark.parametrize("interlace", [0, 1])
def test_decode_parity_every_format(oracle, color_type, bit_depth, interlace):
    rng = np.random.default_rng(1000 + color_type * 100 + bit_depth * 2 + interlace)
    for (h, w) in ((1, 1), (2, 3), (7, 9), (8, 8), (33, 70), (64, 257)):
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
                             idat_split=int(rng.integers(0, 3)) * 997)
            want, got = decode_both(oracle, png)
            assert want[0] == got[0] == "ok"
            (wimg, wtrunc, _), (gimg, gtrunc) = want[1], got[1]
            g = gimg.to_numpy()
            assert g.shape == wimg.shape and np.array_equal(g, wimg), (color_type, bit_depth, interlace, h, w, with_trns)
            assert np.array_equal(wimg, P.native_model(s, bit_depth, color_type, interlace, palette, trns))
            assert wtrunc == gtrunc is False
            # loadFromBytes(T) for the other two T: the native image through Image.convert
            for kind in ("u8", "rgb_u8", "rgba_u8"):
                conv = zg.png.load_from_bytes(png, kind).to_numpy()
                assert np.array_equal(conv, oracle.png_load(png, kind)), (kind, color_type, bit_depth, interlace)


@pytest.mark.gpu
def test_decode_into_views_host_images_and_float_targets(oracle):
    import torch
    rng = np.random.default_rng(5)
    s = rng.integers(0, 256, (21, 37, 4))
    png = P.make_png(s, 8, P.RGBA, filters=4)
    want, _, _ = oracle.png_decode_native(png)
    # a strided destination (a view of a larger device image)
    big = zg.Image(torch.zeros((40, 64, 4), dtype=torch.uint8, device="cuda"))
    view = big.view((5, 3, 5 + 37, 3 + 21))
    d, trunc = view._desc(), None
    buf = (zg._lib.C.c_uint8 * len(png)).from_buffer_copy(png)
    zg._lib.check(zg.lib().zg_png_decode(buf, len(png), None, zg._lib.C.byref(d), zg.CS_RGBA, None, view._stream()))
    torch.cuda.synchronize()
    out = big.to_numpy()
    assert np.array_equal(out[3:24, 5:42], want)
    out[3:24, 5:42] = 0
    assert not out.any()
    # host destination (zg_png_decode_host)
    host = zg.png.load_from_bytes(png, None, None, device=None)
    assert not host.on_device and np.array_equal(host.data, want)
    # a float T: Image(Rgba(u8)).convert(Rgb(f32)) after the native decode
    dst = zg.Image(torch.zeros((21, 37, 3), dtype=torch.float32, device="cuda"))
    d = dst._desc()
    zg._lib.check(zg.lib().zg_png_decode(buf, len(png), None, zg._lib.C.byref(d), zg.CS_RGB, None, dst._stream()))
    torch.cuda.synchronize()
    assert np.array_equal(dst.to_numpy().view(np.uint32), oracle.convert(want, oracle.CS_RGBA, oracle.CS_RGB, np.float32, 3).view(np.uint32))
    # dimensions must match the header
    with pytest.raises(zg.DimensionMismatch):
        bad = zg.Image(torch.zeros((21, 36, 4), dtype=torch.uint8, device="cuda"))
        d = bad._desc()
        zg._lib.check(zg.lib().zg_png_decode(buf, len(png), None, zg._lib.C.byref(d), zg.CS_RGBA, None, bad._stream()))
    assert trunc is None
    # Image.load_from_bytes / Image.save / Image.load round trip through a file
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "x.PNG")
        zg.Image(torch.from_numpy(want).cuda()).save(path)
        assert np.array_equal(zg.Image.load(path).to_numpy(), want)
        with pytest.raises(zg.ZignalError):
            zg.Image(torch.from_numpy(want).cuda()).save(os.path.join(tmp, "x.bmp"))
    with pytest.raises(zg.ZignalError):
        zg.Image.load_from_bytes(b"\xff\xd8\xff\xe0 not decoded here")


@pytest.mark.gpu
def test_decode_errors_and_truncation_match_oracle(oracle):
    z_bad = bytearray(zlib.compress(bytes(4)))
    z_bad[-1] ^= 1
    rgb1 = P.SIGNATURE + P.ihdr(1, 1, 8, P.RGB)
    pal = [[1, 2, 3], [4, 5, 6]]
    s = np.array([[[0], [2]]], np.uint32)
    raw32 = bytearray(32)
    for r in range(4):
        for i in range(13):
            if r * 13 + i < 32:
                raw32[r * 13 + i] = 0 if i == 0 else (r * 16 + i) & 0xff
    raw68 = bytearray([0xAB] * 68)
    for off in (0, 4, 8, 15, 22, 29, 42, 55):
        raw68[off] = 0
    cases = [
        rgb1 + P.chunk(b"IDAT", bytes([0xFF] * 6)) + P.chunk(b"IEND"),                       # ReadFailed: garbage zlib
        rgb1 + P.chunk(b"IDAT", bytes(z_bad)) + P.chunk(b"IEND"),                            # ReadFailed: Adler-32
        rgb1 + P.chunk(b"IDAT", zlib.compress(bytes(5))) + P.chunk(b"IEND"),                 # ImageTooLarge
        rgb1 + P.chunk(b"IDAT", zlib.compress(bytes([5, 0, 0, 0]))) + P.chunk(b"IEND"),      # InvalidFilterType
        P.make_png(s, 8, P.PALETTE, palette=pal),                                            # InvalidPaletteIndex
        P.make_png(s, 8, P.PALETTE, palette=pal, interlace=1),                               # ... black when interlaced
        P.make_png(s, 8, P.PALETTE, palette=pal, interlace=1, trns=[9]),
        P.make_png(np.array([[[0], [1], [3]]], np.uint32), 2, P.PALETTE, palette=pal + [[7, 8, 9]]),
        rgb1 + P.chunk(b"IDAT", P.EMPTY_ZLIB),                                               # missing IEND, empty stream
        rgb1 + P.chunk(b"IDAT", zlib.compress(bytes(4))[:-2]) + P.chunk(b"IEND"),            # cut inside the checksum: complete
        P.SIGNATURE + P.ihdr(4, 4, 8, P.RGB) + P.chunk(b"IDAT", stored_zlib_cut(bytes(raw32), 52)) + P.chunk(b"IEND"),
        P.SIGNATURE + P.ihdr(8, 8, 8, P.RGB, interlace=1) + P.chunk(b"IDAT", stored_zlib_cut(bytes(raw68), 207)) + P.chunk(b"IEND"),
    ]
    full = P.make_png(rgb_test_image(40, 50).astype(np.uint32), 8, P.RGB, filters=lambda y: y % 5, idat_split=97)
    cases += [full[:-12], full[:-8], full[:-12] + bytes([0, 0, 0, 0x20]) + b"tEXt" + b"AB"]
    rng = np.random.default_rng(21)
    inter = P.make_png(rng.integers(0, 256, (30, 41, 4)), 8, P.RGBA, interlace=1, filters=lambda y: int(rng.integers(0, 5)))
    for base in (full, inter):  # cuts at every kind of place, Huffman blocks included
        cases += [base[:int(c)] for c in rng.integers(60, len(base), 40)]
    seen = set()
    for i, data in enumerate(cases):
        want, got = decode_both(oracle, data)
        assert want[0] == got[0], (i, want[0], got[0], want[1] if want[0] == "err" else "", got[1] if got[0] == "err" else "")
        if want[0] == "err":
            assert want[1] == got[1], (i, want, got)
            seen.add(want[1])
        else:
            (wimg, wtrunc, _), (gimg, gtrunc) = want[1], got[1]
            assert np.array_equal(gimg.to_numpy(), wimg) and wtrunc == gtrunc, (i, wtrunc, gtrunc)
            seen.add("truncated" if wtrunc else "complete")
    assert {"ReadFailed", "ImageTooLarge", "InvalidFilterType", "InvalidPaletteIndex", "truncated", "complete"} <= seen


@pytest.mark.gpu
@pytest.mark.parametrize("kind,ch", [("u8", 1), ("rgb_u8", 3), ("rgba_u8", 4)])
def test_filter_parity(oracle, kind, ch):
    import torch
    rng = np.random.default_rng(ch)
    for (h, w) in ((1, 1), (2, 8), (13, 21), (64, 300), (600, 40), (1030, 17)):
        img = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
        img[h // 3:h // 2] = (img[h // 3:h // 2] // 64) * 64
        if h > 100:
            img[100:300] = img[100]  # long runs where 'up' wins, so the sampling state machine settles and is re-triggered
        host = img if ch > 1 else img[..., 0]
        dev = zg.Image(torch.from_numpy(np.ascontiguousarray(host)).cuda())
        for mode in (-1, 0, 1, 2, 3, 4):
            got = zg.png.filter_scanlines(dev, mode).cpu().numpy()
            assert np.array_equal(got, oracle.png_filter(host, mode)), (h, w, mode)
    # a view with stride != cols
    base = rng.integers(0, 256, (50, 90, ch), dtype=np.uint8)
    big = zg.Image(torch.from_numpy(base if ch > 1 else np.ascontiguousarray(base[..., 0])).cuda())
    view = big.view((7, 4, 7 + 61, 4 + 33))
    sub = base[4:37, 7:68] if ch > 1 else base[4:37, 7:68, 0]
    assert np.array_equal(zg.png.filter_scanlines(view).cpu().numpy(), oracle.png_filter(np.ascontiguousarray(sub)))


def split_chunks(png):
    out, pos = [], 8
    while pos + 12 <= len(png):
        ln = struct.unpack(">I", png[pos:pos + 4])[0]
        assert zlib.crc32(png[pos + 4:pos + 8 + ln]) & 0xffffffff == struct.unpack(">I", png[pos + 8 + ln:pos + 12 + ln])[0]
        out.append((png[pos + 4:pos + 8], png[pos + 8:pos + 8 + ln]))
        pos += 12 + ln
    assert pos == len(png)
    return out


@pytest.mark.gpu
def test_encode_files(oracle):
    import torch
    rng = np.random.default_rng(8)
    for shape in ((4, 4, 3), (31, 45), (31, 45, 4), (700, 33, 3)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        if shape[0] > 512:
            img[50:400] = img[50]
        dev = zg.Image(torch.from_numpy(img).cuda())
        for mode in (-1, 0, 4):
            png = zg.png.encode(dev, zg.png.EncodeOptions(filter=mode))
            chunks = split_chunks(png)
            assert png[:8] == P.SIGNATURE and [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
            ch = 1 if img.ndim == 2 else img.shape[2]
            assert chunks[0][1] == struct.pack(">IIBBBBB", shape[1], shape[0], 8, {1: 0, 3: 2, 4: 6}[ch], 0, 0, 0)
            # the compressed stream holds exactly the reference's filtered scanlines
            assert zlib.decompress(chunks[1][1]) == oracle.png_filter(img, mode).tobytes()
            # both decoders read it back
            back, t, _ = oracle.png_decode_native(png)
            assert not t and np.array_equal(back, img)
            assert np.array_equal(zg.png.load_from_bytes(png).to_numpy(), img)
        # host image in (zg_png_encode_host), stored blocks (level 0)
        png0 = zg.png.encode(zg.Image(img), zg.png.EncodeOptions(compression_level=0))
        assert np.array_equal(oracle.png_decode_native(png0)[0], img) and len(png0) > img.size
    # the reference's 4 x 4 round-trip image (png.zig:2586-2642) and the colour-management chunks (:2824-2877)
    img = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0]], np.uint8).reshape(2, 2, 3)
    dev = zg.Image(torch.from_numpy(img).cuda())
    names = [c[0] for c in split_chunks(zg.png.encode(dev, zg.png.EncodeOptions(srgb_intent=0)))]
    assert names == [b"IHDR", b"sRGB", b"IDAT", b"IEND"]
    chunks = split_chunks(zg.png.encode(dev, zg.png.EncodeOptions(gamma=1.0 / 2.2)))
    assert [c[0] for c in chunks] == [b"IHDR", b"gAMA", b"IDAT", b"IEND"]
    assert abs(struct.unpack(">I", chunks[1][1])[0] - int((1.0 / 2.2) * 100000.0)) <= 1
    h = zg.png.get_info(zg.png.encode(dev, zg.png.EncodeOptions(gamma=1.0 / 2.2)))
    assert h.has_gamma and abs(h.gamma - 1 / 2.2) < 1e-4
    both = [c[0] for c in split_chunks(zg.png.encode(dev, zg.png.EncodeOptions(gamma=0.5, srgb_intent=1)))]
    assert both == [b"IHDR", b"sRGB", b"IDAT", b"IEND"]  # sRGB wins
    # any other T is converted to Rgb first: Image(Rgba(f32)) -> convertColor(Rgb, px)
    f = rng.random((9, 14, 4), dtype=np.float32)
    png = zg.png.encode(zg.Image(torch.from_numpy(f).cuda()))
    want = oracle.convert(f, oracle.CS_RGBA, oracle.CS_RGB, np.uint8, 3)
    assert np.array_equal(oracle.png_decode_native(png)[0], want)
    g = rng.random((9, 14), dtype=np.float32)
    png = zg.png.encode(zg.Image(torch.from_numpy(g).cuda()))
    assert np.array_equal(oracle.png_decode_native(png)[0], oracle.convert(g, oracle.CS_GRAY, oracle.CS_RGB, np.uint8, 3))


def test_scan_size_arithmetic_is_checked_like_the_reference(oracle):
    """png.zig:229-245: scanDataLength / adam7TotalSize use std.math.add / std.math.mul and return error.ImageTooLarge. With every
    limit disabled a crafted IHDR (gray16, 2147516415 x 4294901762) used to wrap the byte count to 196606: a buffer of that size
    and a de-filter walk of 4e9 bytes per row past it. It must be refused before anything is sized."""
    import struct
    import zlib

    def png_with_ihdr(w, h, depth, ctype, interlace):
        ihdr = P.chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace))
        return b"\x89PNG\r\n\x1a\n" + ihdr + P.chunk(b"IDAT", zlib.compress(b"\0" * 64)) + P.chunk(b"IEND")

    off = zg.png.decode_limits(max_png_bytes=0, max_chunk_bytes=0, max_idat_bytes=0, max_chunks=0, max_width=0, max_height=0, max_pixels=0,
                               max_decompressed_bytes=0)
    cases = [(2147516415, 4294901762, 16, 0, 0),   # the advisor's reproducer: (2 w + 1) h wraps to 196606
             (0xFFFFFFFF, 0xFFFFFFFF, 16, 6, 0),   # 8 bytes per pixel: row bytes * rows leaves 64 bits
             (0xFFFFFFFF, 0xFFFFFFFF, 16, 6, 1)]   # and the Adam7 sum of passes
    for w, h, depth, ctype, inter in cases:
        data = png_with_ihdr(w, h, depth, ctype, inter)
        assert outcome(zg.png.scan_hash, data, off) == ("err", "ImageTooLarge"), (w, h, inter)
        assert outcome(zg.png.decode, data, off) == ("err", "ImageTooLarge")
        olim = oracle.png_limits(**{k: 0 for k in ("max_png_bytes", "max_chunk_bytes", "max_idat_bytes", "max_chunks", "max_width", "max_height",
                                                    "max_pixels", "max_decompressed_bytes")}) if hasattr(oracle, "png_limits") else None
        if olim is not None:
            assert outcome(oracle.png_scan_hash, data, olim) == ("err", "ImageTooLarge")
    # a large but representable size is still only refused by its limit, not by the arithmetic
    ok = png_with_ihdr(70000, 70000, 8, 0, 0)
    assert outcome(zg.png.scan_hash, ok, zg.png.decode_limits(max_width=0, max_height=0, max_pixels=0, max_decompressed_bytes=1 << 20)) == ("err", "ImageTooLarge")
