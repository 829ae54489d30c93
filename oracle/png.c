/*
 * oracle/png.c — CPU restatement of the reference's PNG codec (src/codecs/png.zig). TEST INFRASTRUCTURE ONLY (see zo.h).
 *
 * Follows the reference function by function: the chunk layer with its ordering rules and limits (decode :629-794,
 * ChunkReader :513-558, parseHeader :561-625, getInfo :308-410), the scan-data recovery rules for cut streams
 * (toNativeImage :801-850, completeScanPrefix :254-272), defiltering (:1442-1533, :1721-1803), native-type selection
 * and pixel extraction incl. Adam7 (:852-1146, :1805-2053), loadFromBytes (:1151-1186), row filtering and the adaptive
 * filter heuristic (:1265-1294, :1535-1719) and the container writer (:1198-1398).
 *
 * The reference inflates through Zig's std.compress.flate (Zig std, not in /root/reference); DEFLATE / zlib are RFC 1951 /
 * RFC 1950, any conforming inflater produces the same bytes on a valid stream, and this file carries its own small one so
 * the oracle shares no code with the product (which links zlib). How many bytes survive a stream cut in the middle of a
 * Huffman block is implementation-defined: pinned here only for the stored-block cases the reference's tests use
 * (png.zig:2353-2447); everything after row-rounding is otherwise "parity unpinned" for cut Huffman streams.
 * The oracle's own encoder writes stored blocks: compressed bytes are never compared, only what they decode to.
 */
#include "zo.h"
#include <stdlib.h>
#include <string.h>

static const uint8_t SIGNATURE[8] = {137, 80, 78, 71, 13, 10, 26, 10};

static const char *const ERR_NAMES[] = {
    "ok", "InvalidPngSignature", "PngDataTooLarge", "TooManyChunks", "ChunkDataLimitExceeded", "InvalidChunkLength",
    "ChunkBeforeHeader", "MultipleHeaders", "InvalidHeaderLength", "InvalidDimensions", "InvalidColorType", "InvalidBitDepth",
    "UnsupportedCompressionMethod", "UnsupportedFilterMethod", "UnsupportedInterlaceMethod", "ImageTooLarge",
    "PaletteForbiddenForColorType", "PaletteAfterImageData", "DuplicatePalette", "InvalidPaletteLength", "PaletteTooLarge",
    "MultipleTransparencyChunks", "TransparencyAfterImageData", "InvalidTransparencyLength", "TransparencyBeforePalette",
    "MissingPalette", "InvalidTransparencyForColorType", "GammaAfterPalette", "GammaAfterImageData", "InvalidGammaLength",
    "SrgbAfterPalette", "SrgbAfterImageData", "InvalidSrgbLength", "ColorProfileConflict", "InvalidSrgbIntent",
    "IccpAfterPalette", "IccpAfterImageData", "NonConsecutiveIdatChunks", "ImageDataLimitExceeded", "MissingHeader",
    "MissingImageData", "InvalidCrc", "ReadFailed", "InvalidScanlineData", "InvalidFilterType", "InvalidPaletteIndex",
    "EndOfStream", "OutOfMemory",
};
enum {
    E_OK, E_InvalidPngSignature, E_PngDataTooLarge, E_TooManyChunks, E_ChunkDataLimitExceeded, E_InvalidChunkLength,
    E_ChunkBeforeHeader, E_MultipleHeaders, E_InvalidHeaderLength, E_InvalidDimensions, E_InvalidColorType, E_InvalidBitDepth,
    E_UnsupportedCompressionMethod, E_UnsupportedFilterMethod, E_UnsupportedInterlaceMethod, E_ImageTooLarge,
    E_PaletteForbiddenForColorType, E_PaletteAfterImageData, E_DuplicatePalette, E_InvalidPaletteLength, E_PaletteTooLarge,
    E_MultipleTransparencyChunks, E_TransparencyAfterImageData, E_InvalidTransparencyLength, E_TransparencyBeforePalette,
    E_MissingPalette, E_InvalidTransparencyForColorType, E_GammaAfterPalette, E_GammaAfterImageData, E_InvalidGammaLength,
    E_SrgbAfterPalette, E_SrgbAfterImageData, E_InvalidSrgbLength, E_ColorProfileConflict, E_InvalidSrgbIntent,
    E_IccpAfterPalette, E_IccpAfterImageData, E_NonConsecutiveIdatChunks, E_ImageDataLimitExceeded, E_MissingHeader,
    E_MissingImageData, E_InvalidCrc, E_ReadFailed, E_InvalidScanlineData, E_InvalidFilterType, E_InvalidPaletteIndex,
    E_EndOfStream, E_OutOfMemory,
};
ZO_API const char *zo_png_error_name(int code) {
    return code >= 0 && code < (int)(sizeof ERR_NAMES / sizeof ERR_NAMES[0]) ? ERR_NAMES[code] : "?";
}

ZO_API void zo_png_default_limits(zo_png_limits *l) { /* png.zig:16-41 */
    const size_t max_file = 100u * 1024 * 1024;
    l->max_png_bytes = l->max_chunk_bytes = l->max_idat_bytes = max_file;
    l->max_chunks = 8192;
    l->max_width = l->max_height = 8192;
    l->max_pixels = 67108864ull;
    l->max_decompressed_bytes = 536886272u;
}

/* ---- CRC-32 (png.zig:437-511: the PNG polynomial, bytewise table) ----------------------------------------------------- */
static uint32_t crc_table[256];
static int crc_ready;
static uint32_t crc_update(uint32_t c, const uint8_t *p, size_t n) {
    if (!crc_ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t v = i;
            for (int k = 0; k < 8; ++k) v = (v & 1) ? 0xedb88320u ^ (v >> 1) : v >> 1;
            crc_table[i] = v;
        }
        crc_ready = 1;
    }
    for (size_t i = 0; i < n; ++i) c = crc_table[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return c;
}
ZO_API uint32_t zo_png_crc(const uint8_t *p, size_t n) { return crc_update(0xffffffffu, p, n) ^ 0xffffffffu; }

static uint32_t be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
static void put_be32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
static int exceeds(uint64_t limit, uint64_t v) { return limit != 0 && v > limit; }

/* ---- header geometry (png.zig:135-272) --------------------------------------------------------------------------------- */
static int channels_of(int color_type) {
    switch (color_type) { case 0: return 1; case 2: return 3; case 3: return 1; case 4: return 2; default: return 4; }
}
static size_t scanline_bytes(const zo_png_header *h) { return ((size_t)h->width * channels_of(h->color_type) * h->bit_depth + 7) / 8; }
static int bytes_per_pixel(const zo_png_header *h) { return (channels_of(h->color_type) * h->bit_depth + 7) / 8; }

static const uint32_t A7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}}; /* x0 y0 dx dy */
static void a7_dims(int pass, uint32_t w, uint32_t h, uint32_t *pw, uint32_t *ph) {
    *pw = w > A7[pass][0] ? (uint32_t)(((uint64_t)w - A7[pass][0] + A7[pass][2] - 1) / A7[pass][2]) : 0;
    *ph = h > A7[pass][1] ? (uint32_t)(((uint64_t)h - A7[pass][1] + A7[pass][3] - 1) / A7[pass][3]) : 0;
}
static size_t a7_scanline_bytes(uint32_t pw, const zo_png_header *h) { return ((size_t)pw * channels_of(h->color_type) * h->bit_depth + 7) / 8; }
/* scanDataLength / adam7TotalSize (:219-245): std.math.add / std.math.mul, error.ImageTooLarge when a usize overflows. */
static int scan_data_length(const zo_png_header *h, size_t *out) {
    size_t stride, total = 0;
    if (h->interlace_method != 1) {
        if (__builtin_add_overflow(scanline_bytes(h), (size_t)1, &stride) || __builtin_mul_overflow(stride, (size_t)h->height, &total)) return E_ImageTooLarge;
        *out = total;
        return E_OK;
    }
    for (int p = 0; p < 7; ++p) {
        uint32_t pw, ph;
        size_t pass_total;
        a7_dims(p, h->width, h->height, &pw, &ph);
        if (!pw || !ph) continue;
        if (__builtin_add_overflow(a7_scanline_bytes(pw, h), (size_t)1, &stride) || __builtin_mul_overflow(stride, (size_t)ph, &pass_total) ||
            __builtin_add_overflow(total, pass_total, &total)) return E_ImageTooLarge;
    }
    *out = total;
    return E_OK;
}
static size_t complete_scan_prefix(size_t len, const zo_png_header *h) { /* :254-272 */
    if (h->interlace_method != 1) {
        const size_t stride = scanline_bytes(h) + 1;
        return len - len % stride;
    }
    size_t kept = 0, rem = len;
    for (int p = 0; p < 7; ++p) {
        uint32_t pw, ph;
        a7_dims(p, h->width, h->height, &pw, &ph);
        if (!pw || !ph) continue;
        const size_t stride = a7_scanline_bytes(pw, h) + 1, total = stride * ph;
        if (rem < total) return kept + rem - rem % stride;
        kept += total;
        rem -= total;
    }
    return kept;
}

static int parse_header(const uint8_t *d, uint32_t length, zo_png_header *h) { /* :561-625 */
    if (length != 13) return E_InvalidHeaderLength;
    memset(h, 0, sizeof *h);
    h->width = be32(d);
    h->height = be32(d + 4);
    h->bit_depth = d[8];
    if (h->width == 0 || h->height == 0) return E_InvalidDimensions;
    const int ct = d[9], bd = d[8];
    if (!(ct == 0 || ct == 2 || ct == 3 || ct == 4 || ct == 6)) return E_InvalidColorType;
    h->color_type = (uint8_t)ct;
    int ok;
    switch (ct) {
    case 0: ok = bd == 1 || bd == 2 || bd == 4 || bd == 8 || bd == 16; break;
    case 3: ok = bd == 1 || bd == 2 || bd == 4 || bd == 8; break;
    default: ok = bd == 8 || bd == 16;
    }
    if (!ok) return E_InvalidBitDepth;
    if (d[10] != 0) return E_UnsupportedCompressionMethod;
    if (d[11] != 0) return E_UnsupportedFilterMethod;
    if (d[12] > 1) return E_UnsupportedInterlaceMethod;
    h->interlace_method = d[12];
    return E_OK;
}

/* ---- getInfo (:308-410): a streaming reader, no CRC checks, stops at IDAT / IEND ---------------------------------------- */
ZO_API int zo_png_info(const uint8_t *png, size_t len, const zo_png_limits *lim_in, zo_png_header *out) {
    zo_png_limits lim;
    if (lim_in) lim = *lim_in; else zo_png_default_limits(&lim);
    size_t pos = 0, bytes_read = 0, chunks = 0;
    if (len < 8) return E_EndOfStream;
    if (memcmp(png, SIGNATURE, 8)) return E_InvalidPngSignature;
    pos = bytes_read = 8;
    zo_png_header h;
    int found = 0;
    for (;;) {
        if (exceeds(lim.max_png_bytes, bytes_read)) return E_PngDataTooLarge;
        if (len - pos < 4) break; /* takeInt -> EndOfStream -> break */
        const uint32_t length = be32(png + pos);
        pos += 4; bytes_read += 4;
        if (len - pos < 4) return E_EndOfStream;
        const uint8_t *type = png + pos;
        pos += 4; bytes_read += 4;
        if (exceeds(lim.max_chunks, ++chunks)) return E_TooManyChunks;
        if (lim.max_png_bytes != 0 && bytes_read + (size_t)length + 4 > lim.max_png_bytes) return E_PngDataTooLarge;
        if (!memcmp(type, "IDAT", 4) || !memcmp(type, "IEND", 4)) break;
        if (!memcmp(type, "IHDR", 4)) {
            if (found) return E_MultipleHeaders;
            if (length != 13) return E_InvalidHeaderLength;
            if (len - pos < 13) return E_EndOfStream;
            const uint8_t *d = png + pos;
            memset(&h, 0, sizeof h);
            h.width = be32(d); h.height = be32(d + 4);
            if (h.width == 0 || h.height == 0) return E_InvalidDimensions;
            if (!(d[9] == 0 || d[9] == 2 || d[9] == 3 || d[9] == 4 || d[9] == 6)) return E_InvalidColorType;
            h.bit_depth = d[8]; h.color_type = d[9]; h.compression_method = d[10]; h.filter_method = d[11]; h.interlace_method = d[12];
            found = 1;
            pos += 13; bytes_read += 13;
            const size_t skip = len - pos < 4 ? len - pos : 4; /* discard(limited 4) stops quietly at the end */
            pos += skip; bytes_read += skip;
        } else if (!memcmp(type, "gAMA", 4) && found) {
            if (length != 4) return E_InvalidGammaLength;
            if (len - pos < 4) return E_EndOfStream;
            h.has_gamma = 1;
            h.gamma = (float)be32(png + pos) / 100000.0f;
            pos += 4; bytes_read += 4;
            const size_t skip = len - pos < 4 ? len - pos : 4;
            pos += skip; bytes_read += skip;
        } else if (!memcmp(type, "sRGB", 4) && found) {
            if (length != 1) return E_InvalidSrgbLength;
            if (len - pos < 1) return E_EndOfStream;
            if (png[pos] > 3) return E_InvalidSrgbIntent;
            h.has_srgb = 1;
            h.srgb_intent = png[pos];
            pos += 1; bytes_read += 1;
            const size_t skip = len - pos < 4 ? len - pos : 4;
            pos += skip; bytes_read += skip;
        } else {
            const uint64_t want = (uint64_t)length + 4;
            const size_t skip = (uint64_t)(len - pos) < want ? len - pos : (size_t)want;
            pos += skip; bytes_read += skip;
        }
    }
    if (!found) return E_MissingHeader;
    *out = h;
    return E_OK;
}

/* ---- decode: the chunk layer (:629-794) ------------------------------------------------------------------------------- */
typedef struct png_state {
    zo_png_header header;
    uint8_t palette[256][3];
    int palette_len; /* -1: none */
    uint8_t trns[256];
    int trns_len; /* -1: none */
    uint8_t *idat;
    size_t idat_len;
    size_t scan_data_bytes;
    int truncated;
} png_state;

static int decode_chunks(const uint8_t *png, size_t len, const zo_png_limits *lim, png_state *st) {
    memset(st, 0, sizeof *st);
    st->palette_len = st->trns_len = -1;
    if (len < 8 || memcmp(png, SIGNATURE, 8)) return E_InvalidPngSignature;
    if (exceeds(lim->max_png_bytes, len)) return E_PngDataTooLarge;
    const uint8_t *data = png + 8;
    const size_t dlen = len - 8;
    size_t pos = 0, total_chunk_bytes = 0, total_idat = 0, chunks = 0;
    int header_found = 0, seen_plte = 0, seen_trns = 0, seen_idat = 0, seen_iend = 0, seen_iccp = 0, seen_srgb = 0, idat_done = 0;
    st->idat = (uint8_t *)malloc(len ? len : 1); /* the IDAT payloads cannot exceed the file */
    if (!st->idat) return E_OutOfMemory;
    while (pos + 8 <= dlen) {
        /* ChunkReader.nextChunk (:521-557) */
        const uint32_t length = be32(data + pos);
        const uint8_t *type = data + pos + 4;
        pos += 8;
        const uint8_t *cdata = data + pos;
        size_t clen = length;
        int truncated = 0;
        if ((uint64_t)pos + length + 4 > dlen) {
            clen = (uint64_t)length < dlen - pos ? length : dlen - pos;
            pos = dlen;
            truncated = 1;
        } else {
            pos += length;
            const uint32_t want = be32(data + pos);
            pos += 4;
            if (zo_png_crc(type, (size_t)length + 4) != want) return E_InvalidCrc;
        }
        if (exceeds(lim->max_chunks, ++chunks)) return E_TooManyChunks;
        total_chunk_bytes += clen;
        if (exceeds(lim->max_chunk_bytes, total_chunk_bytes)) return E_ChunkDataLimitExceeded;
        const int is_idat = !memcmp(type, "IDAT", 4), is_ihdr = !memcmp(type, "IHDR", 4);
        if (truncated && !is_idat) {
            if (!seen_idat) return E_InvalidChunkLength;
            break;
        }
        if (!header_found && !is_ihdr) return E_ChunkBeforeHeader;
        if (seen_idat && !is_idat) idat_done = 1;
        const uint32_t chunk_length = truncated ? (uint32_t)clen : length; /* Chunk.length of a cut chunk is what was taken */
        if (is_ihdr) {
            if (header_found) return E_MultipleHeaders;
            const int rc = parse_header(cdata, chunk_length, &st->header);
            if (rc) return rc;
            header_found = 1;
            if (exceeds(lim->max_width, st->header.width) || exceeds(lim->max_height, st->header.height)) return E_ImageTooLarge;
            if (exceeds(lim->max_pixels, (uint64_t)st->header.width * st->header.height)) return E_ImageTooLarge;
        } else if (!memcmp(type, "PLTE", 4)) {
            if (st->header.color_type == 0 || st->header.color_type == 4) return E_PaletteForbiddenForColorType;
            if (seen_idat) return E_PaletteAfterImageData;
            if (st->palette_len >= 0) return E_DuplicatePalette;
            if (chunk_length % 3 != 0) return E_InvalidPaletteLength;
            if (chunk_length / 3 > 256) return E_PaletteTooLarge;
            st->palette_len = (int)(chunk_length / 3);
            memcpy(st->palette, cdata, chunk_length);
            seen_plte = 1;
        } else if (!memcmp(type, "tRNS", 4)) {
            if (seen_trns) return E_MultipleTransparencyChunks;
            if (seen_idat) return E_TransparencyAfterImageData;
            switch (st->header.color_type) {
            case 0: if (chunk_length != 2) return E_InvalidTransparencyLength; break;
            case 2: if (chunk_length != 6) return E_InvalidTransparencyLength; break;
            case 3:
                if (!seen_plte) return E_TransparencyBeforePalette;
                if (st->palette_len < 0) return E_MissingPalette;
                if (chunk_length > (uint32_t)st->palette_len) return E_InvalidTransparencyLength;
                break;
            default: return E_InvalidTransparencyForColorType;
            }
            st->trns_len = (int)chunk_length;
            memcpy(st->trns, cdata, chunk_length);
            seen_trns = 1;
        } else if (!memcmp(type, "gAMA", 4)) {
            if (seen_plte) return E_GammaAfterPalette;
            if (seen_idat) return E_GammaAfterImageData;
            if (chunk_length != 4) return E_InvalidGammaLength;
            st->header.has_gamma = 1;
            st->header.gamma = (float)be32(cdata) / 100000.0f;
        } else if (!memcmp(type, "sRGB", 4)) {
            if (seen_plte) return E_SrgbAfterPalette;
            if (seen_idat) return E_SrgbAfterImageData;
            if (chunk_length != 1) return E_InvalidSrgbLength;
            if (seen_iccp) return E_ColorProfileConflict;
            if (cdata[0] > 3) return E_InvalidSrgbIntent;
            st->header.has_srgb = 1;
            st->header.srgb_intent = cdata[0];
            seen_srgb = 1;
        } else if (!memcmp(type, "iCCP", 4)) {
            if (seen_plte) return E_IccpAfterPalette;
            if (seen_idat) return E_IccpAfterImageData;
            if (seen_srgb) return E_ColorProfileConflict;
            seen_iccp = 1;
        } else if (is_idat) {
            if (idat_done) return E_NonConsecutiveIdatChunks;
            if (st->header.color_type == 3 && st->palette_len < 0) return E_MissingPalette;
            total_idat += clen;
            if (exceeds(lim->max_idat_bytes, total_idat)) return E_ImageDataLimitExceeded;
            memcpy(st->idat + st->idat_len, cdata, clen);
            st->idat_len += clen;
            seen_idat = 1;
            if (truncated) break;
        } else if (!memcmp(type, "IEND", 4)) {
            seen_iend = 1;
            break;
        }
    }
    if (!header_found) return E_MissingHeader;
    if (st->idat_len == 0) return E_MissingImageData;
    if (!seen_iend) st->truncated = 1;
    if (scan_data_length(&st->header, &st->scan_data_bytes) != E_OK) return E_ImageTooLarge;
    if (exceeds(lim->max_decompressed_bytes, st->scan_data_bytes)) return E_ImageTooLarge;
    return E_OK;
}

/* ---- inflate (RFC 1950 / 1951) ------------------------------------------------------------------------------------------ */
/* Returns 0 = stream ended (Adler-32 verified), 1 = input ran out (everything decodable was delivered), 2 = corrupt.
 * Output beyond `cap` is counted, not stored: *produced may exceed cap by one, which the caller reads as "more data". */
typedef struct bitreader { const uint8_t *p; size_t n, pos; uint64_t acc; int bits; } bitreader;
static int need(bitreader *b, int k) { /* 1 when k bits are available */
    while (b->bits < k) {
        if (b->pos >= b->n) return 0;
        b->acc |= (uint64_t)b->p[b->pos++] << b->bits;
        b->bits += 8;
    }
    return 1;
}
static uint32_t take(bitreader *b, int k) { const uint32_t v = (uint32_t)(b->acc & ((1ull << k) - 1)); b->acc >>= k; b->bits -= k; return v; }
typedef struct huff { uint16_t count[16], symbol[288]; } huff;
static int huff_build(huff *h, const uint8_t *lengths, int n) {
    memset(h->count, 0, sizeof h->count);
    for (int i = 0; i < n; ++i) h->count[lengths[i]]++;
    if (h->count[0] == n) return 0; /* no codes: legal, decoding any symbol then fails */
    int left = 1;
    for (int len = 1; len < 16; ++len) {
        left <<= 1;
        left -= h->count[len];
        if (left < 0) return -1; /* over-subscribed */
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (int len = 1; len < 15; ++len) offs[len + 1] = (uint16_t)(offs[len] + h->count[len]);
    for (int i = 0; i < n; ++i)
        if (lengths[i]) h->symbol[offs[lengths[i]]++] = (uint16_t)i;
    return left; /* > 0: incomplete set */
}
/* -1 = input ran out, -2 = invalid code */
static int huff_decode(bitreader *b, const huff *h) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len < 16; ++len) {
        if (!need(b, 1)) return -1;
        code |= (int)take(b, 1);
        const int count = h->count[len];
        if (code - count < first) return h->symbol[index + (code - first)];
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -2;
}
static int zlib_inflate(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *produced) {
    static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    static const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    size_t op = 0;
    *produced = 0;
    if (n < 2) return 1;
    if ((in[0] & 0x0f) != 8 || (in[0] >> 4) > 7 || ((in[0] << 8) | in[1]) % 31 != 0 || (in[1] & 0x20)) return 2;
    bitreader b = {in, n, 2, 0, 0};
    /* a window for back-references past `cap` is not needed: the caller never asks for more than cap + 1 bytes */
#define EMIT(byte) do { if (op < cap) out[op] = (uint8_t)(byte); ++op; if (op > cap) { *produced = op; return 0; } } while (0)
    for (;;) {
        if (!need(&b, 3)) { *produced = op; return 1; }
        const int last = (int)take(&b, 1), type = (int)take(&b, 2);
        if (type == 3) return 2;
        if (type == 0) {
            take(&b, b.bits & 7); /* to the byte boundary */
            if (!need(&b, 32)) { *produced = op; return 1; }
            const uint32_t l = take(&b, 16), nl = take(&b, 16);
            if ((l ^ 0xffff) != nl) return 2;
            for (uint32_t i = 0; i < l; ++i) {
                if (!need(&b, 8)) { *produced = op; return 1; }
                EMIT(take(&b, 8));
            }
        } else {
            huff hl, hd;
            uint8_t lengths[320];
            if (type == 1) {
                int i = 0;
                for (; i < 144; ++i) lengths[i] = 8;
                for (; i < 256; ++i) lengths[i] = 9;
                for (; i < 280; ++i) lengths[i] = 7;
                for (; i < 288; ++i) lengths[i] = 8;
                huff_build(&hl, lengths, 288);
                for (i = 0; i < 30; ++i) lengths[i] = 5;
                huff_build(&hd, lengths, 30);
            } else {
                if (!need(&b, 14)) { *produced = op; return 1; }
                const int nlen = (int)take(&b, 5) + 257, ndist = (int)take(&b, 5) + 1, ncode = (int)take(&b, 4) + 4;
                if (nlen > 286 || ndist > 30) return 2;
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; ++i) {
                    if (!need(&b, 3)) { *produced = op; return 1; }
                    cl[ORDER[i]] = (uint8_t)take(&b, 3);
                }
                huff hc;
                if (huff_build(&hc, cl, 19) != 0) return 2;
                int idx = 0;
                while (idx < nlen + ndist) {
                    const int sym = huff_decode(&b, &hc);
                    if (sym == -1) { *produced = op; return 1; }
                    if (sym < 0) return 2;
                    if (sym < 16) { lengths[idx++] = (uint8_t)sym; continue; }
                    int prev = 0, rep;
                    if (sym == 16) {
                        if (idx == 0) return 2;
                        prev = lengths[idx - 1];
                        if (!need(&b, 2)) { *produced = op; return 1; }
                        rep = 3 + (int)take(&b, 2);
                    } else if (sym == 17) {
                        if (!need(&b, 3)) { *produced = op; return 1; }
                        rep = 3 + (int)take(&b, 3);
                    } else {
                        if (!need(&b, 7)) { *produced = op; return 1; }
                        rep = 11 + (int)take(&b, 7);
                    }
                    if (idx + rep > nlen + ndist) return 2;
                    while (rep--) lengths[idx++] = (uint8_t)prev;
                }
                if (lengths[256] == 0) return 2;
                int rc = huff_build(&hl, lengths, nlen);
                if (rc < 0 || (rc > 0 && nlen - hl.count[0] != 1)) return 2;
                rc = huff_build(&hd, lengths + nlen, ndist);
                if (rc < 0 || (rc > 0 && ndist - hd.count[0] != 1)) return 2;
            }
            for (;;) {
                int sym = huff_decode(&b, &hl);
                if (sym == -1) { *produced = op; return 1; }
                if (sym < 0) return 2;
                if (sym < 256) { EMIT(sym); continue; }
                if (sym == 256) break;
                sym -= 257;
                if (sym >= 29) return 2;
                if (!need(&b, LEXT[sym])) { *produced = op; return 1; }
                int length = LBASE[sym] + (int)take(&b, LEXT[sym]);
                const int ds = huff_decode(&b, &hd);
                if (ds == -1) { *produced = op; return 1; }
                if (ds < 0 || ds >= 30) return 2;
                if (!need(&b, DEXT[ds])) { *produced = op; return 1; }
                const size_t dist = DBASE[ds] + take(&b, DEXT[ds]);
                if (dist > op) return 2;
                while (length--) EMIT(op - dist < cap ? out[op - dist] : 0);
            }
        }
        if (last) break;
    }
#undef EMIT
    *produced = op;
    take(&b, b.bits & 7);
    if (!need(&b, 32)) return 1; /* cut inside the checksum: every data byte arrived */
    uint32_t want = 0;
    for (int i = 0; i < 4; ++i) want = want << 8 | take(&b, 8);
    uint32_t a = 1, s = 0;
    for (size_t i = 0; i < op && i < cap; ++i) { a = (a + out[i]) % 65521; s = (s + a) % 65521; }
    return a + (s << 16) == want ? 0 : 2;
}

/* ---- defiltering (:1442-1533, :1721-1803) ----------------------------------------------------------------------------- */
static uint8_t paeth(int a, int b, int c) { /* :1442-1448 */
    const int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c);
    const int bc = pc < pb ? c : b;
    return (uint8_t)((pb < pa || pc < pa) ? bc : a);
}
ZO_API uint8_t zo_png_paeth(int a, int b, int c) { return paeth(a, b, c); }

static void defilter_row(int filter, uint8_t *cur, const uint8_t *prev, size_t n, int bpp) {
    switch (filter) {
    case 0: break;
    case 1:
        for (size_t i = (size_t)bpp; i < n; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
        break;
    case 2:
        if (prev) for (size_t i = 0; i < n; ++i) cur[i] = (uint8_t)(cur[i] + prev[i]);
        break;
    case 3:
        if (prev) {
            for (size_t i = 0; i < n; ++i) {
                const int left = i >= (size_t)bpp ? cur[i - bpp] : 0;
                cur[i] = (uint8_t)(cur[i] + ((left + prev[i]) >> 1));
            }
        } else {
            for (size_t i = (size_t)bpp; i < n; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp] / 2);
        }
        break;
    default:
        if (prev) {
            for (size_t i = 0; i < (size_t)bpp && i < n; ++i) cur[i] = (uint8_t)(cur[i] + prev[i]);
            for (size_t i = (size_t)bpp; i < n; ++i) cur[i] = (uint8_t)(cur[i] + paeth(cur[i - bpp], prev[i], prev[i - bpp]));
        } else {
            for (size_t i = (size_t)bpp; i < n; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
        }
    }
}
static int defilter_block(uint8_t *data, size_t row_bytes, uint32_t rows, int bpp) {
    uint8_t *prev = NULL;
    for (uint32_t y = 0; y < rows; ++y) {
        uint8_t *row = data + (size_t)y * (row_bytes + 1);
        if (row[0] > 4) return E_InvalidFilterType;
        defilter_row(row[0], row + 1, prev, row_bytes, bpp);
        prev = row + 1;
    }
    return E_OK;
}
static int defilter_scanlines(uint8_t *data, const zo_png_header *h) {
    const int bpp = bytes_per_pixel(h);
    if (h->interlace_method != 1) return defilter_block(data, scanline_bytes(h), h->height, bpp);
    size_t off = 0;
    for (int p = 0; p < 7; ++p) {
        uint32_t pw, ph;
        a7_dims(p, h->width, h->height, &pw, &ph);
        if (!pw || !ph) continue;
        const size_t rb = a7_scanline_bytes(pw, h);
        const int rc = defilter_block(data + off, rb, ph, bpp);
        if (rc) return rc;
        off += (rb + 1) * ph;
    }
    return E_OK;
}

/* ---- pixel extraction (:1855-2053); out = r, g, b, a -------------------------------------------------------------------- */
static int hi8(const uint8_t *p) { return p[0]; } /* readInt(u16, big) >> 8 */
static uint8_t sub_byte(const uint8_t *row, size_t row_len, size_t x, int bd, int *in_range) {
    const int ppb = 8 / bd;
    const uint8_t mask = (uint8_t)((1 << bd) - 1);
    const size_t byte = x / ppb;
    *in_range = byte < row_len;
    if (!*in_range) return 0;
    return (uint8_t)((row[byte] >> ((ppb - 1 - (int)(x % ppb)) * bd)) & mask);
}
static void extract_gray(const uint8_t *row, size_t n, size_t x, const zo_png_header *h, const uint8_t *trns, int trns_len, uint8_t px[4]) {
    uint8_t alpha = 255, v = 0;
    const int ga = h->color_type == 4;
    if (h->bit_depth == 8) {
        if (ga) { if (x * 2 + 1 < n) alpha = row[x * 2 + 1]; v = row[x * 2]; } else v = row[x];
    } else if (h->bit_depth == 16) {
        const size_t off = ga ? x * 4 : x * 2;
        if (off + 1 < n) {
            if (ga && off + 3 < n) alpha = (uint8_t)hi8(row + off + 2);
            v = (uint8_t)hi8(row + off);
        }
    } else {
        int ok;
        const uint8_t raw = sub_byte(row, n, x, h->bit_depth, &ok);
        v = ok ? (uint8_t)(raw * (255 / ((1 << h->bit_depth) - 1))) : 0;
    }
    if (h->color_type == 0 && trns && trns_len >= 2) {
        const uint8_t t = h->bit_depth == 16 ? trns[0] : trns[1];
        if (v == t) alpha = 0;
    }
    px[0] = px[1] = px[2] = v;
    px[3] = alpha;
}
static void extract_rgb(const uint8_t *row, size_t n, size_t x, const zo_png_header *h, const uint8_t *trns, int trns_len, uint8_t px[4]) {
    const size_t cs = h->bit_depth == 16 ? 2 : 1, total = cs * channels_of(h->color_type), off = x * total;
    if (off + total > n) { px[0] = px[1] = px[2] = 0; px[3] = 255; return; }
    px[0] = row[off]; px[1] = row[off + cs]; px[2] = row[off + 2 * cs]; /* the high byte of a 16-bit sample comes first */
    px[3] = 255;
    if (h->color_type == 6) { px[3] = row[off + 3 * cs]; return; }
    if (trns && trns_len >= 6) {
        const int k = h->bit_depth == 16 ? 0 : 1;
        if (px[0] == trns[k] && px[1] == trns[2 + k] && px[2] == trns[4 + k]) px[3] = 0;
    }
}
/* returns 0 when the index is outside the palette */
static int extract_palette(const uint8_t *row, size_t n, size_t x, const zo_png_header *h, const png_state *st, uint8_t px[4], int *row_ok) {
    int index, ok = 1;
    if (h->bit_depth == 8) { ok = x < n; index = ok ? row[x] : 0; }
    else index = sub_byte(row, n, x, h->bit_depth, &ok);
    *row_ok = ok;
    if (index >= st->palette_len) { px[0] = px[1] = px[2] = 0; px[3] = 255; return 0; }
    memcpy(px, st->palette[index], 3);
    px[3] = st->trns_len >= 0 && index < st->trns_len ? st->trns[index] : 255;
    return 1;
}
static void store_native(uint8_t *dst, int native, const uint8_t px[4], int from_color) {
    if (native == ZO_U8) dst[0] = from_color ? (uint8_t)((px[0] + px[1] + px[2]) / 3) : px[0];
    else if (native == ZO_RGB_U8) memcpy(dst, px, 3);
    else memcpy(dst, px, 4);
}

/* ---- toNativeImage (:801-1146) ------------------------------------------------------------------------------------------- */
ZO_API int zo_png_decode_native(const uint8_t *png, size_t len, const zo_png_limits *lim_in, zo_png_header *header_out, int *native_out,
                                uint8_t **pixels_out, int *truncated_out) {
    zo_png_limits lim;
    if (lim_in) lim = *lim_in; else zo_png_default_limits(&lim);
    png_state st;
    int rc = decode_chunks(png, len, &lim, &st);
    uint8_t *scan = NULL, *out = NULL;
    if (rc) goto done;
    const zo_png_header *h = &st.header;
    if (header_out) *header_out = *h;
    const size_t want = st.scan_data_bytes;
    scan = (uint8_t *)calloc(want + 1, 1);
    if (!scan) { rc = E_OutOfMemory; goto done; }
    size_t produced = 0;
    const int zrc = zlib_inflate(st.idat, st.idat_len, scan, want, &produced);
    if (zrc == 2) { rc = E_ReadFailed; goto done; }
    if (produced > want) { rc = E_ImageTooLarge; goto done; }
    if (produced < want) { /* cut or short stream: keep whole rows, zero the rest (:846-851) */
        st.truncated = 1;
        const size_t keep = complete_scan_prefix(produced, h);
        memset(scan + keep, 0, want - keep);
    }
    if ((rc = defilter_scanlines(scan, h))) goto done;

    const int ct = h->color_type, has_trns = st.trns_len >= 0;
    int native;
    if (h->interlace_method == 1) { /* :862-893: grey + alpha without tRNS comes out as Image(u8) on this branch */
        if (ct == 0 || ct == 4) native = has_trns ? ZO_RGBA_U8 : ZO_U8;
        else if (ct == 2) native = has_trns ? ZO_RGBA_U8 : ZO_RGB_U8;
        else if (ct == 6) native = ZO_RGBA_U8;
        else native = has_trns ? ZO_RGBA_U8 : ZO_RGB_U8;
    } else {
        if (ct == 0 || ct == 4) native = (ct == 4 || has_trns) ? ZO_RGBA_U8 : ZO_U8;
        else if (ct == 2) native = has_trns ? ZO_RGBA_U8 : ZO_RGB_U8;
        else if (ct == 6) native = ZO_RGBA_U8;
        else native = has_trns ? ZO_RGBA_U8 : ZO_RGB_U8;
    }
    const size_t ps = zo_pixel_size(native);
    out = (uint8_t *)calloc((size_t)h->width * h->height * ps + 1, 1);
    if (!out) { rc = E_OutOfMemory; goto done; }
    const uint8_t *trns = has_trns ? st.trns : NULL;
    if (h->interlace_method == 1) {
        size_t off = 0;
        for (int p = 0; p < 7; ++p) {
            uint32_t pw, ph;
            a7_dims(p, h->width, h->height, &pw, &ph);
            if (!pw || !ph) continue;
            const size_t rb = a7_scanline_bytes(pw, h);
            for (uint32_t py = 0; py < ph; ++py) {
                const uint8_t *row = scan + off + (size_t)py * (rb + 1) + 1;
                const uint32_t fy = A7[p][1] + py * A7[p][3];
                for (uint32_t pxi = 0; pxi < pw; ++pxi) {
                    const uint32_t fx = A7[p][0] + pxi * A7[p][2];
                    uint8_t px[4];
                    int ok;
                    if (ct == 0 || ct == 4) extract_gray(row, rb, pxi, h, trns, st.trns_len, px);
                    else if (ct == 3) extract_palette(row, rb, pxi, h, &st, px, &ok); /* bad indices fall back to black (:2038-2045) */
                    else extract_rgb(row, rb, pxi, h, trns, st.trns_len, px);
                    store_native(out + ((size_t)fy * h->width + fx) * ps, native, px, ct == 2 || ct == 3 || ct == 6);
                }
            }
            off += (rb + 1) * ph;
        }
    } else {
        const size_t rb = scanline_bytes(h);
        for (uint32_t y = 0; y < h->height && !rc; ++y) {
            const uint8_t *row = scan + (size_t)y * (rb + 1) + 1;
            for (uint32_t x = 0; x < h->width; ++x) {
                uint8_t px[4];
                if (ct == 0 || ct == 4) extract_gray(row, rb, x, h, trns, st.trns_len, px);
                else if (ct == 3) {
                    int ok;
                    const int found = extract_palette(row, rb, x, h, &st, px, &ok);
                    if (!ok) { rc = E_InvalidScanlineData; break; }
                    if (!found) { rc = E_InvalidPaletteIndex; break; } /* :1080, :1119 */
                } else extract_rgb(row, rb, x, h, trns, st.trns_len, px);
                store_native(out + ((size_t)y * h->width + x) * ps, native, px, 0);
            }
        }
        if (rc) goto done;
    }
    *native_out = native;
    *pixels_out = out;
    out = NULL;
    if (truncated_out) *truncated_out = st.truncated;
done:
    free(st.idat);
    free(scan);
    free(out);
    return rc;
}
ZO_API void zo_png_free(void *p) { free(p); }

/* The host half of a decode on its own: chunk layer, inflate with the recovery rules, de-filtering, and the palette-index check
 * of the non-interlaced path; FNV-1a over the de-filtered scan data (filter bytes included). */
ZO_API int zo_png_scan_hash(const uint8_t *png, size_t len, const zo_png_limits *lim_in, uint64_t *hash_out, int *truncated_out) {
    zo_png_limits lim;
    if (lim_in) lim = *lim_in; else zo_png_default_limits(&lim);
    png_state st;
    int rc = decode_chunks(png, len, &lim, &st);
    uint8_t *scan = NULL;
    if (rc) goto done;
    const zo_png_header *h = &st.header;
    const size_t want = st.scan_data_bytes;
    scan = (uint8_t *)calloc(want + 1, 1);
    if (!scan) { rc = E_OutOfMemory; goto done; }
    size_t produced = 0;
    const int zrc = zlib_inflate(st.idat, st.idat_len, scan, want, &produced);
    if (zrc == 2) { rc = E_ReadFailed; goto done; }
    if (produced > want) { rc = E_ImageTooLarge; goto done; }
    if (produced < want) {
        st.truncated = 1;
        memset(scan + complete_scan_prefix(produced, h), 0, want - complete_scan_prefix(produced, h));
    }
    if ((rc = defilter_scanlines(scan, h))) goto done;
    if (h->color_type == 3 && h->interlace_method != 1) {
        const size_t rb = scanline_bytes(h);
        for (uint32_t y = 0; y < h->height && !rc; ++y)
            for (uint32_t x = 0; x < h->width; ++x) {
                uint8_t px[4];
                int ok;
                if (!extract_palette(scan + (size_t)y * (rb + 1) + 1, rb, x, h, &st, px, &ok)) { rc = E_InvalidPaletteIndex; break; }
            }
        if (rc) goto done;
    }
    uint64_t hash = 1469598103934665603ull;
    for (size_t i = 0; i < want; ++i) { hash ^= scan[i]; hash *= 1099511628211ull; }
    *hash_out = hash;
    if (truncated_out) *truncated_out = st.truncated;
done:
    free(st.idat);
    free(scan);
    return rc;
}

/* decode() alone: header + the chunk layer's truncated flag (what png.decode returns before any inflate) */
ZO_API int zo_png_decode_chunks(const uint8_t *png, size_t len, const zo_png_limits *lim_in, zo_png_header *header_out, int *truncated_out,
                                int *palette_len, int *trns_len) {
    zo_png_limits lim;
    if (lim_in) lim = *lim_in; else zo_png_default_limits(&lim);
    png_state st;
    const int rc = decode_chunks(png, len, &lim, &st);
    if (!rc) {
        if (header_out) *header_out = st.header;
        if (truncated_out) *truncated_out = st.truncated;
        if (palette_len) *palette_len = st.palette_len;
        if (trns_len) *trns_len = st.trns_len;
    }
    free(st.idat);
    return rc;
}

/* ---- filtering (:1265-1294, :1535-1719) -------------------------------------------------------------------------------- */
static void filter_row(int filter, uint8_t *dst, const uint8_t *src, const uint8_t *prev, size_t n, int bpp) {
    for (size_t i = 0; i < n; ++i) {
        const int left = i >= (size_t)bpp ? src[i - bpp] : 0, above = prev ? prev[i] : 0, ul = prev && i >= (size_t)bpp ? prev[i - bpp] : 0;
        int pred;
        switch (filter) {
        case 0: pred = 0; break;
        case 1: pred = left; break;
        case 2: pred = above; break; /* first row: a copy (:1557-1565) */
        case 3: pred = (left + above) >> 1; break;
        default: pred = prev ? (i >= (size_t)bpp ? paeth(left, above, ul) : above) : left; /* :1585-1606 */
        }
        dst[i] = (uint8_t)(src[i] - pred);
    }
}
static uint32_t filter_cost(const uint8_t *p, size_t n) { /* :1621-1631 */
    uint32_t cost = 0;
    for (size_t i = 0; i < n; ++i) { const int v = (int8_t)p[i]; cost += (uint32_t)(v < 0 ? -v : v); }
    return cost;
}
static int select_best(const uint8_t *src, const uint8_t *prev, size_t n, int bpp, uint8_t *tmp) { /* :1634-1658 */
    int best = 0;
    uint32_t best_cost = 0xffffffffu;
    for (int f = 0; f < 5; ++f) {
        if (!prev && f >= 2) continue;
        filter_row(f, tmp, src, prev, n, bpp);
        const uint32_t c = filter_cost(tmp, n);
        if (c < best_cost) { best_cost = c; best = f; }
    }
    return best;
}
/* mode: -1 adaptive, 0..4 fixed. `filtered` holds rows * (row_bytes + 1) bytes. */
ZO_API int zo_png_filter(const uint8_t *raw, uint32_t rows, size_t row_bytes, int bpp, int mode, uint8_t *filtered) {
    uint8_t *tmp = (uint8_t *)malloc(row_bytes ? row_bytes : 1);
    if (!tmp) return E_OutOfMemory;
    const uint32_t sample_rate = rows > 512 ? 8 : 1;
    int last = 0;
    uint32_t streak = 0;
    for (uint32_t y = 0; y < rows; ++y) {
        const uint8_t *src = raw + (size_t)y * row_bytes, *prev = y ? src - row_bytes : NULL;
        uint8_t *dst = filtered + (size_t)y * (row_bytes + 1);
        int f = mode;
        if (mode < 0) { /* filterScanlinesAdaptive :1661-1719 */
            const int analyze = y % sample_rate == 0 || streak == 0 || y < 3 || y + 3 >= rows;
            if (analyze) {
                f = select_best(src, prev, row_bytes, bpp, tmp);
                if (f == last) streak = streak + 1 < sample_rate ? streak + 1 : sample_rate;
                else { streak = 0; last = f; }
            } else f = last;
        }
        dst[0] = (uint8_t)f;
        filter_row(f, dst + 1, src, prev, row_bytes, bpp);
    }
    free(tmp);
    return E_OK;
}

/* ---- a minimal writer (container per :1335-1398; IDAT in stored blocks) -------------------------------------------------- */
static size_t put_chunk(uint8_t *o, const char *type, const uint8_t *data, size_t n) {
    put_be32(o, (uint32_t)n);
    memcpy(o + 4, type, 4);
    if (n) memcpy(o + 8, data, n);
    put_be32(o + 8 + n, zo_png_crc(o + 4, n + 4));
    return n + 12;
}
ZO_API int zo_png_encode_stored(const zo_image *img, int mode, uint8_t **out, size_t *out_len) {
    if (!(img->pixel == ZO_U8 || img->pixel == ZO_RGB_U8 || img->pixel == ZO_RGBA_U8)) return E_InvalidColorType;
    const int bpp = (int)zo_pixel_size(img->pixel);
    const size_t rb = (size_t)img->cols * bpp, fl = (rb + 1) * img->rows;
    uint8_t *raw = (uint8_t *)malloc(rb * img->rows + 1), *filtered = (uint8_t *)malloc(fl + 1);
    const size_t nblocks = fl / 65535 + 1, zl = 2 + fl + nblocks * 5 + 4;
    uint8_t *z = (uint8_t *)malloc(zl), *o = (uint8_t *)malloc(8 + 25 + zl + 12 + 12);
    if (!raw || !filtered || !z || !o) { free(raw); free(filtered); free(z); free(o); return E_OutOfMemory; }
    for (uint32_t y = 0; y < img->rows; ++y) memcpy(raw + (size_t)y * rb, (const uint8_t *)img->data + (size_t)y * img->stride * bpp, rb);
    zo_png_filter(raw, img->rows, rb, bpp, mode, filtered);
    size_t zp = 0;
    z[zp++] = 0x78; z[zp++] = 0x01;
    uint32_t a = 1, s = 0;
    for (size_t i = 0; i < fl; ++i) { a = (a + filtered[i]) % 65521; s = (s + a) % 65521; }
    for (size_t off = 0, blk = 0; blk < nblocks; ++blk) {
        const size_t n = fl - off < 65535 ? fl - off : 65535;
        z[zp++] = blk + 1 == nblocks;
        z[zp++] = (uint8_t)n; z[zp++] = (uint8_t)(n >> 8); z[zp++] = (uint8_t)~n; z[zp++] = (uint8_t)(~n >> 8);
        memcpy(z + zp, filtered + off, n);
        zp += n; off += n;
    }
    put_be32(z + zp, s << 16 | a);
    zp += 4;
    size_t p = 0;
    memcpy(o, SIGNATURE, 8); p = 8;
    uint8_t ihdr[13] = {0};
    put_be32(ihdr, img->cols); put_be32(ihdr + 4, img->rows);
    ihdr[8] = 8; ihdr[9] = img->pixel == ZO_U8 ? 0 : (img->pixel == ZO_RGB_U8 ? 2 : 6);
    p += put_chunk(o + p, "IHDR", ihdr, 13);
    p += put_chunk(o + p, "IDAT", z, zp);
    p += put_chunk(o + p, "IEND", NULL, 0);
    free(raw); free(filtered); free(z);
    *out = o; *out_len = p;
    return E_OK;
}
