/*
 * oracle/jpeg.c — CPU restatement of the reference's JPEG decoder (src/codecs/jpeg.zig). TEST INFRASTRUCTURE ONLY (see zo.h).
 *
 * Follows the reference step by step, quirks included, because the parity target is what the reference produces:
 *   getInfo :77-179, decode (marker loop) :2035-2151, parseSOF / DHT / DQT / SOS / DRI :1314-1645, BitReader :1660-1736
 *   (restart markers are swallowed by the bit filler; a restart boundary then drops whatever was pre-fetched), readCode /
 *   readMagnitudeCoded / decodeAC :1196-1310, the progressive scan and its block decoder :1740-1930 ("non-interleaved" means
 *   one scan component whose id is 1; restart boundaries by MCU index), the baseline block scan :2397-2479, dequantisation
 *   :2482-2495, the stb-style integer IDCT :2204-2394 with the +128 level shift on component 0 only :2498-2515, the four
 *   chroma layouts :2518-2749 (4:4:4 in integer arithmetic; 4:2:2 / 4:1:1 / 4:2:0 with f32 bilinear chroma taps that never
 *   leave the MCU's own 8 x 8 chroma block, then Ycbcr(u8).to(.rgb)), rendering :2752-2784, toNativeImage :2786-2821 and
 *   loadFromBytes :2825-2851.
 * Pinned by the reference's own known answers (jpeg.zig:3028-3116: limits, the hand-built progressive stream and its cut /
 * capped variants) and cross-checked against an independent decoder (Pillow / libjpeg) within the tolerance two different
 * IDCT + upsampling designs allow (tests/test_oracle_jpeg.py).
 */
#include "zo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static const char *const JERR[] = {
    "ok", "InvalidJpegFile", "InvalidMarker", "InvalidSOF", "DuplicateSOF", "InvalidDHT", "InvalidDQT", "InvalidSOS", "InvalidDRI",
    "UnsupportedExtendedSequential", "UnsupportedLosslessJpeg", "UnsupportedJpegVariant", "UnsupportedArithmeticCoding",
    "Unsupported12BitPrecision", "Unsupported16BitPrecision", "UnsupportedPrecision", "UnsupportedComponentCount",
    "UnsupportedSamplingFactor", "UnsupportedHierarchicalJpeg", "InvalidComponentCount", "InvalidHuffmanTable", "InvalidQuantTable",
    "NoScanData", "UnexpectedEndOfData", "InvalidHuffmanCode", "MissingHuffmanTable", "MissingQuantTable", "InvalidDCCoefficient",
    "InvalidACCoefficient", "BlockStorageNotAllocated", "JpegDataTooLarge", "MarkerDataLimitExceeded", "BlockMemoryLimitExceeded",
    "ImageTooLarge", "MissingSOF", "EndOfStream", "OutOfMemory",
};
enum {
    J_OK, J_InvalidJpegFile, J_InvalidMarker, J_InvalidSOF, J_DuplicateSOF, J_InvalidDHT, J_InvalidDQT, J_InvalidSOS, J_InvalidDRI,
    J_UnsupportedExtendedSequential, J_UnsupportedLosslessJpeg, J_UnsupportedJpegVariant, J_UnsupportedArithmeticCoding,
    J_Unsupported12BitPrecision, J_Unsupported16BitPrecision, J_UnsupportedPrecision, J_UnsupportedComponentCount,
    J_UnsupportedSamplingFactor, J_UnsupportedHierarchicalJpeg, J_InvalidComponentCount, J_InvalidHuffmanTable, J_InvalidQuantTable,
    J_NoScanData, J_UnexpectedEndOfData, J_InvalidHuffmanCode, J_MissingHuffmanTable, J_MissingQuantTable, J_InvalidDCCoefficient,
    J_InvalidACCoefficient, J_BlockStorageNotAllocated, J_JpegDataTooLarge, J_MarkerDataLimitExceeded, J_BlockMemoryLimitExceeded,
    J_ImageTooLarge, J_MissingSOF, J_EndOfStream, J_OutOfMemory,
};
ZO_API const char *zo_jpeg_error_name(int code) { return code >= 0 && code < (int)(sizeof JERR / sizeof JERR[0]) ? JERR[code] : "?"; }

ZO_API void zo_jpeg_default_limits(zo_jpeg_limits *l) { /* jpeg.zig:19-33 */
    l->max_jpeg_bytes = l->max_marker_bytes = 100u * 1024 * 1024;
    l->max_width = l->max_height = 8192;
    l->max_pixels = 67108864ull;
    l->max_blocks = 1048576;
    l->max_scans = 64;
}
static int exceeds(uint64_t limit, uint64_t v) { return limit != 0 && v > limit; }

static const uint8_t ZIGZAG[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* ---- getInfo (:77-179): a forward reader hunting for the first SOFn ------------------------------------------------------ */
ZO_API int zo_jpeg_info(const uint8_t *d, size_t len, const zo_jpeg_limits *lim_in, zo_jpeg_header *out) {
    zo_jpeg_limits lim;
    if (lim_in) lim = *lim_in; else zo_jpeg_default_limits(&lim);
    size_t pos = 0, bytes_read = 0, markers = 0;
#define TAKE(var) do { if (pos >= len) return J_EndOfStream; (var) = d[pos++]; } while (0)
    if (len < 2) return J_EndOfStream;
    if (d[0] != 0xFF || d[1] != 0xD8) return J_InvalidJpegFile;
    pos = bytes_read = 2;
    for (;;) {
        int byte;
        for (;;) {
            TAKE(byte);
            bytes_read += 1;
            if (bytes_read > lim.max_jpeg_bytes) return J_ImageTooLarge;
            if (byte == 0xFF) break;
        }
        int m;
        TAKE(m);
        bytes_read += 1;
        while (m == 0xFF) {
            TAKE(m);
            bytes_read += 1;
            if (bytes_read > lim.max_jpeg_bytes) return J_ImageTooLarge;
        }
        if (m == 0x00) continue;
        if (++markers > 10000) return J_ImageTooLarge;
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7) || m == 0xD8) continue;
        if (m == 0xD9) return J_MissingSOF;
        if (len - pos < 2) return J_EndOfStream;
        const unsigned length = (unsigned)d[pos] << 8 | d[pos + 1];
        pos += 2; bytes_read += 2;
        if (length < 2) return J_InvalidMarker;
        const int is_sof = m == 0xC0 || m == 0xC1 || m == 0xC2 || m == 0xC3 || m == 0xC5 || m == 0xC6 || m == 0xC7 || m == 0xC9 || m == 0xCA ||
                           m == 0xCB || m == 0xCD || m == 0xCE || m == 0xCF;
        if (is_sof) {
            const unsigned payload = length - 2;
            if (payload < 6) return J_InvalidSOF;
            if (bytes_read + payload > lim.max_jpeg_bytes) return J_ImageTooLarge;
            if (len - pos < 6) return J_EndOfStream;
            memset(out, 0, sizeof *out);
            out->precision = d[pos];
            out->height = (uint32_t)d[pos + 1] << 8 | d[pos + 2];
            out->width = (uint32_t)d[pos + 3] << 8 | d[pos + 4];
            out->num_components = d[pos + 5];
            pos += 6;
            out->subsampling = -1;
            if (out->num_components == 3 && payload - 6 >= 9) {
                if (len - pos < 9) return J_EndOfStream;
                const uint8_t f0 = d[pos + 1], f1 = d[pos + 4], f2 = d[pos + 7];
                if (f1 == 0x11 && f2 == 0x11) out->subsampling = f0 == 0x11 ? 0 : (f0 == 0x21 ? 1 : (f0 == 0x22 ? 2 : -1));
            }
            out->progressive = m == 0xC2;
            return J_OK;
        }
        const unsigned skip = length - 2;
        if (bytes_read + skip > lim.max_jpeg_bytes) return J_ImageTooLarge;
        const size_t adv = len - pos < skip ? len - pos : skip;
        pos += adv; bytes_read += adv;
    }
#undef TAKE
}

/* ---- decoder state ---------------------------------------------------------------------------------------------------------- */
typedef struct huff_table {
    int present;
    uint8_t fast_table[512], fast_size[512];
    int32_t max_code[17];
    uint16_t min_code[17], val_ptr[17];
    uint8_t huffval[256];
} huff_table;
typedef struct bit_reader { const uint8_t *data; size_t len, byte_pos; uint64_t buffer; int count; } bit_reader;
typedef struct component { uint8_t id, h, v, tq; } component;
typedef struct scan_component { uint8_t id, dc, ac; } scan_component;
typedef struct scan_info { scan_component comp[4]; int n, ss, se, ah, al; } scan_info;
typedef struct jstate {
    zo_jpeg_header header;
    component comp[4];
    huff_table dc[4], ac[4];
    int have_q[4];
    uint16_t q[4][64];
    scan_info baseline_scan;
    unsigned restart_interval;
    bit_reader br;
    unsigned block_width, block_height, block_width_actual, block_height_actual;
    size_t nblocks;
    int32_t (*blocks)[4][64];
    uint8_t (*rgb)[3][64];
    int32_t dc_pred[4];
    uint32_t skip_count;
    int scan_limit_reached;
} jstate;

/* BitReader (:1660-1736) */
static int br_fill(bit_reader *b, int n) {
    while (b->count <= 56 && b->count < n) {
        if (b->byte_pos >= b->len) return J_UnexpectedEndOfData;
        uint64_t cur = b->data[b->byte_pos++];
        if (cur == 0xFF) {
            for (;;) {
                if (b->byte_pos >= b->len) return J_UnexpectedEndOfData;
                const uint8_t next = b->data[b->byte_pos++];
                if (next == 0x00) break;
                if (next == 0xFF) continue;
                if (next >= 0xD0 && next <= 0xD7) {
                    if (b->byte_pos >= b->len) return J_UnexpectedEndOfData;
                    cur = b->data[b->byte_pos++];
                    if (cur == 0xFF) continue;
                    break;
                }
                b->byte_pos -= 2; /* a real marker: stop here */
                return J_UnexpectedEndOfData;
            }
        }
        b->buffer |= cur << (56 - b->count);
        b->count += 8;
    }
    return J_OK;
}
static int br_peek(bit_reader *b, int n, uint32_t *out) {
    if (n == 0) { *out = 0; return J_OK; }
    const int rc = br_fill(b, n);
    if (rc) return rc;
    *out = (uint32_t)(b->buffer >> (64 - n));
    return J_OK;
}
static void br_consume(bit_reader *b, int n) {
    if (n == 0) return;
    b->buffer <<= n;
    b->count -= n;
}
static int br_get(bit_reader *b, int n, uint32_t *out) {
    const int rc = br_peek(b, n, out);
    if (rc) return rc;
    br_consume(b, n);
    return J_OK;
}
static void br_flush(bit_reader *b) { b->buffer = 0; b->count = 0; }

/* readCode (:1196-1236) */
static int read_code(jstate *s, const huff_table *t, int *sym) {
    uint32_t fast_index = 0;
    if (br_peek(&s->br, 9, &fast_index)) fast_index = 0;
    if (s->br.count >= 9) {
        const uint8_t v = t->fast_table[fast_index];
        if (v != 255) {
            br_consume(&s->br, t->fast_size[fast_index]);
            *sym = v;
            return J_OK;
        }
    }
    uint32_t code = 0;
    int length = 0;
    if (s->br.count >= 9) {
        br_consume(&s->br, 9);
        code = fast_index;
        length = 9;
    }
    while (length < 16) {
        uint32_t bit;
        const int rc = br_get(&s->br, 1, &bit);
        if (rc) return rc == J_UnexpectedEndOfData ? rc : J_InvalidHuffmanCode;
        code = (code << 1 | bit) & 0xffff;
        length += 1;
        if ((int32_t)code <= t->max_code[length]) {
            /* reachable below min_code only through a table that holds symbol 255, which the fast table reads as
             * "empty"; the reference asserts here (:1238) and its index would be out of range: an error, not a read */
            if (code < t->min_code[length]) return J_InvalidHuffmanCode;
            *sym = t->huffval[(size_t)t->val_ptr[length] + code - t->min_code[length]];
            return J_OK;
        }
    }
    return J_InvalidHuffmanCode;
}
/* readMagnitudeCoded (:1239-1252) */
static int read_magnitude(jstate *s, int magnitude, int32_t *out) {
    if (magnitude == 0) { *out = 0; return J_OK; }
    uint32_t bits;
    const int rc = br_peek(&s->br, magnitude, &bits);
    if (rc) return rc;
    br_consume(&s->br, magnitude);
    int32_t coeff = (int32_t)bits;
    if (coeff < (int32_t)1 << (magnitude - 1)) coeff -= ((int32_t)1 << magnitude) - 1;
    *out = coeff;
    return J_OK;
}
/* decodeAC (:1255-1310) */
static int decode_ac(jstate *s, const huff_table *t, int32_t *block) {
    int k = 1;
    while (k < 64) {
        int symbol, rc;
        if ((rc = read_code(s, t, &symbol))) return rc;
        if (symbol == 0) {
            while (k < 64) block[ZIGZAG[k++]] = 0;
            return J_OK;
        }
        const int run = symbol >> 4, size = symbol & 0x0F;
        if (size == 0) {
            if (run != 15) return J_InvalidACCoefficient;
            for (int i = 0; i < 16 && k < 64; ++i) block[ZIGZAG[k++]] = 0;
        } else {
            for (int i = 0; i < run && k < 64; ++i) block[ZIGZAG[k++]] = 0;
            if (k >= 64) break;
            int32_t value;
            if ((rc = read_magnitude(s, size, &value))) return rc;
            block[ZIGZAG[k++]] = value;
        }
    }
    return J_OK;
}

/* parseSOF (:1314-1442) */
static int parse_sof(jstate *s, const uint8_t *d, size_t n, int progressive, const zo_jpeg_limits *lim) {
    if (s->blocks) return J_DuplicateSOF;
    s->header.progressive = progressive;
    if (n < 6) return J_InvalidSOF;
    s->header.precision = d[0];
    if (d[0] == 12) return J_Unsupported12BitPrecision;
    if (d[0] == 16) return J_Unsupported16BitPrecision;
    if (d[0] != 8) return J_UnsupportedPrecision;
    s->header.height = (uint32_t)d[1] << 8 | d[2];
    s->header.width = (uint32_t)d[3] << 8 | d[4];
    s->header.num_components = d[5];
    if (s->header.width == 0 || s->header.height == 0) return J_InvalidSOF;
    if (exceeds(lim->max_width, s->header.width) || exceeds(lim->max_height, s->header.height)) return J_ImageTooLarge;
    const int nc = d[5];
    if (nc == 4) return J_UnsupportedComponentCount;
    if (nc != 1 && nc != 3) return J_InvalidComponentCount;
    size_t pos = 6;
    int max_h = 0, max_v = 0;
    for (int i = 0; i < nc; ++i) {
        if (pos + 3 > n) return J_InvalidSOF;
        s->comp[i] = (component){d[pos], (uint8_t)(d[pos + 1] >> 4), (uint8_t)(d[pos + 1] & 0x0F), d[pos + 2]};
        if (s->comp[i].h > max_h) max_h = s->comp[i].h;
        if (s->comp[i].v > max_v) max_v = s->comp[i].v;
        pos += 3;
    }
    if (max_h > 4 || max_v > 4) return J_UnsupportedSamplingFactor;
    /* a one-component frame whose sampling byte has a zero nibble: the reference goes on to divide by the MCU size (:1401-1404,
     * a panic); the restatement has to answer something, and answers with the error the three-component check gives */
    if (max_h == 0 || max_v == 0) return J_UnsupportedSamplingFactor;
    s->header.subsampling = -1;
    if (nc == 3) {
        const component *c = s->comp;
        if (c[1].h != c[2].h || c[1].v != c[2].v) return J_InvalidComponentCount;
        const int chroma11 = c[1].h == 1 && c[1].v == 1;
        const int ok = chroma11 && ((c[0].h == 1 && c[0].v == 1) || (c[0].h == 2 && c[0].v == 2) || (c[0].h == 2 && c[0].v == 1) || (c[0].h == 4 && c[0].v == 1));
        if (!ok) return J_UnsupportedSamplingFactor;
    }
    const uint32_t mcu_w = 8u * max_h, mcu_h = 8u * max_v;
    const uint32_t wa = (s->header.width + mcu_w - 1) / mcu_w * mcu_w, ha = (s->header.height + mcu_h - 1) / mcu_h * mcu_h;
    s->block_width = (s->header.width + 7) / 8;
    s->block_height = (s->header.height + 7) / 8;
    s->block_width_actual = (wa + 7) / 8;
    s->block_height_actual = (ha + 7) / 8;
    const uint64_t total = (uint64_t)wa * ha;
    if (exceeds(lim->max_pixels, total)) return J_ImageTooLarge;
    const uint64_t nblocks = total / 64;
    if (exceeds(lim->max_blocks, nblocks)) return J_BlockMemoryLimitExceeded;
    s->nblocks = (size_t)nblocks;
    s->blocks = calloc(s->nblocks ? s->nblocks : 1, sizeof *s->blocks);
    s->rgb = calloc(s->nblocks ? s->nblocks : 1, sizeof *s->rgb);
    if (!s->blocks || !s->rgb) return J_OutOfMemory;
    return J_OK;
}
/* parseDHT (:1445-1540) */
static int parse_dht(jstate *s, const uint8_t *d, size_t n) {
    if (n == 0) return J_InvalidDHT;
    size_t pos = 0;
    while (pos < n) {
        if (pos + 17 > n) return J_InvalidDHT;
        const int info = d[pos], cls = (info >> 4) & 1, id = info & 3;
        pos += 1;
        const uint8_t *bits = d + pos;
        pos += 16;
        unsigned total = 0;
        for (int i = 0; i < 16; ++i) total += bits[i];
        if (total > 256) return J_InvalidHuffmanTable;
        if (pos + total > n) return J_InvalidDHT;
        huff_table t;
        memset(&t, 0, sizeof t);
        memcpy(t.huffval, d + pos, total);
        pos += total;
        memset(t.fast_table, 255, sizeof t.fast_table);
        for (int i = 0; i < 17; ++i) t.max_code[i] = -1;
        unsigned code = 0, idx = 0;
        for (int i = 0; i < 16; ++i) {
            const int len = i + 1, count = bits[i];
            if (count > 0) { t.val_ptr[len] = (uint16_t)idx; t.min_code[len] = (uint16_t)code; }
            for (int j = 0; j < count; ++j) {
                if (code == (1u << (i + 1)) - 1) return J_InvalidHuffmanTable;
                const uint8_t byte = t.huffval[idx++];
                if (len <= 9) {
                    const unsigned first = (code << (9 - len)) & 0xffff, num = 1u << (9 - len);
                    for (unsigned k = 0; k < num; ++k) { t.fast_table[first + k] = byte; t.fast_size[first + k] = (uint8_t)len; }
                }
                code = (code + 1) & 0xffff;
            }
            if (count > 0) t.max_code[len] = (int32_t)code - 1;
            code = (code << 1) & 0xffff;
        }
        t.present = 1;
        if (cls == 0) s->dc[id] = t; else s->ac[id] = t;
    }
    return J_OK;
}
/* parseDQT (:1543-1583) */
static int parse_dqt(jstate *s, const uint8_t *d, size_t n) {
    if (n == 0) return J_InvalidDQT;
    size_t pos = 0;
    while (pos < n) {
        const int info = d[pos], precision = (info >> 4) & 0x0F, id = info & 3;
        pos += 1;
        const size_t es = precision == 0 ? 1 : 2;
        if (pos + 64 * es > n) return J_InvalidDQT;
        for (int i = 0; i < 64; ++i) s->q[id][ZIGZAG[i]] = es == 1 ? d[pos + i] : (uint16_t)((unsigned)d[pos + 2 * i] << 8 | d[pos + 2 * i + 1]);
        pos += 64 * es;
        s->have_q[id] = 1;
    }
    return J_OK;
}
/* parseSOS (:1586-1638) */
static int parse_sos(jstate *s, const uint8_t *d, size_t n, scan_info *si) {
    if (n < 6) return J_InvalidSOS;
    const int nc = d[0];
    if (!s->header.progressive && nc != s->header.num_components) return J_InvalidSOS;
    if (s->header.progressive && (nc == 0 || nc > s->header.num_components)) return J_InvalidSOS;
    size_t pos = 1;
    si->n = nc;
    for (int i = 0; i < nc; ++i) {
        if (pos + 2 > n) return J_InvalidSOS;
        if (i < 4) si->comp[i] = (scan_component){d[pos], (uint8_t)(d[pos + 1] >> 4), (uint8_t)(d[pos + 1] & 0x0F)};
        pos += 2;
    }
    if (pos + 3 > n) return J_InvalidSOS;
    const int ss = d[pos], se = d[pos + 1], approx = d[pos + 2];
    if (!s->header.progressive) {
        if (ss != 0 || se != 63 || approx != 0) return J_InvalidSOS;
    } else {
        if (ss > 63 || se > 63) return J_InvalidSOS;
        if (se < ss) return J_InvalidSOS;
        const int any_zero = ss == 0 || se == 0, both_zero = ss == 0 && se == 0;
        if (any_zero && !both_zero) return J_InvalidSOS;
    }
    si->ss = ss; si->se = se; si->ah = (approx >> 4) & 0x0F; si->al = approx & 0x0F;
    return J_OK;
}

/* decodeBlockProgressive (:1816-1930) */
static int decode_block_progressive(jstate *s, const scan_info *si, const scan_component *sc, int32_t *block, int32_t *dc_pred, uint32_t *skips) {
    int rc;
    uint32_t u;
    if (si->ss == 0) {
        const huff_table *t = &s->dc[sc->dc & 3];
        if (sc->dc > 3 || !t->present) return J_MissingHuffmanTable;
        if (si->ah == 0) {
            int mag;
            if ((rc = read_code(s, t, &mag))) return rc;
            if (mag > 11) return J_InvalidDCCoefficient;
            int32_t diff;
            if ((rc = read_magnitude(s, mag, &diff))) return rc;
            const int32_t v = diff + *dc_pred;
            *dc_pred = v;
            block[0] = (int32_t)((uint32_t)v << si->al);
        } else {
            if ((rc = br_get(&s->br, 1, &u))) return rc;
            block[0] += (int32_t)(u << si->al);
        }
        return J_OK;
    }
    const huff_table *t = &s->ac[sc->ac & 3];
    if (sc->ac > 3 || !t->present) return J_MissingHuffmanTable;
    int ac = si->ss;
    if (si->ah == 0) {
        if (*skips == 0) {
            while (ac <= si->se && ac < 64) {
                int32_t coeff = 0;
                int sym;
                if ((rc = read_code(s, t, &sym))) return rc;
                const int run = sym >> 4, mag = sym & 0x0F;
                if (mag == 0) {
                    if (run < 15) {
                        if ((rc = br_get(&s->br, run, &u))) return rc;
                        *skips = (1u << run) + u;
                        break;
                    }
                } else {
                    if (mag > 10) return J_InvalidACCoefficient;
                    if ((rc = read_magnitude(s, mag, &coeff))) return rc;
                }
                for (int i = 0; i < run && ac < 64; ++i) block[ZIGZAG[ac++]] = 0;
                if (ac >= 64) break;
                block[ZIGZAG[ac++]] = (int32_t)((uint32_t)coeff << si->al);
            }
        }
        if (*skips > 0) {
            *skips -= 1;
            while (ac <= si->se && ac < 64) block[ZIGZAG[ac++]] = 0;
        }
        return J_OK;
    }
    const int32_t bit = (int32_t)1 << si->al;
    if (*skips == 0) {
        while (ac <= si->se && ac < 64) {
            int32_t coeff = 0;
            int sym;
            if ((rc = read_code(s, t, &sym))) return rc;
            int run = sym >> 4;
            const int mag = sym & 0x0F;
            if (mag == 0) {
                if (run < 15) {
                    *skips = 1u << run;
                    if ((rc = br_get(&s->br, run, &u))) return rc;
                    *skips += u;
                    break;
                }
            } else {
                if ((rc = br_get(&s->br, 1, &u))) return rc;
                coeff = u == 1 ? bit : -bit;
            }
            while (ac <= si->se && ac < 64) {
                int32_t *c = &block[ZIGZAG[ac]];
                if (*c == 0) {
                    if (run > 0) { run -= 1; ac += 1; }
                    else { *c = coeff; ac += 1; break; }
                } else {
                    if ((rc = br_get(&s->br, 1, &u))) return rc;
                    if (u != 0) *c += *c > 0 ? bit : -bit;
                    ac += 1;
                }
            }
        }
    }
    if (*skips > 0) {
        for (; ac <= si->se && ac < 64; ++ac) {
            int32_t *c = &block[ZIGZAG[ac]];
            if (*c != 0) {
                if ((rc = br_get(&s->br, 1, &u))) return rc;
                if (u != 0) *c += *c > 0 ? bit : -bit;
            }
        }
        *skips -= 1;
    }
    return J_OK;
}

static void max_factors(const jstate *s, int *mh, int *mv) {
    *mh = *mv = 1;
    for (int i = 0; i < s->header.num_components; ++i) {
        if (s->comp[i].h > *mh) *mh = s->comp[i].h;
        if (s->comp[i].v > *mv) *mv = s->comp[i].v;
    }
}
/* performProgressiveScan (:1740-1813) */
static int progressive_scan(jstate *s, const scan_info *si) {
    if (!s->blocks) return J_BlockStorageNotAllocated;
    uint32_t skips = 0;
    const int nonint = si->n == 1 && si->comp[0].id == 1;
    int mh, mv;
    max_factors(s, &mh, &mv);
    const unsigned y_step = nonint ? 1 : mv, x_step = nonint ? 1 : mh;
    for (unsigned y = 0; y < s->block_height; y += y_step) {
        for (unsigned x = 0; x < s->block_width; x += x_step) {
            const size_t mcu_id = (size_t)y * s->block_width_actual + x;
            if (s->restart_interval != 0 && mcu_id % ((size_t)s->restart_interval * y_step * x_step) == 0) {
                br_flush(&s->br);
                memset(s->dc_pred, 0, sizeof s->dc_pred);
                skips = 0;
            }
            for (int index = 0; index < si->n; ++index) {
                const scan_component *sc = &si->comp[index];
                size_t ci = 0;
                unsigned vmax = 0, hmax = 0; /* an id that matches no frame component leaves these undefined in the reference; zero here */
                for (int i = 0; i < s->header.num_components; ++i)
                    if (s->comp[i].id == sc->id) { ci = i; vmax = nonint ? 1 : s->comp[i].v; hmax = nonint ? 1 : s->comp[i].h; break; }
                for (unsigned v = 0; v < vmax; ++v)
                    for (unsigned h = 0; h < hmax; ++h) {
                        const size_t block_id = (size_t)(y + v) * s->block_width_actual + (x + h);
                        if (block_id >= s->nblocks) continue;
                        (void)br_fill(&s->br, 24);
                        const int rc = decode_block_progressive(s, si, sc, s->blocks[block_id][ci], &s->dc_pred[ci], &skips);
                        if (rc == J_UnexpectedEndOfData) return J_OK; /* truncated scan: keep what was decoded */
                        if (rc) return rc;
                    }
            }
        }
    }
    if (si->ss != 0) s->skip_count = skips;
    return J_OK;
}
/* performBlockScan (:2397-2479) with decodeBlockBaseline (:1933-1952) */
static int baseline_scan(jstate *s) {
    if (!s->blocks) return J_BlockStorageNotAllocated;
    const scan_info *si = &s->baseline_scan;
    int mh, mv;
    max_factors(s, &mh, &mv);
    const int nonint = si->n == 1 && si->comp[0].id == 1;
    const unsigned y_step = nonint ? 1 : mv, x_step = nonint ? 1 : mh;
    int32_t pred[4] = {0, 0, 0, 0};
    uint32_t since = 0;
    for (unsigned y = 0; y < s->block_height; y += y_step) {
        for (unsigned x = 0; x < s->block_width; x += x_step) {
            if (s->restart_interval != 0 && since == s->restart_interval) {
                memset(pred, 0, sizeof pred);
                since = 0;
                br_flush(&s->br);
            }
            for (int index = 0; index < si->n; ++index) {
                const scan_component *sc = &si->comp[index];
                size_t ci = 0;
                unsigned vmax = 0, hmax = 0;
                for (int i = 0; i < s->header.num_components; ++i)
                    if (s->comp[i].id == sc->id) { ci = i; vmax = nonint ? 1 : s->comp[i].v; hmax = nonint ? 1 : s->comp[i].h; break; }
                for (unsigned v = 0; v < vmax; ++v)
                    for (unsigned h = 0; h < hmax; ++h) {
                        const unsigned ax = x + h, ay = y + v;
                        int32_t tmp[64];
                        int32_t *block = (ay < s->block_height && ax < s->block_width) ? s->blocks[(size_t)ay * s->block_width_actual + ax][ci] : tmp;
                        (void)br_fill(&s->br, 24);
                        memset(block, 0, 64 * sizeof *block);
                        int rc, sym;
                        const huff_table *dt = &s->dc[sc->dc & 3], *at = &s->ac[sc->ac & 3];
                        if (sc->dc > 3 || !dt->present) return J_MissingHuffmanTable;
                        rc = read_code(s, dt, &sym);
                        if (!rc && sym > 11) return J_InvalidDCCoefficient;
                        int32_t diff = 0;
                        if (!rc) rc = read_magnitude(s, sym, &diff);
                        if (!rc) {
                            pred[ci] += diff;
                            block[0] = pred[ci];
                            if (sc->ac > 3 || !at->present) return J_MissingHuffmanTable;
                            rc = decode_ac(s, at, block);
                        }
                        if (rc == J_UnexpectedEndOfData) return J_OK;
                        if (rc) return rc;
                    }
            }
            since += 1;
        }
    }
    return J_OK;
}

/* findScanEnd (:1955-1978) */
static size_t find_scan_end(const uint8_t *d, size_t len, size_t start) {
    size_t e = start;
    while (e + 1 < len) {
        if (d[e] == 0xFF) {
            const uint8_t nb = d[e + 1];
            if (nb == 0x00 || (nb >= 0xD0 && nb <= 0xD7)) { e += 2; continue; }
            break;
        }
        e += 1;
    }
    return e;
}

static int known_marker(int m) { /* Marker.fromBytes (:1045-1105): the enum's members */
    return (m >= 0xC0 && m <= 0xC4) || m == 0xCC || (m >= 0xD0 && m <= 0xDF) || (m >= 0xE0 && m <= 0xEF) || m == 0xFE;
}

/* decode (:2035-2151) */
static int decode_stream(jstate *s, const uint8_t *d, size_t len, const zo_jpeg_limits *lim) {
    if (len < 2 || d[0] != 0xFF || d[1] != 0xD8) return J_InvalidJpegFile;
    if (exceeds(lim->max_jpeg_bytes, len)) return J_JpegDataTooLarge;
    size_t pos = 2, marker_bytes = 0, scans = 0;
#define ACCUM(n) do { marker_bytes += (n); if (lim->max_marker_bytes != 0 && marker_bytes > lim->max_marker_bytes) return J_MarkerDataLimitExceeded; } while (0)
    while (pos + 1 < len) {
        if (d[pos] != 0xFF) return J_InvalidMarker;
        const int m = d[pos + 1];
        if (!known_marker(m)) {
            pos += 2;
            if (pos + 2 > len) break;
            const unsigned length = (unsigned)d[pos] << 8 | d[pos + 1];
            if (length < 2) return J_InvalidMarker;
            pos += length;
            continue;
        }
        if (m == 0xD8) { pos += 2; continue; }
        if (m == 0xD9) break;
        if (m == 0xC1) return J_UnsupportedExtendedSequential;
        if (m == 0xC3) return J_UnsupportedLosslessJpeg;
        if (m == 0xCC) return J_UnsupportedArithmeticCoding;
        if (m == 0xDE) return J_UnsupportedHierarchicalJpeg;
        if (m == 0xDC) return J_UnsupportedJpegVariant;
        if (m == 0xC0 || m == 0xC2 || m == 0xC4 || m == 0xDB || m == 0xDD) { /* readMarkerPayload (:2020-2033) */
            if (pos + 4 > len) return J_UnexpectedEndOfData;
            const unsigned length = (unsigned)d[pos + 2] << 8 | d[pos + 3];
            if (length < 2) return J_InvalidMarker;
            const size_t end = pos + 2 + length;
            if (end > len) return J_InvalidMarker;
            ACCUM(length);
            const uint8_t *p = d + pos + 4;
            const size_t n = end - (pos + 4);
            pos = end;
            int rc;
            if (m == 0xC0 || m == 0xC2) rc = parse_sof(s, p, n, m == 0xC2, lim);
            else if (m == 0xC4) rc = parse_dht(s, p, n);
            else if (m == 0xDB) rc = parse_dqt(s, p, n);
            else { if (n != 2) return J_InvalidDRI; s->restart_interval = (unsigned)p[0] << 8 | p[1]; rc = J_OK; }
            if (rc) return rc;
            continue;
        }
        if (m == 0xDA) {
            if (exceeds(lim->max_scans, scans + 1)) { s->scan_limit_reached = 1; break; }
            scans += 1;
            /* processScanMarker (:1987-2018) */
            if (pos + 4 > len) return J_UnexpectedEndOfData;
            const unsigned hl = (unsigned)d[pos + 2] << 8 | d[pos + 3];
            if (hl < 2) return J_InvalidMarker;
            const size_t end = pos + 2 + hl;
            if (end > len) return J_InvalidMarker;
            scan_info si;
            memset(&si, 0, sizeof si);
            int rc = parse_sos(s, d + pos + 4, end - (pos + 4), &si);
            if (rc) return rc;
            const size_t scan_end = find_scan_end(d, len, end);
            s->br = (bit_reader){d + end, scan_end - end, 0, 0, 0};
            if (!s->header.progressive) {
                s->baseline_scan = si;
                ACCUM(scan_end - pos);
                return J_OK; /* the block scan runs in toNativeImage */
            }
            if ((rc = progressive_scan(s, &si))) return rc;
            ACCUM(scan_end - pos);
            pos = scan_end;
            continue;
        }
        /* APPn, COM, and the enum's remaining members (RSTn, EXP): skipped by their length field (:2119-2139) */
        if (pos + 4 > len) break;
        const unsigned length = (unsigned)d[pos + 2] << 8 | d[pos + 3];
        ACCUM(length);
        pos += 2 + length;
    }
#undef ACCUM
    return s->header.progressive ? J_OK : J_NoScanData;
}

/* ---- IDCT (:2204-2394) ------------------------------------------------------------------------------------------------------ */
#define F2F(x) ((int32_t)((x) * 4096.0 + ((x) < 0 ? -0.5 : 0.5)))
static void idct1d(const int32_t s[8], int32_t x[4], int32_t t[4]) {
    int32_t p2 = s[2], p3 = s[6];
    int32_t p1 = (p2 + p3) * F2F(0.5411961);
    int32_t t2 = p1 + p3 * F2F(-1.847759065), t3 = p1 + p2 * F2F(0.765366865);
    p2 = s[0]; p3 = s[4];
    int32_t t0 = (p2 + p3) * 4096, t1 = (p2 - p3) * 4096;
    x[0] = t0 + t3; x[3] = t0 - t3; x[1] = t1 + t2; x[2] = t1 - t2;
    t0 = s[7]; t1 = s[5]; t2 = s[3]; t3 = s[1];
    p3 = t0 + t2;
    int32_t p4 = t1 + t3;
    p1 = t0 + t3; p2 = t1 + t2;
    const int32_t p5 = (p3 + p4) * F2F(1.175875602);
    t0 = t0 * F2F(0.298631336); t1 = t1 * F2F(2.053119869); t2 = t2 * F2F(3.072711026); t3 = t3 * F2F(1.501321110);
    p1 = p5 + p1 * F2F(-0.899976223); p2 = p5 + p2 * F2F(-2.562915447);
    p3 = p3 * F2F(-1.961570560); p4 = p4 * F2F(-0.390180644);
    t[3] = t3 + p1 + p4; t[2] = t2 + p2 + p3; t[1] = t1 + p2 + p4; t[0] = t0 + p1 + p3;
}
ZO_API void zo_jpeg_idct8x8(int32_t block[64]) {
    int any = 0;
    for (int i = 1; i < 64; ++i) any |= block[i];
    if (!any) { /* all-AC-zero: (dc + 4) >> 3 (:2259-2266) */
        const int32_t v = (block[0] + 4) >> 3;
        for (int i = 0; i < 64; ++i) block[i] = v;
        return;
    }
    int32_t tmp[64];
    for (int c = 0; c < 8; ++c) { /* pass 1: down each column, descale by 10 */
        int32_t s[8], x[4], t[4];
        for (int r = 0; r < 8; ++r) s[r] = block[r * 8 + c];
        idct1d(s, x, t);
        for (int k = 0; k < 4; ++k) x[k] += 512;
        tmp[0 * 8 + c] = (x[0] + t[3]) >> 10; tmp[1 * 8 + c] = (x[1] + t[2]) >> 10; tmp[2 * 8 + c] = (x[2] + t[1]) >> 10; tmp[3 * 8 + c] = (x[3] + t[0]) >> 10;
        tmp[4 * 8 + c] = (x[3] - t[0]) >> 10; tmp[5 * 8 + c] = (x[2] - t[1]) >> 10; tmp[6 * 8 + c] = (x[1] - t[2]) >> 10; tmp[7 * 8 + c] = (x[0] - t[3]) >> 10;
    }
    for (int r = 0; r < 8; ++r) { /* pass 2: along each row, descale by 17 */
        int32_t x[4], t[4];
        idct1d(tmp + r * 8, x, t);
        for (int k = 0; k < 4; ++k) x[k] += 65536;
        int32_t *o = block + r * 8;
        o[0] = (x[0] + t[3]) >> 17; o[1] = (x[1] + t[2]) >> 17; o[2] = (x[2] + t[1]) >> 17; o[3] = (x[3] + t[0]) >> 17;
        o[4] = (x[3] - t[0]) >> 17; o[5] = (x[2] - t[1]) >> 17; o[6] = (x[1] - t[2]) >> 17; o[7] = (x[0] - t[3]) >> 17;
    }
}

/* ---- colour (:2518-2749) ------------------------------------------------------------------------------------------------------- */
static uint8_t clamp_u8(int64_t v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
static float lerpf(float a, float b, float t) { return fmaf(b - a, t, a); } /* std.math.lerp = @mulAdd(b - a, t, a) */
static int32_t round_away(float v) { return (int32_t)roundf(v); }
static void ycbcr_u8_to_rgb(int32_t Y, int32_t Cb, int32_t Cr, uint8_t out[3]) { /* Ycbcr(u8){clamped}.to(.rgb): color.zig:1057-1068 */
    const int64_t y = clamp_u8(Y), cb = (int64_t)clamp_u8(Cb + 128) - 128, cr = (int64_t)clamp_u8(Cr + 128) - 128;
    out[0] = clamp_u8((65536 * y + 91881 * cr + 32768) >> 16);
    out[1] = clamp_u8((65536 * y - 22554 * cb - 46802 * cr + 32768) >> 16);
    out[2] = clamp_u8((65536 * y + 116130 * cb + 32768) >> 16);
}
/* chroma tap position along one axis for luma sample index i (0 .. 8 * factor - 1) */
static void chroma_tap(int i, float scale, int *c0, int *c1, float *f) {
    const float pos = ((float)i + 0.5f) * scale - 0.5f;
    float fl = floorf(pos);
    if (fl < 0) fl = 0;
    if (fl > 7) fl = 7;
    *c0 = (int)fl;
    *c1 = *c0 + 1 < 7 ? *c0 + 1 : 7;
    *f = pos - (float)*c0;
}
static void convert_blocks(jstate *s) {
    if (s->header.num_components == 1) {
        for (size_t b = 0; b < s->nblocks; ++b)
            for (int i = 0; i < 64; ++i) s->rgb[b][0][i] = s->rgb[b][1][i] = s->rgb[b][2][i] = clamp_u8(s->blocks[b][0][i]);
        return;
    }
    const int mh = s->comp[0].h, mv = s->comp[0].v;
    if (mh == 1 && mv == 1) { /* 4:4:4 (:2541-2564) */
        for (size_t b = 0; b < s->nblocks; ++b)
            for (int i = 0; i < 64; ++i) {
                const int32_t Y = s->blocks[b][0][i], Cb = s->blocks[b][1][i], Cr = s->blocks[b][2][i];
                s->rgb[b][0][i] = clamp_u8(Y + ((91881 * Cr + 32768) >> 16));
                s->rgb[b][1][i] = clamp_u8(Y - ((22554 * Cb + 46802 * Cr + 32768) >> 16));
                s->rgb[b][2][i] = clamp_u8(Y + ((116130 * Cb + 32768) >> 16));
            }
        return;
    }
    /* 4:2:2 and 4:1:1 step one block row at a time (:2567-2670), 4:2:0 steps MCUs (:2672-2749); in all three the chroma of an
     * MCU is the 8 x 8 block stored at the MCU's first luma block, and the taps are clamped to that block */
    const float sx = mh == 4 ? 0.25f : 0.5f;
    const int vertical = mv == 2;
    for (unsigned my = 0; my < s->block_height; my += mv)
        for (unsigned mx = 0; mx < s->block_width; mx += mh) {
            const size_t cblock = (size_t)my * s->block_width_actual + mx;
            for (int v = 0; v < mv; ++v)
                for (int h = 0; h < mh; ++h) {
                    const unsigned by = my + v, bx = mx + h;
                    if (by >= s->block_height || bx >= s->block_width) continue;
                    const size_t yb = (size_t)by * s->block_width_actual + bx;
                    for (int p = 0; p < 64; ++p) {
                        const int py = p / 8, px = p % 8;
                        int cx0, cx1, cy0 = py, cy1 = py;
                        float fx, fy = 0;
                        chroma_tap(h * 8 + px, sx, &cx0, &cx1, &fx);
                        int32_t Cb, Cr;
                        const int32_t *cbp = s->blocks[cblock][1], *crp = s->blocks[cblock][2];
                        if (vertical) {
                            chroma_tap(v * 8 + py, 0.5f, &cy0, &cy1, &fy);
                            Cb = round_away(lerpf(lerpf((float)cbp[cy0 * 8 + cx0], (float)cbp[cy0 * 8 + cx1], fx), lerpf((float)cbp[cy1 * 8 + cx0], (float)cbp[cy1 * 8 + cx1], fx), fy));
                            Cr = round_away(lerpf(lerpf((float)crp[cy0 * 8 + cx0], (float)crp[cy0 * 8 + cx1], fx), lerpf((float)crp[cy1 * 8 + cx0], (float)crp[cy1 * 8 + cx1], fx), fy));
                        } else {
                            Cb = round_away(lerpf((float)cbp[py * 8 + cx0], (float)cbp[py * 8 + cx1], fx));
                            Cr = round_away(lerpf((float)crp[py * 8 + cx0], (float)crp[py * 8 + cx1], fx));
                        }
                        uint8_t rgb[3];
                        ycbcr_u8_to_rgb(s->blocks[yb][0][p], Cb, Cr, rgb);
                        s->rgb[yb][0][p] = rgb[0]; s->rgb[yb][1][p] = rgb[1]; s->rgb[yb][2][p] = rgb[2];
                    }
                }
        }
}

/* decode + toNativeImage (:2786-2821). *pixels_out: rows * cols of ZO_U8 (one component) or ZO_RGB_U8. */
ZO_API int zo_jpeg_decode_native(const uint8_t *data, size_t len, const zo_jpeg_limits *lim_in, zo_jpeg_header *header_out, int *native_out,
                                 uint8_t **pixels_out, int *scan_limit_reached_out) {
    zo_jpeg_limits lim;
    if (lim_in) lim = *lim_in; else zo_jpeg_default_limits(&lim);
    jstate *s = calloc(1, sizeof *s);
    if (!s) return J_OutOfMemory;
    s->header.precision = 8;
    uint8_t *out = NULL;
    int rc = decode_stream(s, data, len, &lim);
    if (rc) goto done;
    if (header_out) *header_out = s->header;
    if (scan_limit_reached_out) *scan_limit_reached_out = s->scan_limit_reached;
    if (!pixels_out) goto done; /* decode() only */
    if (!s->header.progressive && (rc = baseline_scan(s))) goto done;
    if (!s->blocks) { rc = J_BlockStorageNotAllocated; goto done; }
    for (size_t b = 0; b < s->nblocks; ++b) /* dequantizeAllBlocks (:2482-2495) */
        for (int c = 0; c < s->header.num_components; ++c) {
            const int tq = s->comp[c].tq;
            if (tq > 3 || !s->have_q[tq]) { rc = J_MissingQuantTable; goto done; }
            for (int i = 0; i < 64; ++i) s->blocks[b][c][i] *= (int32_t)s->q[tq][i];
        }
    for (size_t b = 0; b < s->nblocks; ++b) /* idctAllBlocks (:2498-2515): level shift on component 0 only */
        for (int c = 0; c < s->header.num_components; ++c) {
            zo_jpeg_idct8x8(s->blocks[b][c]);
            if (c == 0) for (int i = 0; i < 64; ++i) s->blocks[b][c][i] += 128;
        }
    convert_blocks(s);
    const int gray = s->header.num_components == 1, ch = gray ? 1 : 3;
    const uint32_t W = s->header.width, H = s->header.height;
    out = calloc((size_t)W * H * ch + 1, 1);
    if (!out) { rc = J_OutOfMemory; goto done; }
    for (unsigned by = 0; by < s->block_height; ++by) /* renderRgbBlocksToPixels (:2752-2784); convertColor(u8, rgb) of r = g = b is r */
        for (unsigned bx = 0; bx < s->block_width; ++bx) {
            const size_t b = (size_t)by * s->block_width_actual + bx;
            for (int y = 0; y < 8; ++y)
                for (int x = 0; x < 8; ++x) {
                    const size_t py = (size_t)by * 8 + y, px = (size_t)bx * 8 + x;
                    if (py >= H || px >= W) continue;
                    uint8_t *o = out + (py * W + px) * ch;
                    if (gray) o[0] = s->rgb[b][0][y * 8 + x];
                    else { o[0] = s->rgb[b][0][y * 8 + x]; o[1] = s->rgb[b][1][y * 8 + x]; o[2] = s->rgb[b][2][y * 8 + x]; }
                }
        }
    *native_out = gray ? ZO_U8 : ZO_RGB_U8;
    *pixels_out = out;
    out = NULL;
done:
    free(s->blocks);
    free(s->rgb);
    free(s);
    free(out);
    return rc;
}
ZO_API void zo_jpeg_free(void *p) { free(p); }

/* BitReader.getBits (:1660-1736) on its own: reads counts[i] bits n times; returns how many reads succeeded */
ZO_API int zo_jpeg_get_bits(const uint8_t *data, size_t len, const int *counts, int n, uint32_t *out) {
    bit_reader b = {data, len, 0, 0, 0};
    int i = 0;
    for (; i < n; ++i)
        if (br_get(&b, counts[i], &out[i])) break;
    return i;
}

/* decode (+ performBlockScan for a baseline file): FNV-1a (one 32-bit coefficient per step) over the coefficient blocks, component by component, block by block
 * — the state toNativeImage starts from. Lets the entropy decoders be compared without rendering anything. */
ZO_API int zo_jpeg_coefficient_hash(const uint8_t *data, size_t len, const zo_jpeg_limits *lim_in, uint64_t *hash_out) {
    zo_jpeg_limits lim;
    if (lim_in) lim = *lim_in; else zo_jpeg_default_limits(&lim);
    jstate *s = calloc(1, sizeof *s);
    if (!s) return J_OutOfMemory;
    s->header.precision = 8;
    int rc = decode_stream(s, data, len, &lim);
    if (!rc && !s->header.progressive) rc = baseline_scan(s);
    if (!rc && !s->blocks) rc = J_BlockStorageNotAllocated;
    if (!rc) {
        uint64_t h = 1469598103934665603ull;
        for (int c = 0; c < s->header.num_components; ++c)
            for (size_t b = 0; b < s->nblocks; ++b)
                for (int i = 0; i < 64; ++i) {
                    h ^= (uint32_t)s->blocks[b][c][i];
                    h *= 1099511628211ull;
                }
        *hash_out = h;
    }
    free(s->blocks);
    free(s->rgb);
    free(s);
    return rc;
}

/* ==== encoder (jpeg.zig:293-1043): baseline SOF0, the reference's own tables, LLM forward DCT, reciprocal quantisation ============
 * The output is a deterministic function of the pixels and options (fixed Huffman tables), so files are compared byte for byte. */
static const uint8_t Q_LUMA[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                                   18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t Q_CHROMA[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
/* jpeg.zig:358-391. The luma DC table carries the chroma table's code lengths (0 3 1 1 ...), not Annex K's (0 1 5 1 ...). */
static const uint8_t BITS_DC[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t VAL_DC[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t BITS_AC_LUMA[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125};
static const uint8_t VAL_AC_LUMA[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1,
    0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a,
    0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
    0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t BITS_AC_CHROMA[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119};
static const uint8_t VAL_AC_CHROMA[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1,
    0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
    0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

typedef struct huff_enc { uint16_t code[256]; uint8_t size[256]; } huff_enc;
static void build_enc(huff_enc *e, const uint8_t *bits, const uint8_t *vals) { /* buildHuffmanEncoder (:399-415) */
    memset(e, 0, sizeof *e);
    unsigned code = 0, k = 0;
    for (int i = 0; i < 16; ++i) {
        for (int j = 0; j < bits[i]; ++j) { e->code[vals[k]] = (uint16_t)code; e->size[vals[k]] = (uint8_t)(i + 1); code += 1; k += 1; }
        code = (code << 1) & 0xffff;
    }
}
typedef struct out_buf { uint8_t *p; size_t n, cap; int oom; uint32_t bit_buf; int bit_count; } out_buf;
static void ob_byte(out_buf *o, uint8_t b) {
    if (o->n == o->cap) {
        const size_t nc = o->cap ? o->cap * 2 : 4096;
        uint8_t *np = realloc(o->p, nc);
        if (!np) { o->oom = 1; return; }
        o->p = np; o->cap = nc;
    }
    o->p[o->n++] = b;
}
static void ob_bytes(out_buf *o, const void *src, size_t n) { for (size_t i = 0; i < n; ++i) ob_byte(o, ((const uint8_t *)src)[i]); }
static void ob_segment(out_buf *o, int marker, const uint8_t *payload, size_t n) { /* writeSegment (:457-462) */
    ob_byte(o, 0xFF); ob_byte(o, (uint8_t)marker);
    ob_byte(o, (uint8_t)((n + 2) >> 8)); ob_byte(o, (uint8_t)(n + 2));
    ob_bytes(o, payload, n);
}
static void ew_bits(out_buf *o, uint32_t code, int size) { /* EntropyWriter.writeBits (:431-440) */
    if (size == 0) { return; }
    o->bit_buf = (o->bit_buf << size) | (code & ((1u << size) - 1));
    o->bit_count += size;
    while (o->bit_count >= 8) {
        const uint8_t b = (uint8_t)((o->bit_buf >> (o->bit_count - 8)) & 0xFF);
        ob_byte(o, b);
        if (b == 0xFF) ob_byte(o, 0x00);
        o->bit_count -= 8;
    }
}
static void scale_quant(int quality, uint8_t ql[64], uint8_t qc[64]) { /* scaleQuantTables (:464-476) */
    const int q = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
    const int scale = q < 50 ? 5000 / q : 200 - q * 2;
    for (int i = 0; i < 64; ++i) {
        const int l = (Q_LUMA[i] * scale + 50) / 100, c = (Q_CHROMA[i] * scale + 50) / 100;
        ql[i] = (uint8_t)(l < 1 ? 1 : (l > 255 ? 255 : l));
        qc[i] = (uint8_t)(c < 1 ? 1 : (c > 255 ? 255 : c));
    }
}
static int32_t descale(int64_t x, int n) { return (int32_t)((x + ((int64_t)1 << (n - 1))) >> n); }
#define FIX13(x) ((int64_t)((x) * 8192.0 + 0.5))
static void fdct_1d(const int64_t in[8], int32_t out[8], int first_pass) { /* one pass of fdct8x8_llm (:634-741) */
    const int64_t tmp0 = in[0] + in[7], tmp7 = in[0] - in[7], tmp1 = in[1] + in[6], tmp6 = in[1] - in[6];
    const int64_t tmp2 = in[2] + in[5], tmp5 = in[2] - in[5], tmp3 = in[3] + in[4], tmp4 = in[3] - in[4];
    const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    const int shift = first_pass ? 13 - 2 : 13 + 2;
    if (first_pass) { out[0] = (int32_t)((tmp10 + tmp11) << 2); out[4] = (int32_t)((tmp10 - tmp11) << 2); }
    else { out[0] = descale(tmp10 + tmp11, 2); out[4] = descale(tmp10 - tmp11, 2); }
    const int64_t z1 = (tmp12 + tmp13) * FIX13(0.541196100);
    out[2] = descale(z1 + tmp13 * FIX13(0.765366865), shift);
    out[6] = descale(z1 + tmp12 * (-FIX13(1.847759065)), shift);
    int64_t z1o = tmp4 + tmp7, z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
    const int64_t z5 = (z3 + z4) * FIX13(1.175875602);
    const int64_t t4 = tmp4 * FIX13(0.298631336), t5 = tmp5 * FIX13(2.053119869), t6 = tmp6 * FIX13(3.072711026), t7 = tmp7 * FIX13(1.501321110);
    z1o *= -FIX13(0.899976223); z2 *= -FIX13(2.562915447); z3 *= -FIX13(1.961570560); z4 *= -FIX13(0.390180644);
    z3 += z5; z4 += z5;
    out[7] = descale(t4 + z1o + z3, shift); out[5] = descale(t5 + z2 + z4, shift); out[3] = descale(t6 + z2 + z3, shift); out[1] = descale(t7 + z1o + z4, shift);
}
ZO_API void zo_jpeg_fdct8x8(const int32_t src[64], int32_t dst[64]) {
    int32_t data[64];
    for (int y = 0; y < 8; ++y) {
        int64_t in[8];
        for (int x = 0; x < 8; ++x) in[x] = src[y * 8 + x];
        fdct_1d(in, data + y * 8, 1);
    }
    for (int x = 0; x < 8; ++x) {
        int64_t in[8];
        int32_t out[8];
        for (int y = 0; y < 8; ++y) in[y] = data[y * 8 + x];
        fdct_1d(in, out, 0);
        for (int y = 0; y < 8; ++y) dst[y * 8 + x] = out[y];
    }
}
static void build_recip(uint32_t out[64], const uint8_t q[64]) { /* buildQuantRecipLLM (:749-761) */
    for (int i = 0; i < 64; ++i) {
        double r = 16777216.0 / ((double)q[i] * 8.0);
        if (r < 0.0) r = 0.0;
        if (r > 4294967295.0) r = 4294967295.0;
        out[i] = (uint32_t)round(r);
    }
}
static int32_t quantize_recip(int32_t v, uint32_t recip) { /* quantizeWithRecip (:763-770) */
    if (v == 0) return 0;
    const int64_t a = v < 0 ? -(int64_t)v : v;
    int64_t q = (a * (int64_t)recip + ((int64_t)1 << 23)) >> 24;
    if (v < 0) q = -q;
    return (int32_t)q;
}
static int magnitude_category(int32_t v) { int c = 0; uint32_t a = (uint32_t)(v < 0 ? -v : v); while (a) { a >>= 1; ++c; } return c; }
static uint32_t magnitude_bits(int32_t v, int mag) { return mag == 0 ? 0 : (v >= 0 ? (uint32_t)v : (uint32_t)(((int32_t)1 << mag) - 1 + v)); }
static void encode_block(const int32_t block[64], const uint32_t recip[64], out_buf *o, const huff_enc *dc, const huff_enc *ac, int32_t *prev_dc) { /* :771-817 */
    int32_t dct[64], co[64];
    zo_jpeg_fdct8x8(block, dct);
    for (int i = 0; i < 64; ++i) co[i] = quantize_recip(dct[i], recip[i]);
    const int32_t diff = co[0] - *prev_dc;
    *prev_dc = co[0];
    const int mag = magnitude_category(diff);
    ew_bits(o, dc->code[mag], dc->size[mag]);
    if (mag > 0) ew_bits(o, magnitude_bits(diff, mag), mag);
    int run = 0;
    for (int k = 1; k < 64; ++k) {
        const int32_t v = co[ZIGZAG[k]];
        if (v == 0) {
            if (++run == 16) { ew_bits(o, ac->code[0xF0], ac->size[0xF0]); run = 0; } /* a ZRL as soon as sixteen zeros pile up, trailing ones included */
        } else {
            const int amag = magnitude_category(v), sym = (run << 4) | amag;
            ew_bits(o, ac->code[sym & 0xff], ac->size[sym & 0xff]);
            ew_bits(o, magnitude_bits(v, amag), amag);
            run = 0;
        }
    }
    if (run > 0) ew_bits(o, ac->code[0x00], ac->size[0x00]);
}
static void rgb_to_ycc(const uint8_t *p, uint8_t out[3]) { /* convertColor(Ycbcr, Rgb(u8)): color.zig:987-1009 */
    const int64_t r = p[0], g = p[1], b = p[2];
    out[0] = clamp_u8((19595 * r + 38470 * g + 7471 * b + 32768) >> 16);
    out[1] = clamp_u8(((-11059 * r - 21710 * g + 32768 * b + 32768) >> 16) + 128);
    out[2] = clamp_u8(((32768 * r - 27439 * g - 5329 * b + 32768) >> 16) + 128);
}
/* jpeg.encode (:307-329) for Image(u8) and Image(Rgb); subsampling 0 = 4:4:4, 1 = 4:2:2, 2 = 4:2:0; comment may be NULL.
 * Status: 0 ok, 1 = error.InvalidImageDimensions, 2 = error.ImageTooLarge, 3 = out of memory / unsupported pixel type. */
ZO_API int zo_jpeg_encode(const zo_image *img, int quality, int subsampling, int density_dpi, const uint8_t *comment, size_t comment_len, uint8_t **out, size_t *out_len) {
    if (img->rows == 0 || img->cols == 0) return 1;
    if (img->rows > 65535 || img->cols > 65535) return 2;
    if (img->pixel != ZO_U8 && img->pixel != ZO_RGB_U8) return 3;
    const int gray = img->pixel == ZO_U8;
    const size_t rows = img->rows, cols = img->cols;
    out_buf o;
    memset(&o, 0, sizeof o);
    ob_byte(&o, 0xFF); ob_byte(&o, 0xD8);
    const uint8_t jfif[14] = {'J', 'F', 'I', 'F', 0, 1, 1, 1, (uint8_t)(density_dpi >> 8), (uint8_t)density_dpi, (uint8_t)(density_dpi >> 8), (uint8_t)density_dpi, 0, 0};
    ob_segment(&o, 0xE0, jfif, sizeof jfif);
    if (comment) ob_segment(&o, 0xFE, comment, comment_len);
    uint8_t ql[64], qc[64], seg[2 * 65 + 4 * 200];
    scale_quant(quality, ql, qc);
    size_t n = 0;
    seg[n++] = 0x00;
    for (int i = 0; i < 64; ++i) seg[n++] = ql[ZIGZAG[i]];
    if (!gray) { seg[n++] = 0x01; for (int i = 0; i < 64; ++i) seg[n++] = qc[ZIGZAG[i]]; }
    ob_segment(&o, 0xDB, seg, n);
    const uint8_t luma_factors = subsampling == 0 ? 0x11 : (subsampling == 1 ? 0x21 : 0x22);
    n = 0;
    seg[n++] = 8; seg[n++] = (uint8_t)(rows >> 8); seg[n++] = (uint8_t)rows; seg[n++] = (uint8_t)(cols >> 8); seg[n++] = (uint8_t)cols;
    if (gray) { seg[n++] = 1; seg[n++] = 1; seg[n++] = 0x11; seg[n++] = 0; }
    else { const uint8_t c[10] = {3, 1, luma_factors, 0, 2, 0x11, 1, 3, 0x11, 1}; memcpy(seg + n, c, 10); n += 10; }
    ob_segment(&o, 0xC0, seg, n);
    n = 0;
    seg[n++] = 0x00; memcpy(seg + n, BITS_DC, 16); n += 16; memcpy(seg + n, VAL_DC, 12); n += 12;
    seg[n++] = 0x10; memcpy(seg + n, BITS_AC_LUMA, 16); n += 16; memcpy(seg + n, VAL_AC_LUMA, 162); n += 162;
    if (!gray) {
        seg[n++] = 0x01; memcpy(seg + n, BITS_DC, 16); n += 16; memcpy(seg + n, VAL_DC, 12); n += 12;
        seg[n++] = 0x11; memcpy(seg + n, BITS_AC_CHROMA, 16); n += 16; memcpy(seg + n, VAL_AC_CHROMA, 162); n += 162;
    }
    ob_segment(&o, 0xC4, seg, n);
    if (gray) { const uint8_t s[6] = {1, 1, 0x00, 0, 63, 0}; ob_segment(&o, 0xDA, s, 6); }
    else { const uint8_t s[10] = {3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0}; ob_segment(&o, 0xDA, s, 10); }

    huff_enc dc_l, ac_l, dc_c, ac_c;
    build_enc(&dc_l, BITS_DC, VAL_DC); build_enc(&ac_l, BITS_AC_LUMA, VAL_AC_LUMA);
    build_enc(&dc_c, BITS_DC, VAL_DC); build_enc(&ac_c, BITS_AC_CHROMA, VAL_AC_CHROMA);
    uint32_t rl[64], rc[64];
    build_recip(rl, ql); build_recip(rc, qc);
    const uint8_t *base = (const uint8_t *)img->data;
    int32_t block[64];
    if (gray) { /* encodeGrayscale (:977-1043): edge pixels replicate into the padding */
        int32_t prev = 0;
        for (size_t br = 0; br < (rows + 7) / 8; ++br)
            for (size_t bc = 0; bc < (cols + 7) / 8; ++bc) {
                for (size_t y = 0; y < 8; ++y) {
                    const size_t iy = br * 8 + y < rows - 1 ? br * 8 + y : rows - 1;
                    for (size_t x = 0; x < 8; ++x) {
                        const size_t ix = bc * 8 + x < cols - 1 ? bc * 8 + x : cols - 1;
                        block[y * 8 + x] = (int32_t)base[iy * img->stride + ix] - 128;
                    }
                }
                encode_block(block, rl, &o, &dc_l, &ac_l, &prev);
            }
    } else { /* encodeBlocksRgb (:819-927) */
        const size_t hm = subsampling == 0 ? 1 : 2, vm = subsampling == 2 ? 2 : 1, mw = 8 * hm, mh = 8 * vm;
        int32_t py = 0, pcb = 0, pcr = 0;
        uint8_t mcu[16][16][3];
        for (size_t my = 0; my < (rows + mh - 1) / mh; ++my)
            for (size_t mx = 0; mx < (cols + mw - 1) / mw; ++mx) {
                for (size_t y = 0; y < mh; ++y) {
                    const size_t iy = my * mh + y < rows - 1 ? my * mh + y : rows - 1;
                    for (size_t x = 0; x < mw; ++x) {
                        const size_t ix = mx * mw + x < cols - 1 ? mx * mw + x : cols - 1;
                        rgb_to_ycc(base + (iy * img->stride + ix) * 3, mcu[y][x]);
                    }
                }
                for (size_t vy = 0; vy < vm; ++vy)
                    for (size_t hx = 0; hx < hm; ++hx) {
                        for (size_t y = 0; y < 8; ++y)
                            for (size_t x = 0; x < 8; ++x) block[y * 8 + x] = (int32_t)mcu[vy * 8 + y][hx * 8 + x][0] - 128;
                        encode_block(block, rl, &o, &dc_l, &ac_l, &py);
                    }
                for (int c = 1; c <= 2; ++c) {
                    for (size_t y = 0; y < 8; ++y)
                        for (size_t x = 0; x < 8; ++x) {
                            uint32_t sum = 0;
                            for (size_t dy = 0; dy < vm; ++dy)
                                for (size_t dx = 0; dx < hm; ++dx) sum += mcu[y * vm + dy][x * hm + dx][c];
                            block[y * 8 + x] = (int32_t)(sum / (uint32_t)(hm * vm)) - 128;
                        }
                    encode_block(block, rc, &o, &dc_c, &ac_c, c == 1 ? &pcb : &pcr);
                }
            }
    }
    if (o.bit_count > 0) { const int pad = 8 - o.bit_count; ew_bits(&o, (1u << pad) - 1, pad); } /* flush (:441-446) */
    ob_byte(&o, 0xFF); ob_byte(&o, 0xD9);
    if (o.oom) { free(o.p); return 3; }
    *out = o.p; *out_len = o.n;
    return 0;
}
