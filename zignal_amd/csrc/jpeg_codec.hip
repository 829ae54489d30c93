// jpeg_codec.hip — JPEG files -> device images (reference src/codecs/jpeg.zig).
//
//   host    markers, tables, limits and errors (jpeg.zig:2035-2151, :1314-1645) and Huffman decoding of every scan into
//           coefficient blocks (:1196-1310, :1740-1952, :2397-2479): a bit-serial chain, one per scan. The reference's bit
//           reader is kept as it is (:1660-1736): restart markers are swallowed by the filler and a restart boundary drops
//           whatever was pre-fetched, so most files with restart intervals come out as the reference produces them — wrong.
//   device  per block: dequantise, integer IDCT (:2204-2394), +128 on component 0 (:2498-2515); per pixel: the chroma taps
//           of the file's layout, YCbCr -> RGB, crop (:2518-2784); then zg_convert when T is not the native type.
// Coefficients travel as i32 (the reference's own storage type: streams that decode to nonsense must produce the same
// nonsense, and i16 would saturate differently); chroma goes up compacted to one block per MCU.
#include "zg_common.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

namespace zg {
namespace {

int jpeg_fail(const char *zig_error, const char *where) {
    set_error("%s (%s)", zig_error, where);
    return ZG_ERR_CODEC;
}
#define JPEG_FAIL(name) return jpeg_fail(name, __func__)
enum { kEndOfData = -1000 }; // error.UnexpectedEndOfData travelling inside the entropy decoder (it ends a scan quietly)

inline bool over(uint64_t limit, uint64_t value) { return limit != 0 && value > limit; }
inline unsigned be16(const uint8_t *p) { return (unsigned)p[0] << 8 | p[1]; }

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ---- entropy layer -------------------------------------------------------------------------------------------------------------
struct Huffman { // HuffmanTable (:1648-1657)
    bool present = false;
    uint8_t fast_symbol[512], fast_length[512];
    int32_t max_code[17];
    uint16_t min_code[17], first_value[17];
    uint8_t values[256];
};

class BitReader { // :1660-1736
  public:
    const uint8_t *data = nullptr;
    size_t len = 0, pos = 0;
    uint64_t window = 0;
    int held = 0;

    bool fill(int want) { // false = error.UnexpectedEndOfData
        while (held <= 56 && held < want) {
            if (pos >= len) return false;
            uint64_t byte = data[pos++];
            if (byte == 0xFF) {
                for (;;) {
                    if (pos >= len) return false;
                    const uint8_t next = data[pos++];
                    if (next == 0x00) break;           // a stuffed 0xFF
                    if (next == 0xFF) continue;        // fill bytes
                    if (next >= 0xD0 && next <= 0xD7) { // RSTn: skipped, the byte after it is data
                        if (pos >= len) return false;
                        byte = data[pos++];
                        if (byte == 0xFF) continue;
                        break;
                    }
                    pos -= 2; // any other marker ends the entropy data
                    return false;
                }
            }
            window |= byte << (56 - held);
            held += 8;
        }
        return true;
    }
    bool peek(int n, uint32_t *out) {
        if (n == 0) { *out = 0; return true; }
        if (!fill(n)) return false;
        *out = (uint32_t)(window >> (64 - n));
        return true;
    }
    void drop(int n) {
        if (n == 0) return;
        window <<= n;
        held -= n;
    }
    bool take(int n, uint32_t *out) {
        if (!peek(n, out)) return false;
        drop(n);
        return true;
    }
    void flush() { window = 0; held = 0; }
};

// Coefficient storage: calloc'd so that pages nothing writes to (most of a subsampled chroma plane) are never materialised.
// The last decode's three arrays stay with the thread that made them: a frame-sized calloc is an mmap, a page fault per 4 KiB
// written and an munmap (a TLB shoot-down across every thread of the process) per call, where clearing a kept array is
// one memset. Arrays above 256 MiB are not kept, and one that is more than twice what is asked for is not reused.
struct CoefficientCache {
    struct Slot { int32_t *p = nullptr; size_t words = 0; } slot[3];
    ~CoefficientCache() { for (Slot &s : slot) free(s.p); }
    int32_t *take(size_t n, size_t *words) {
        Slot *best = nullptr;
        for (Slot &s : slot)
            if (s.p && s.words >= n && s.words / 2 <= n && (!best || s.words < best->words)) best = &s;
        if (!best) return nullptr;
        int32_t *p = best->p;
        *words = best->words;
        *best = Slot{};
        return p;
    }
    void give(int32_t *p, size_t words) {
        Slot *into = nullptr;
        if (words * sizeof(int32_t) <= ((size_t)256 << 20))
            for (Slot &s : slot)
                if (!into || s.words < into->words) into = &s; // an empty slot, else the smallest kept array
        if (!into || (into->p && into->words >= words)) { free(p); return; }
        free(into->p);
        *into = Slot{p, words};
    }
};
inline CoefficientCache &coefficient_cache() {
    static thread_local CoefficientCache cache;
    return cache;
}
struct ZeroedWords {
    int32_t *p = nullptr;
    size_t words = 0;
    ~ZeroedWords() { release(); }
    void release() {
        if (p) coefficient_cache().give(p, words);
        p = nullptr;
        words = 0;
    }
    bool reset(size_t n) {
        release();
        if (n == 0) n = 1;
        if ((p = coefficient_cache().take(n, &words))) {
            memset(p, 0, n * sizeof(int32_t));
            return true;
        }
        p = (int32_t *)calloc(n, sizeof(int32_t));
        words = n;
        return p != nullptr;
    }
    int32_t *data() { return p; }
};

struct FrameComponent { uint8_t id, h, v, tq; };
struct ScanComponent { uint8_t id, dc, ac; };
struct Scan { ScanComponent comp[4]; int n = 0, ss = 0, se = 0, ah = 0, al = 0; };

struct Decoder {
    zg_jpeg_header header{};
    FrameComponent comp[4]{};
    Huffman dc[4], ac[4];
    bool have_q[4] = {false, false, false, false};
    uint16_t q[4][64];
    Scan baseline;
    unsigned restart_interval = 0;
    BitReader bits;
    unsigned bw = 0, bh = 0, bwa = 0, bha = 0; // block_width / height, and the MCU-padded grid
    size_t nblocks = 0;
    bool allocated = false;
    // The reference's block_storage[block][component] (:1157), one array per component. Luma-grid positions a component can
    // never be written at are not stored: the chroma of a subsampled frame only ever lands on MCU origins (its sampling
    // factors are 1 x 1, :1383-1395), so it is kept as the (bha / mv) x (bwa / mh) grid of those — a quarter of the pages
    // for 4:2:0, and already the layout the device wants. A component whose id is 1 keeps the full grid: a scan of that
    // component alone walks every position (the reference's "non-interleaved" rule, :1745, :2402).
    ZeroedWords coef[3];
    bool compact[3] = {false, false, false};
    unsigned fmh = 1, fmv = 1, cgw = 0, cgh = 0; // frame sampling maxima; the compact grid
    int32_t dc_pred[4] = {0, 0, 0, 0};
    bool scan_limit_reached = false;

    int32_t *block(size_t ay, size_t ax, size_t c) {
        return coef[c].data() + (compact[c] ? (ay / fmv) * cgw + ax / fmh : ay * bwa + ax) * 64;
    }
    const int32_t *block_or_zero(size_t ay, size_t ax, size_t c) { // any position of the luma grid, stored or not
        static const int32_t zeros[64] = {0};
        return compact[c] && (ay % fmv != 0 || ax % fmh != 0) ? zeros : block(ay, ax, c);
    }

    // readCode (:1196-1236): 0 ok, kEndOfData, or a ZG status with the error set
    int read_symbol(const Huffman &t, int *symbol) {
        uint32_t index = 0;
        if (!bits.peek(9, &index)) index = 0;
        if (bits.held >= 9) {
            const uint8_t v = t.fast_symbol[index];
            if (v != 255) {
                bits.drop(t.fast_length[index]);
                *symbol = v;
                return 0;
            }
        }
        uint32_t code = 0;
        int length = 0;
        if (bits.held >= 9) {
            bits.drop(9);
            code = index;
            length = 9;
        }
        while (length < 16) {
            uint32_t bit;
            if (!bits.take(1, &bit)) return kEndOfData;
            code = ((code << 1) | bit) & 0xffff;
            ++length;
            if ((int32_t)code <= t.max_code[length]) {
                // Only a table holding the symbol 255 (the fast table's own "empty" mark) gets here below min_code;
                // the reference asserts (:1238) and would index out of range, so this is an error rather than a read.
                if (code < t.min_code[length]) JPEG_FAIL("InvalidHuffmanCode");
                *symbol = t.values[(size_t)t.first_value[length] + code - t.min_code[length]];
                return 0;
            }
        }
        JPEG_FAIL("InvalidHuffmanCode");
    }
    int read_extended(int magnitude, int32_t *out) { // readMagnitudeCoded (:1239-1252)
        if (magnitude == 0) { *out = 0; return 0; }
        uint32_t raw;
        if (!bits.peek(magnitude, &raw)) return kEndOfData;
        bits.drop(magnitude);
        int32_t v = (int32_t)raw;
        if (v < (int32_t)1 << (magnitude - 1)) v -= ((int32_t)1 << magnitude) - 1;
        *out = v;
        return 0;
    }
    // decodeAC (:1255-1310). The reference stores a zero for every skipped position; its only caller has just cleared the
    // block (:1939) and positions are visited once, in order, so skipping them leaves the same 64 words.
    int read_ac_run(const Huffman &t, int32_t *blk) {
        int k = 1, rc, symbol;
        while (k < 64) {
            if ((rc = read_symbol(t, &symbol))) return rc;
            if (symbol == 0) return 0; // end of block
            const int run = symbol >> 4, size = symbol & 15;
            if (size == 0) {
                if (run != 15) JPEG_FAIL("InvalidACCoefficient");
                k += 16;
                continue;
            }
            k += run;
            if (k >= 64) break;
            int32_t value;
            if ((rc = read_extended(size, &value))) return rc;
            blk[kZigzag[k++]] = value;
        }
        return 0;
    }

    // decodeBlockBaseline (:1933-1952). `clear` is the reference's @memset: not needed for a block of the zeroed storage that
    // this, the only block scan of a baseline file, reaches for the first time.
    int decode_baseline_block(const ScanComponent &sc, int32_t *blk, int32_t *pred, bool clear) {
        if (clear) memset(blk, 0, 64 * sizeof(int32_t));
        if (sc.dc > 3 || !dc[sc.dc].present) JPEG_FAIL("MissingHuffmanTable");
        int rc, symbol;
        if ((rc = read_symbol(dc[sc.dc], &symbol))) return rc;
        if (symbol > 11) JPEG_FAIL("InvalidDCCoefficient");
        int32_t diff;
        if ((rc = read_extended(symbol, &diff))) return rc;
        *pred = (int32_t)((uint32_t)*pred + (uint32_t)diff);
        blk[0] = *pred;
        if (sc.ac > 3 || !ac[sc.ac].present) JPEG_FAIL("MissingHuffmanTable");
        return read_ac_run(ac[sc.ac], blk);
    }

    int refine_bit(int32_t *c, int32_t bit) { // one correction bit for an already non-zero coefficient
        uint32_t u;
        if (!bits.take(1, &u)) return kEndOfData;
        if (u) *c = (int32_t)((uint32_t)*c + (uint32_t)(*c > 0 ? bit : -bit));
        return 0;
    }
    int decode_progressive_block(const Scan &s, const ScanComponent &sc, int32_t *blk, int32_t *pred, uint32_t *eob_run) { // :1816-1930
        int rc, symbol;
        uint32_t u;
        if (s.ss == 0) {
            if (sc.dc > 3 || !dc[sc.dc].present) JPEG_FAIL("MissingHuffmanTable");
            if (s.ah == 0) {
                if ((rc = read_symbol(dc[sc.dc], &symbol))) return rc;
                if (symbol > 11) JPEG_FAIL("InvalidDCCoefficient");
                int32_t diff;
                if ((rc = read_extended(symbol, &diff))) return rc;
                *pred = (int32_t)((uint32_t)diff + (uint32_t)*pred);
                blk[0] = (int32_t)((uint32_t)*pred << s.al);
            } else {
                if (!bits.take(1, &u)) return kEndOfData;
                blk[0] = (int32_t)((uint32_t)blk[0] + (u << s.al));
            }
            return 0;
        }
        if (sc.ac > 3 || !ac[sc.ac].present) JPEG_FAIL("MissingHuffmanTable");
        const Huffman &t = ac[sc.ac];
        int k = s.ss;
        if (s.ah == 0) { // first pass over this band
            if (*eob_run == 0) {
                while (k <= s.se && k < 64) {
                    int32_t value = 0;
                    if ((rc = read_symbol(t, &symbol))) return rc;
                    const int run = symbol >> 4, size = symbol & 15;
                    if (size == 0) {
                        if (run < 15) {
                            if (!bits.take(run, &u)) return kEndOfData;
                            *eob_run = (1u << run) + u;
                            break;
                        }
                    } else {
                        if (size > 10) JPEG_FAIL("InvalidACCoefficient");
                        if ((rc = read_extended(size, &value))) return rc;
                    }
                    for (int i = 0; i < run && k < 64; ++i) blk[kZigzag[k++]] = 0;
                    if (k >= 64) break;
                    blk[kZigzag[k++]] = (int32_t)((uint32_t)value << s.al);
                }
            }
            if (*eob_run > 0) {
                *eob_run -= 1;
                while (k <= s.se && k < 64) blk[kZigzag[k++]] = 0;
            }
            return 0;
        }
        const int32_t bit = (int32_t)1 << s.al; // refinement pass
        if (*eob_run == 0) {
            while (k <= s.se && k < 64) {
                int32_t value = 0;
                if ((rc = read_symbol(t, &symbol))) return rc;
                int run = symbol >> 4;
                if ((symbol & 15) == 0) {
                    if (run < 15) {
                        *eob_run = 1u << run;
                        if (!bits.take(run, &u)) return kEndOfData;
                        *eob_run += u;
                        break;
                    }
                } else {
                    if (!bits.take(1, &u)) return kEndOfData;
                    value = u == 1 ? bit : -bit;
                }
                while (k <= s.se && k < 64) {
                    int32_t *c = &blk[kZigzag[k]];
                    if (*c == 0) {
                        if (run > 0) { --run; ++k; }
                        else { *c = value; ++k; break; }
                    } else {
                        if ((rc = refine_bit(c, bit))) return rc;
                        ++k;
                    }
                }
            }
        }
        if (*eob_run > 0) {
            for (; k <= s.se && k < 64; ++k) {
                int32_t *c = &blk[kZigzag[k]];
                if (*c != 0 && (rc = refine_bit(c, bit))) return rc;
            }
            *eob_run -= 1;
        }
        return 0;
    }

    void sampling_maxima(int *mh, int *mv) const {
        *mh = *mv = 1;
        for (int i = 0; i < header.num_components; ++i) {
            if (comp[i].h > *mh) *mh = comp[i].h;
            if (comp[i].v > *mv) *mv = comp[i].v;
        }
    }
    // the frame component a scan component names, and how many of its blocks sit in one step of the scan grid
    bool locate(const ScanComponent &sc, bool single, size_t *index, unsigned *vcount, unsigned *hcount) const {
        for (int i = 0; i < header.num_components; ++i)
            if (comp[i].id == sc.id) {
                *index = (size_t)i;
                *vcount = single ? 1 : comp[i].v;
                *hcount = single ? 1 : comp[i].h;
                return true;
            }
        return false; // the reference leaves the counts undefined here; nothing is decoded for such a component
    }

    int run_progressive_scan(const Scan &s) { // performProgressiveScan (:1740-1813)
        if (!allocated) JPEG_FAIL("BlockStorageNotAllocated");
        uint32_t eob_run = 0;
        const bool single = s.n == 1 && s.comp[0].id == 1; // the reference's notion of a non-interleaved scan
        int mh, mv;
        sampling_maxima(&mh, &mv);
        const unsigned ystep = single ? 1 : (unsigned)mv, xstep = single ? 1 : (unsigned)mh;
        for (unsigned y = 0; y < bh; y += ystep)
            for (unsigned x = 0; x < bw; x += xstep) {
                const size_t mcu = (size_t)y * bwa + x;
                if (restart_interval != 0 && mcu % ((size_t)restart_interval * ystep * xstep) == 0) {
                    bits.flush();
                    memset(dc_pred, 0, sizeof dc_pred);
                    eob_run = 0;
                }
                for (int i = 0; i < s.n; ++i) {
                    size_t ci;
                    unsigned vcount, hcount;
                    if (!locate(s.comp[i], single, &ci, &vcount, &hcount)) continue;
                    for (unsigned v = 0; v < vcount; ++v)
                        for (unsigned h = 0; h < hcount; ++h) {
                            const size_t id = (size_t)(y + v) * bwa + (x + h);
                            if (id >= nblocks) continue;
                            (void)bits.fill(24);
                            const int rc = decode_progressive_block(s, s.comp[i], block(y + v, x + h, ci), &dc_pred[ci], &eob_run);
                            if (rc == kEndOfData) return ZG_OK; // a cut scan keeps what it decoded (:1799-1803)
                            if (rc) return rc;
                        }
                }
            }
        return ZG_OK;
    }
    int run_baseline_scan() { // performBlockScan (:2397-2479)
        if (!allocated) JPEG_FAIL("BlockStorageNotAllocated");
        const Scan &s = baseline;
        int mh, mv;
        sampling_maxima(&mh, &mv);
        const bool single = s.n == 1 && s.comp[0].id == 1;
        const unsigned ystep = single ? 1 : (unsigned)mv, xstep = single ? 1 : (unsigned)mh;
        int32_t pred[4] = {0, 0, 0, 0}, spare[64];
        uint32_t since_restart = 0;
        bool revisits = false; // a scan that names a component twice decodes into the same blocks twice
        for (int i = 0; i < s.n; ++i)
            for (int j = 0; j < i; ++j) revisits = revisits || s.comp[i].id == s.comp[j].id;
        for (unsigned y = 0; y < bh; y += ystep)
            for (unsigned x = 0; x < bw; x += xstep) {
                if (restart_interval != 0 && since_restart == restart_interval) {
                    memset(pred, 0, sizeof pred);
                    since_restart = 0;
                    bits.flush();
                }
                for (int i = 0; i < s.n; ++i) {
                    size_t ci;
                    unsigned vcount, hcount;
                    if (!locate(s.comp[i], single, &ci, &vcount, &hcount)) continue;
                    for (unsigned v = 0; v < vcount; ++v)
                        for (unsigned h = 0; h < hcount; ++h) {
                            const unsigned ax = x + h, ay = y + v;
                            int32_t *blk = (ay < bh && ax < bw) ? block(ay, ax, ci) : spare; // padding blocks are decoded and dropped
                            (void)bits.fill(24);
                            const int rc = decode_baseline_block(s.comp[i], blk, &pred[ci], revisits || blk == spare);
                            if (rc == kEndOfData) return ZG_OK;
                            if (rc) return rc;
                        }
                }
                ++since_restart;
            }
        return ZG_OK;
    }

    // ---- segments ------------------------------------------------------------------------------------------------------------
    int parse_frame(const uint8_t *d, size_t n, bool progressive, const zg_jpeg_limits &lim) { // parseSOF (:1314-1442)
        if (allocated) JPEG_FAIL("DuplicateSOF");
        header.progressive = progressive ? 1 : 0;
        if (n < 6) JPEG_FAIL("InvalidSOF");
        header.precision = d[0];
        if (d[0] == 12) JPEG_FAIL("Unsupported12BitPrecision");
        if (d[0] == 16) JPEG_FAIL("Unsupported16BitPrecision");
        if (d[0] != 8) JPEG_FAIL("UnsupportedPrecision");
        header.height = be16(d + 1);
        header.width = be16(d + 3);
        header.num_components = d[5];
        if (header.width == 0 || header.height == 0) JPEG_FAIL("InvalidSOF");
        if (over(lim.max_width, header.width) || over(lim.max_height, header.height)) JPEG_FAIL("ImageTooLarge");
        const int nc = d[5];
        if (nc == 4) JPEG_FAIL("UnsupportedComponentCount");
        if (nc != 1 && nc != 3) JPEG_FAIL("InvalidComponentCount");
        int mh = 0, mv = 0;
        for (int i = 0; i < nc; ++i) {
            const size_t at = 6 + (size_t)i * 3;
            if (at + 3 > n) JPEG_FAIL("InvalidSOF");
            comp[i] = FrameComponent{d[at], (uint8_t)(d[at + 1] >> 4), (uint8_t)(d[at + 1] & 15), d[at + 2]};
            if (comp[i].h > mh) mh = comp[i].h;
            if (comp[i].v > mv) mv = comp[i].v;
        }
        if (mh > 4 || mv > 4) JPEG_FAIL("UnsupportedSamplingFactor");
        // a one-component frame with a zero sampling nibble: the reference divides by the MCU size next (:1401-1404, a panic);
        // here it is the error the three-component layouts get
        if (mh == 0 || mv == 0) JPEG_FAIL("UnsupportedSamplingFactor");
        if (nc == 3) {
            if (comp[1].h != comp[2].h || comp[1].v != comp[2].v) JPEG_FAIL("InvalidComponentCount");
            const bool chroma_unit = comp[1].h == 1 && comp[1].v == 1;
            const int lh = comp[0].h, lv = comp[0].v;
            const bool known = (lh == 1 && lv == 1) || (lh == 2 && lv == 2) || (lh == 2 && lv == 1) || (lh == 4 && lv == 1);
            if (!chroma_unit || !known) JPEG_FAIL("UnsupportedSamplingFactor");
        }
        const uint32_t mcu_w = 8u * (uint32_t)mh, mcu_h = 8u * (uint32_t)mv;
        const uint32_t wa = (header.width + mcu_w - 1) / mcu_w * mcu_w, ha = (header.height + mcu_h - 1) / mcu_h * mcu_h;
        bw = (header.width + 7) / 8;
        bh = (header.height + 7) / 8;
        bwa = wa / 8;
        bha = ha / 8;
        const uint64_t padded = (uint64_t)wa * ha;
        if (over(lim.max_pixels, padded)) JPEG_FAIL("ImageTooLarge");
        if (over(lim.max_blocks, padded / 64)) JPEG_FAIL("BlockMemoryLimitExceeded");
        nblocks = (size_t)(padded / 64);
        fmh = (unsigned)mh;
        fmv = (unsigned)mv;
        cgw = bwa / fmh;
        cgh = bha / fmv;
        for (int i = 0; i < nc; ++i) {
            compact[i] = nc == 3 && i > 0 && comp[i].id != 1 && (mh > 1 || mv > 1);
            if (!coef[i].reset(compact[i] ? (size_t)cgw * cgh * 64 : nblocks * 64)) { set_error("jpeg: out of host memory for %zu blocks", nblocks); return ZG_ERR_OUT_OF_MEMORY; }
        }
        allocated = true;
        return ZG_OK;
    }
    int parse_huffman(const uint8_t *d, size_t n) { // parseDHT (:1445-1540)
        if (n == 0) JPEG_FAIL("InvalidDHT");
        size_t at = 0;
        while (at < n) {
            if (at + 17 > n) JPEG_FAIL("InvalidDHT");
            const int cls = (d[at] >> 4) & 1, id = d[at] & 3;
            const uint8_t *counts = d + at + 1;
            at += 17;
            unsigned total = 0;
            for (int i = 0; i < 16; ++i) total += counts[i];
            if (total > 256) JPEG_FAIL("InvalidHuffmanTable");
            if (at + total > n) JPEG_FAIL("InvalidDHT");
            Huffman t;
            memset(t.values, 0, sizeof t.values);
            memcpy(t.values, d + at, total);
            at += total;
            memset(t.fast_symbol, 255, sizeof t.fast_symbol);
            memset(t.fast_length, 0, sizeof t.fast_length);
            for (int i = 0; i < 17; ++i) { t.max_code[i] = -1; t.min_code[i] = 0; t.first_value[i] = 0; }
            unsigned code = 0, next = 0;
            for (int len = 1; len <= 16; ++len) {
                const int count = counts[len - 1];
                if (count > 0) { t.first_value[len] = (uint16_t)next; t.min_code[len] = (uint16_t)code; }
                for (int j = 0; j < count; ++j) {
                    if (code == (1u << len) - 1) JPEG_FAIL("InvalidHuffmanTable"); // the all-ones code is reserved
                    const uint8_t symbol = t.values[next++];
                    if (len <= 9) {
                        const unsigned first = (code << (9 - len)) & 0xffff, span = 1u << (9 - len);
                        for (unsigned k = 0; k < span; ++k) { t.fast_symbol[first + k] = symbol; t.fast_length[first + k] = (uint8_t)len; }
                    }
                    code = (code + 1) & 0xffff;
                }
                if (count > 0) t.max_code[len] = (int32_t)code - 1;
                code = (code << 1) & 0xffff;
            }
            t.present = true;
            (cls == 0 ? dc : ac)[id] = t;
        }
        return ZG_OK;
    }
    int parse_quant(const uint8_t *d, size_t n) { // parseDQT (:1543-1583)
        if (n == 0) JPEG_FAIL("InvalidDQT");
        size_t at = 0;
        while (at < n) {
            const int wide = (d[at] >> 4) & 15, id = d[at] & 3;
            ++at;
            const size_t width = wide == 0 ? 1 : 2;
            if (at + 64 * width > n) JPEG_FAIL("InvalidDQT");
            for (int i = 0; i < 64; ++i) q[id][kZigzag[i]] = width == 1 ? d[at + i] : (uint16_t)be16(d + at + 2 * i);
            at += 64 * width;
            have_q[id] = true;
        }
        return ZG_OK;
    }
    int parse_scan_header(const uint8_t *d, size_t n, Scan *s) { // parseSOS (:1586-1638)
        if (n < 6) JPEG_FAIL("InvalidSOS");
        const int nc = d[0];
        if (!header.progressive && nc != header.num_components) JPEG_FAIL("InvalidSOS");
        if (header.progressive && (nc == 0 || nc > header.num_components)) JPEG_FAIL("InvalidSOS");
        size_t at = 1;
        s->n = nc;
        for (int i = 0; i < nc; ++i) {
            if (at + 2 > n) JPEG_FAIL("InvalidSOS");
            if (i < 4) s->comp[i] = ScanComponent{d[at], (uint8_t)(d[at + 1] >> 4), (uint8_t)(d[at + 1] & 15)};
            at += 2;
        }
        if (at + 3 > n) JPEG_FAIL("InvalidSOS");
        const int ss = d[at], se = d[at + 1], approx = d[at + 2];
        if (!header.progressive) {
            if (ss != 0 || se != 63 || approx != 0) JPEG_FAIL("InvalidSOS");
        } else {
            if (ss > 63 || se > 63 || se < ss) JPEG_FAIL("InvalidSOS");
            if ((ss == 0 || se == 0) && !(ss == 0 && se == 0)) JPEG_FAIL("InvalidSOS"); // a DC scan is 0..0, an AC scan starts above 0
        }
        s->ss = ss; s->se = se; s->ah = approx >> 4; s->al = approx & 15;
        return ZG_OK;
    }

    // decode (:2035-2151)
    int read_stream(const uint8_t *d, size_t len, const zg_jpeg_limits &lim) {
        if (len < 2 || d[0] != 0xFF || d[1] != 0xD8) JPEG_FAIL("InvalidJpegFile");
        if (over(lim.max_jpeg_bytes, len)) JPEG_FAIL("JpegDataTooLarge");
        size_t at = 2, marker_bytes = 0, scans = 0;
        auto charge = [&](size_t n) { marker_bytes += n; return lim.max_marker_bytes != 0 && marker_bytes > lim.max_marker_bytes; };
        while (at + 1 < len) {
            if (d[at] != 0xFF) JPEG_FAIL("InvalidMarker");
            const int m = d[at + 1];
            const bool in_enum = (m >= 0xC0 && m <= 0xC4) || m == 0xCC || (m >= 0xD0 && m <= 0xDF) || (m >= 0xE0 && m <= 0xEF) || m == 0xFE; // Marker (:1045-1098)
            if (!in_enum) { // skipped by its length
                at += 2;
                if (at + 2 > len) break;
                const unsigned length = be16(d + at);
                if (length < 2) JPEG_FAIL("InvalidMarker");
                at += length;
                continue;
            }
            switch (m) {
            case 0xD8: at += 2; continue;
            case 0xD9: at = len; continue; // EOI ends the loop
            case 0xC1: JPEG_FAIL("UnsupportedExtendedSequential");
            case 0xC3: JPEG_FAIL("UnsupportedLosslessJpeg");
            case 0xCC: JPEG_FAIL("UnsupportedArithmeticCoding");
            case 0xDE: JPEG_FAIL("UnsupportedHierarchicalJpeg");
            case 0xDC: JPEG_FAIL("UnsupportedJpegVariant");
            case 0xC0: case 0xC2: case 0xC4: case 0xDB: case 0xDD: { // readMarkerPayload (:2020-2033)
                if (at + 4 > len) JPEG_FAIL("UnexpectedEndOfData");
                const unsigned length = be16(d + at + 2);
                if (length < 2) JPEG_FAIL("InvalidMarker");
                const size_t end = at + 2 + length;
                if (end > len) JPEG_FAIL("InvalidMarker");
                if (charge(length)) JPEG_FAIL("MarkerDataLimitExceeded");
                const uint8_t *payload = d + at + 4;
                const size_t n = end - (at + 4);
                at = end;
                int rc = ZG_OK;
                if (m == 0xC0 || m == 0xC2) rc = parse_frame(payload, n, m == 0xC2, lim);
                else if (m == 0xC4) rc = parse_huffman(payload, n);
                else if (m == 0xDB) rc = parse_quant(payload, n);
                else {
                    if (n != 2) JPEG_FAIL("InvalidDRI");
                    restart_interval = be16(payload);
                }
                if (rc) return rc;
                continue;
            }
            case 0xDA: { // processScanMarker (:1987-2018)
                if (over(lim.max_scans, scans + 1)) { scan_limit_reached = true; at = len; continue; }
                ++scans;
                if (at + 4 > len) JPEG_FAIL("UnexpectedEndOfData");
                const unsigned hl = be16(d + at + 2);
                if (hl < 2) JPEG_FAIL("InvalidMarker");
                const size_t end = at + 2 + hl;
                if (end > len) JPEG_FAIL("InvalidMarker");
                Scan s;
                int rc = parse_scan_header(d + at + 4, end - (at + 4), &s);
                if (rc) return rc;
                size_t stop = end; // findScanEnd (:1955-1978): up to the next real marker; the very last byte of the buffer is never taken
                while (stop + 1 < len) {
                    if (d[stop] == 0xFF) {
                        const uint8_t nb = d[stop + 1];
                        if (nb == 0x00 || (nb >= 0xD0 && nb <= 0xD7)) { stop += 2; continue; }
                        break;
                    }
                    ++stop;
                }
                bits = BitReader{};
                bits.data = d + end;
                bits.len = stop - end;
                if (!header.progressive) {
                    baseline = s;
                    if (charge(stop - at)) JPEG_FAIL("MarkerDataLimitExceeded");
                    return ZG_OK; // a baseline stream is one scan; its blocks are decoded by run_baseline_scan
                }
                if ((rc = run_progressive_scan(s))) return rc;
                if (charge(stop - at)) JPEG_FAIL("MarkerDataLimitExceeded");
                at = stop;
                continue;
            }
            default: { // APPn, COM, and the enum's leftovers (RSTn, EXP): skipped by their length, unchecked (:2119-2139)
                if (at + 4 > len) { at = len; continue; }
                const unsigned length = be16(d + at + 2);
                if (charge(length)) JPEG_FAIL("MarkerDataLimitExceeded");
                at += 2 + (size_t)length;
                continue;
            }
            }
        }
        if (header.progressive) return ZG_OK;
        JPEG_FAIL("NoScanData");
    }
};

// ---- device ----------------------------------------------------------------------------------------------------------------------
// i32 arithmetic that wraps like the hardware does (a corrupt stream can drive the reference's i32 values anywhere)
__device__ inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
__device__ inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }

// idct1D (:2209-2247): stb_image's butterfly; the constants are @round(x * 4096)
__device__ inline void idct_1d(const int32_t s[8], int32_t x[4], int32_t t[4]) {
    int32_t p2 = s[2], p3 = s[6];
    int32_t p1 = wmul(wadd(p2, p3), 2217);
    int32_t t2 = wadd(p1, wmul(p3, -7568)), t3 = wadd(p1, wmul(p2, 3135));
    p2 = s[0]; p3 = s[4];
    int32_t t0 = wmul(wadd(p2, p3), 4096), t1 = wmul(wsub(p2, p3), 4096);
    x[0] = wadd(t0, t3); x[3] = wsub(t0, t3); x[1] = wadd(t1, t2); x[2] = wsub(t1, t2);
    t0 = s[7]; t1 = s[5]; t2 = s[3]; t3 = s[1];
    p3 = wadd(t0, t2);
    int32_t p4 = wadd(t1, t3);
    p1 = wadd(t0, t3); p2 = wadd(t1, t2);
    const int32_t p5 = wmul(wadd(p3, p4), 4816);
    t0 = wmul(t0, 1223); t1 = wmul(t1, 8410); t2 = wmul(t2, 12586); t3 = wmul(t3, 6149);
    p1 = wadd(p5, wmul(p1, -3686)); p2 = wadd(p5, wmul(p2, -10498));
    p3 = wmul(p3, -8035); p4 = wmul(p4, -1598);
    t[3] = wadd(t3, wadd(p1, p4)); t[2] = wadd(t2, wadd(p2, p3)); t[1] = wadd(t1, wadd(p2, p4)); t[0] = wadd(t0, wadd(p1, p3));
}

struct QuantTable { uint16_t q[64]; };

// Dequantise + IDCT + level shift for one component. 32 blocks per workgroup, 8 lanes per block: lane c of a block runs the
// column pass for column c (results to LDS), then the row pass for row c. `blocks` is blocks_x * blocks_y x 64 coefficients in
// natural order; `plane` is the component's sample plane (blocks_y * 8 rows of blocks_x * 8 i32 samples).
__global__ __launch_bounds__(256) void k_jpeg_idct(const int32_t *blocks, QuantTable qt, unsigned nblocks, unsigned blocks_x, int32_t shift, int32_t *plane) {
    __shared__ int32_t tile[32][72]; // row stride 8, block stride 72: the eight lanes of eight blocks hit 64 different banks
    const unsigned local = threadIdx.x >> 3, lane = threadIdx.x & 7, b = blockIdx.x * 32 + local;
    const bool live = b < nblocks;
    int32_t s[8], x[4], t[4];
    int32_t ac = 0;
    if (live) {
        const int32_t *src = blocks + (size_t)b * 64;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            s[r] = wmul(src[r * 8 + lane], (int32_t)qt.q[r * 8 + lane]); // dequantizeAllBlocks (:2482-2495)
            if (r > 0 || lane > 0) ac |= s[r];
        }
    }
    // every coefficient but the DC zero: the reference short-cuts to (dc + 4) >> 3 (:2259-2266)
    ac |= __shfl_xor(ac, 1);
    ac |= __shfl_xor(ac, 2);
    ac |= __shfl_xor(ac, 4);
    const int32_t dc = __shfl(live ? s[0] : 0, (int)(threadIdx.x & 63 & ~7u));
    if (live && ac != 0) {
        idct_1d(s, x, t);
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = wadd(x[k], 512);
        int32_t *col = &tile[local][lane];
        col[0 * 8] = wadd(x[0], t[3]) >> 10; col[1 * 8] = wadd(x[1], t[2]) >> 10; col[2 * 8] = wadd(x[2], t[1]) >> 10; col[3 * 8] = wadd(x[3], t[0]) >> 10;
        col[4 * 8] = wsub(x[3], t[0]) >> 10; col[5 * 8] = wsub(x[2], t[1]) >> 10; col[6 * 8] = wsub(x[1], t[2]) >> 10; col[7 * 8] = wsub(x[0], t[3]) >> 10;
    }
    __syncthreads();
    if (!live) return;
    int32_t o[8];
    if (ac != 0) {
        const int32_t *row = &tile[local][lane * 8];
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] = row[k];
        idct_1d(s, x, t);
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = wadd(x[k], 65536);
        o[0] = wadd(x[0], t[3]) >> 17; o[1] = wadd(x[1], t[2]) >> 17; o[2] = wadd(x[2], t[1]) >> 17; o[3] = wadd(x[3], t[0]) >> 17;
        o[4] = wsub(x[3], t[0]) >> 17; o[5] = wsub(x[2], t[1]) >> 17; o[6] = wsub(x[1], t[2]) >> 17; o[7] = wsub(x[0], t[3]) >> 17;
    } else {
        const int32_t flat = wadd(dc, 4) >> 3;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = flat;
    }
    const unsigned by = b / blocks_x, bx = b % blocks_x;
    int32_t *dst = plane + ((size_t)by * 8 + lane) * ((size_t)blocks_x * 8) + (size_t)bx * 8;
    *(int4 *)dst = make_int4(wadd(o[0], shift), wadd(o[1], shift), wadd(o[2], shift), wadd(o[3], shift));
    *(int4 *)(dst + 4) = make_int4(wadd(o[4], shift), wadd(o[5], shift), wadd(o[6], shift), wadd(o[7], shift));
}

__device__ inline int32_t clamp255(int32_t v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ inline int32_t round_half_away(float v) { // @round, then the cast to i32
    const float r = truncf(v);
    return (int32_t)(fabsf(v - r) >= 0.5f ? r + copysignf(1.0f, v) : r);
}
// The chroma tap along one axis for luma sample i of the MCU (:2592-2595, :2700-2710): the position is clamped into the
// MCU's own 8-sample chroma block, the weight is not (it goes negative at the leading edge).
__device__ inline void chroma_tap(int i, float scale, int *c0, int *c1, float *w) {
    const float at = ((float)i + 0.5f) * scale - 0.5f;
    float f = floorf(at);
    f = f < 0.0f ? 0.0f : (f > 7.0f ? 7.0f : f);
    *c0 = (int)f;
    *c1 = *c0 + 1 < 7 ? *c0 + 1 : 7;
    *w = at - (float)*c0;
}
__device__ inline float lerp_fma(float a, float b, float t) { return __builtin_fmaf(b - a, t, a); } // std.math.lerp

struct RenderArgs {
    const int32_t *y, *cb, *cr; // sample planes
    unsigned luma_pitch, chroma_pitch;
    int components, mh, mv;     // luma sampling factors (chroma is 1 x 1)
};
// One lane per pixel of the cropped image: ycbcrToRgbAllBlocks (:2518-2749) + renderRgbBlocksToPixels (:2752-2784).
template <int NATIVE> __global__ __launch_bounds__(256) void k_jpeg_render(RenderArgs a, DImg dst) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dst.cols) return;
    const int32_t Y = a.y[(size_t)y * a.luma_pitch + x];
    uint8_t *out = (uint8_t *)dst.data + ((size_t)y * dst.stride + x) * Px<NATIVE>::BYTES;
    if constexpr (NATIVE == ZG_PIXEL_U8) {
        out[0] = (uint8_t)clamp255(Y); // convertColor(u8, Rgb{v, v, v}) == v
    } else {
        int32_t r, g, b;
        if (a.mh == 1 && a.mv == 1) { // 4:4:4 (:2541-2564): chroma stays centred on zero, no clamping before the matrix
            const int32_t Cb = a.cb[(size_t)y * a.chroma_pitch + x], Cr = a.cr[(size_t)y * a.chroma_pitch + x];
            r = wadd(Y, wadd(wmul(91881, Cr), 32768) >> 16);
            g = wsub(Y, wadd(wadd(wmul(22554, Cb), wmul(46802, Cr)), 32768) >> 16);
            b = wadd(Y, wadd(wmul(116130, Cb), 32768) >> 16);
        } else {
            const int mcu_w = 8 * a.mh, mcu_h = 8 * a.mv;
            const int mx = x / mcu_w, my = y / mcu_h, ix = x % mcu_w, iy = y % mcu_h;
            const int32_t *cbp = a.cb + (size_t)my * 8 * a.chroma_pitch + (size_t)mx * 8, *crp = a.cr + (size_t)my * 8 * a.chroma_pitch + (size_t)mx * 8;
            int cx0, cx1, cy0 = iy, cy1 = iy;
            float wx, wy = 0.0f;
            chroma_tap(ix, a.mh == 4 ? 0.25f : 0.5f, &cx0, &cx1, &wx);
            int32_t Cb, Cr;
            if (a.mv == 2) {
                chroma_tap(iy, 0.5f, &cy0, &cy1, &wy);
                const size_t r0 = (size_t)cy0 * a.chroma_pitch, r1 = (size_t)cy1 * a.chroma_pitch;
                Cb = round_half_away(lerp_fma(lerp_fma((float)cbp[r0 + cx0], (float)cbp[r0 + cx1], wx), lerp_fma((float)cbp[r1 + cx0], (float)cbp[r1 + cx1], wx), wy));
                Cr = round_half_away(lerp_fma(lerp_fma((float)crp[r0 + cx0], (float)crp[r0 + cx1], wx), lerp_fma((float)crp[r1 + cx0], (float)crp[r1 + cx1], wx), wy));
            } else {
                const size_t r0 = (size_t)iy * a.chroma_pitch;
                Cb = round_half_away(lerp_fma((float)cbp[r0 + cx0], (float)cbp[r0 + cx1], wx));
                Cr = round_half_away(lerp_fma((float)crp[r0 + cx0], (float)crp[r0 + cx1], wx));
            }
            // Ycbcr(u8){ clamp(Y), clamp(Cb + 128), clamp(Cr + 128) }.to(.rgb): color.zig:1057-1068
            const int64_t yy = clamp255(Y), cb = (int64_t)clamp255(wadd(Cb, 128)) - 128, cr = (int64_t)clamp255(wadd(Cr, 128)) - 128;
            r = (int32_t)((65536 * yy + 91881 * cr + 32768) >> 16);
            g = (int32_t)((65536 * yy - 22554 * cb - 46802 * cr + 32768) >> 16);
            b = (int32_t)((65536 * yy + 116130 * cb + 32768) >> 16);
        }
        out[0] = (uint8_t)clamp255(r); out[1] = (uint8_t)clamp255(g); out[2] = (uint8_t)clamp255(b);
    }
}

int natural_space(int pixel) { return pixel == ZG_PIXEL_U8 ? ZG_CS_GRAY : ZG_CS_RGB; }

int decode_impl(const uint8_t *jpeg, size_t len, const zg_jpeg_limits *limits, const zg_image *dst, int dst_space, int *scan_limit_reached_out, hipStream_t s) {
    ZG_REQUIRE(jpeg != nullptr, ZG_ERR_INVALID_ARGUMENT, "jpeg: null data");
    zg_jpeg_limits lim;
    if (limits) lim = *limits; else zg_jpeg_default_limits(&lim);
    int rc;
    if ((rc = check_image(dst, "dst"))) return rc;
    std::vector<Decoder> holder(1); // large (Huffman tables): keep it off the stack
    Decoder &d = holder[0];
    d.header.precision = 8;
    if ((rc = d.read_stream(jpeg, len, lim))) return rc;
    if (scan_limit_reached_out) *scan_limit_reached_out = d.scan_limit_reached ? 1 : 0;
    if (!d.header.progressive && (rc = d.run_baseline_scan())) return rc;
    if (!d.allocated) JPEG_FAIL("BlockStorageNotAllocated");
    const int nc = d.header.num_components;
    for (int c = 0; c < nc; ++c)
        if (d.comp[c].tq > 3 || !d.have_q[d.comp[c].tq]) JPEG_FAIL("MissingQuantTable");
    ZG_REQUIRE(dst->rows == d.header.height && dst->cols == d.header.width, ZG_ERR_DIMENSION_MISMATCH, "jpeg: the frame is %ux%u, dst is %ux%u",
               d.header.height, d.header.width, dst->rows, dst->cols);

    // chroma of a subsampled frame: only the block at each MCU's origin is ever read (:2573, :2627, :2680): compact it
    const int mh = nc == 3 ? d.comp[0].h : 1, mv = nc == 3 ? d.comp[0].v : 1;
    const bool subsampled = nc == 3 && !(mh == 1 && mv == 1);
    const unsigned cbx = subsampled ? d.bwa / mh : d.bwa, cby = subsampled ? d.bha / mv : d.bha;
    std::vector<int32_t> packed[2];
    const int32_t *chroma[2] = {nullptr, nullptr};
    for (int c = 0; c < 2 && nc == 3; ++c) {
        chroma[c] = d.coef[c + 1].data(); // the whole grid of a 4:4:4 frame, or the compact grid: MCU origins, row-major
        const bool as_stored = !subsampled || (d.compact[c + 1] && d.cgw == cbx && d.cgh == cby && (unsigned)mh == d.fmh && (unsigned)mv == d.fmv);
        if (as_stored) continue;
        packed[c].resize((size_t)cbx * cby * 64); // a chroma component with id 1 has the full grid: gather the origins
        for (unsigned y = 0; y < cby; ++y)
            for (unsigned x = 0; x < cbx; ++x)
                memcpy(packed[c].data() + ((size_t)y * cbx + x) * 64, d.block_or_zero((size_t)y * mv, (size_t)x * mh, (size_t)c + 1), 64 * sizeof(int32_t));
        chroma[c] = packed[c].data();
    }
    const size_t luma_coefs = d.nblocks * 64, chroma_coefs = nc == 3 ? (size_t)cbx * cby * 64 : 0;
    const size_t coef_words = luma_coefs + 2 * chroma_coefs;
    const int native = nc == 1 ? ZG_PIXEL_U8 : ZG_PIXEL_RGB_U8;
    const bool direct = dst->pixel == native && dst_space == natural_space(native);
    const size_t native_bytes = direct ? 0 : (size_t)d.header.width * d.header.height * pixel_size(native);
    // scratch: coefficients | sample planes (same sizes) | native image when a conversion follows
    int32_t *dev = nullptr;
    if ((rc = scratch_alloc((void **)&dev, coef_words * 2 * sizeof(int32_t) + native_bytes + 256, s))) return rc;
    rc = upload_pageable(dev, d.coef[0].data(), luma_coefs * sizeof(int32_t), s); // the host buffers die with this call
    for (int c = 0; c < 2 && nc == 3 && rc == ZG_OK; ++c)
        rc = upload_pageable(dev + luma_coefs + (size_t)c * chroma_coefs, chroma[c], chroma_coefs * sizeof(int32_t), s);
    if (rc) { scratch_free(dev, s); return rc; }

    int32_t *planes = dev + coef_words;
    for (int c = 0; c < nc; ++c) {
        QuantTable qt;
        memcpy(qt.q, d.q[d.comp[c].tq], sizeof qt.q);
        const unsigned bx = c == 0 ? d.bwa : cbx, count = c == 0 ? (unsigned)d.nblocks : cbx * cby;
        const size_t off = c == 0 ? 0 : luma_coefs + (size_t)(c - 1) * chroma_coefs;
        hipLaunchKernelGGL(k_jpeg_idct, dim3(ceil_div(count, 32)), dim3(256), 0, s, (const int32_t *)(dev + off), qt, count, bx, c == 0 ? 128 : 0, planes + off);
    }
    RenderArgs a{planes, planes + luma_coefs, planes + luma_coefs + chroma_coefs, d.bwa * 8, cbx * 8, nc, mh, mv};
    zg_image native_img{(char *)(dev + coef_words * 2) + 128, d.header.width, d.header.height, d.header.width, native};
    const zg_image *target = direct ? dst : &native_img;
    const dim3 grid(ceil_div(d.header.width, 256), d.header.height);
    if (native == ZG_PIXEL_U8) hipLaunchKernelGGL((k_jpeg_render<ZG_PIXEL_U8>), grid, dim3(256), 0, s, a, dimg(target));
    else hipLaunchKernelGGL((k_jpeg_render<ZG_PIXEL_RGB_U8>), grid, dim3(256), 0, s, a, dimg(target));
    rc = hipGetLastError() == hipSuccess ? ZG_OK : ZG_ERR_HIP;
    if (rc == ZG_OK && !direct) rc = zg_convert(&native_img, natural_space(native), dst, dst_space, nullptr, (zg_stream)s); // Image.convert (:2831-2850)
    scratch_free(dev, s);
    return rc;
}

// ==== encoder (jpeg.zig:293-1043) ================================================================================================
// Device: RGB -> YCbCr (u8 fixed point), edge replication into the MCU padding, chroma box averaging, level shift, the LLM
// forward DCT in the reference's integer form and its reciprocal quantisation: one i16 coefficient block per component block.
// Host: the Huffman coder with the reference's fixed tables, which makes the whole file a deterministic function of the
// pixels and options — files are compared with the oracle's byte for byte.
__device__ inline int32_t descale64(int64_t x, int n) { return (int32_t)((x + ((int64_t)1 << (n - 1))) >> n); }
__device__ inline void fdct_pass(const int64_t in[8], int32_t out[8], bool first) { // one pass of fdct8x8_llm (:634-741)
    const int64_t tmp0 = in[0] + in[7], tmp7 = in[0] - in[7], tmp1 = in[1] + in[6], tmp6 = in[1] - in[6];
    const int64_t tmp2 = in[2] + in[5], tmp5 = in[2] - in[5], tmp3 = in[3] + in[4], tmp4 = in[3] - in[4];
    const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    const int shift = first ? 11 : 15; // CONST_BITS -/+ PASS1_BITS
    if (first) { out[0] = (int32_t)((tmp10 + tmp11) << 2); out[4] = (int32_t)((tmp10 - tmp11) << 2); }
    else { out[0] = descale64(tmp10 + tmp11, 2); out[4] = descale64(tmp10 - tmp11, 2); }
    const int64_t z1 = (tmp12 + tmp13) * 4433;
    out[2] = descale64(z1 + tmp13 * 6270, shift);
    out[6] = descale64(z1 + tmp12 * (-15137), shift);
    int64_t z1o = tmp4 + tmp7, z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
    const int64_t z5 = (z3 + z4) * 9633;
    const int64_t t4 = tmp4 * 2446, t5 = tmp5 * 16819, t6 = tmp6 * 25172, t7 = tmp7 * 12299;
    z1o *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
    z3 += z5; z4 += z5;
    out[7] = descale64(t4 + z1o + z3, shift); out[5] = descale64(t5 + z2 + z4, shift); out[3] = descale64(t6 + z2 + z3, shift); out[1] = descale64(t7 + z1o + z4, shift);
}
struct RecipTable { uint32_t r[64]; };
struct ForwardArgs {
    DImg src;          // Image(u8) (grey) or Image(Rgb(u8))
    int component;     // 0 Y (or grey), 1 Cb, 2 Cr
    int hm, vm;        // luma sampling factors of the frame
    unsigned blocks_x, nblocks;
};
// convertColor(Ycbcr, Rgb(u8)) (color.zig:987-1009), one component of it
__device__ inline int ycc_component(const uint8_t *p, int c) {
    const int64_t r = p[0], g = p[1], b = p[2];
    const int64_t v = c == 0 ? (19595 * r + 38470 * g + 7471 * b + 32768) >> 16
                    : c == 1 ? ((-11059 * r - 21710 * g + 32768 * b + 32768) >> 16) + 128
                             : ((32768 * r - 27439 * g - 5329 * b + 32768) >> 16) + 128;
    return (int)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
template <bool GRAY> __global__ __launch_bounds__(256) void k_jpeg_forward(ForwardArgs a, RecipTable recip, int16_t *out) {
    __shared__ int32_t tile[32][72];
    const unsigned local = threadIdx.x >> 3, lane = threadIdx.x & 7, b = blockIdx.x * 32 + local;
    const bool live = b < a.nblocks;
    const int last_row = a.src.rows - 1, last_col = a.src.cols - 1;
    int32_t res[8];
    if (live) {
        const unsigned by = b / a.blocks_x, bx = b % a.blocks_x;
        int64_t in[8];
        const uint8_t *base = (const uint8_t *)a.src.data;
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            int v;
            if (GRAY) {
                const int iy = min((int)(by * 8 + lane), last_row), ix = min((int)(bx * 8) + x, last_col);
                v = base[(size_t)iy * a.src.stride + ix];
            } else if (a.component == 0) {
                const int iy = min((int)(by * 8 + lane), last_row), ix = min((int)(bx * 8) + x, last_col);
                v = ycc_component(base + ((size_t)iy * a.src.stride + ix) * 3, 0);
            } else { // one chroma sample = the mean of the vm x hm pixels under it (each clamped into the image on its own)
                int sum = 0;
                for (int dy = 0; dy < a.vm; ++dy)
                    for (int dx = 0; dx < a.hm; ++dx) {
                        const int iy = min((int)((by * 8 + lane) * a.vm) + dy, last_row), ix = min((int)((bx * 8 + x) * a.hm) + dx, last_col);
                        sum += ycc_component(base + ((size_t)iy * a.src.stride + ix) * 3, a.component);
                    }
                v = sum / (a.hm * a.vm);
            }
            in[x] = v - 128;
        }
        fdct_pass(in, res, true); // pass 1: along row `lane`
#pragma unroll
        for (int k = 0; k < 8; ++k) tile[local][lane * 8 + k] = res[k];
    }
    __syncthreads();
    if (live) {
        int64_t in[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = tile[local][r * 8 + lane];
        fdct_pass(in, res, false); // pass 2: down column `lane`
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int r = 0; r < 8; ++r) { // quantizeWithRecip (:763-770)
            const int32_t v = res[r];
            const int64_t mag = v < 0 ? -(int64_t)v : v;
            int64_t q = (mag * (int64_t)recip.r[r * 8 + lane] + ((int64_t)1 << 23)) >> 24;
            tile[local][r * 8 + lane] = (int32_t)(v < 0 ? -q : q);
        }
    }
    __syncthreads();
    if (live) { // row `lane` of the block: eight i16 = one 16-byte store
        const int32_t *row = &tile[local][lane * 8];
        uint4 packed;
        packed.x = (uint32_t)(uint16_t)row[0] | (uint32_t)(uint16_t)row[1] << 16;
        packed.y = (uint32_t)(uint16_t)row[2] | (uint32_t)(uint16_t)row[3] << 16;
        packed.z = (uint32_t)(uint16_t)row[4] | (uint32_t)(uint16_t)row[5] << 16;
        packed.w = (uint32_t)(uint16_t)row[6] | (uint32_t)(uint16_t)row[7] << 16;
        *(uint4 *)(out + (size_t)b * 64 + lane * 8) = packed;
    }
}

// The reference's tables (jpeg.zig:331-392). Its luma DC table uses the chroma table's code lengths (0 3 1 1 ...), not Annex K's.
const uint8_t kQLuma[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                            18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kQChroma[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                              99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
const uint8_t kBitsDc[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kBitsAcLuma[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125};
const uint8_t kBitsAcChroma[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119};
const uint8_t kValAcLuma[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1,
    0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a,
    0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
    0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kValAcChroma[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1,
    0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
    0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct HuffmanCodes { // buildHuffmanEncoder (:399-415)
    uint16_t code[256];
    uint8_t size[256];
    HuffmanCodes(const uint8_t *bits, const uint8_t *vals) {
        memset(code, 0, sizeof code);
        memset(size, 0, sizeof size);
        unsigned next = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            for (int j = 0; j < bits[len - 1]; ++j, ++k, ++next) { code[vals[k]] = (uint16_t)next; size[vals[k]] = (uint8_t)len; }
            next = (next << 1) & 0xffff;
        }
    }
};
template <bool STUFF> class EntropyWriter { // :417-447; STUFF = false leaves the 0xFF bytes alone (a piece that is spliced later)
  public:
    std::vector<uint8_t> bytes;
    uint32_t window = 0;
    int held = 0;
    void put(uint32_t value, int n) {
        if (n == 0) return;
        window = (window << n) | (value & ((1u << n) - 1));
        held += n;
        while (held >= 8) {
            const uint8_t b = (uint8_t)(window >> (held - 8));
            bytes.push_back(b);
            if (STUFF && b == 0xFF) bytes.push_back(0x00);
            held -= 8;
        }
    }
    void finish() {
        if (held > 0) { const int pad = 8 - held; put((1u << pad) - 1, pad); }
    }
};
inline int bit_length(int32_t v) { int n = 0; for (uint32_t a = (uint32_t)(v < 0 ? -v : v); a; a >>= 1) ++n; return n; }
inline uint32_t extra_bits(int32_t v, int n) { return v >= 0 ? (uint32_t)v : (uint32_t)(((int32_t)1 << n) - 1 + v); }
template <class Writer> void write_block(const int16_t *co, Writer *w, const HuffmanCodes &dc, const HuffmanCodes &ac, int32_t *prev_dc) { // encodeBlock (:771-817), after the quantiser
    const int32_t diff = co[0] - *prev_dc;
    *prev_dc = co[0];
    const int n = bit_length(diff);
    w->put(dc.code[n], dc.size[n]);
    if (n > 0) w->put(extra_bits(diff, n), n);
    int run = 0;
    for (int k = 1; k < 64; ++k) {
        const int32_t v = co[kZigzag[k]];
        if (v == 0) {
            if (++run == 16) { w->put(ac.code[0xF0], ac.size[0xF0]); run = 0; } // sixteen zeros make a ZRL at once, trailing zeros too
            continue;
        }
        const int m = bit_length(v), symbol = (run << 4) | m;
        w->put(ac.code[symbol], ac.size[symbol]);
        w->put(extra_bits(v, m), m);
        run = 0;
    }
    if (run > 0) w->put(ac.code[0x00], ac.size[0x00]);
}
void push_segment(std::vector<uint8_t> *f, int marker, const std::vector<uint8_t> &payload) { // writeSegment (:457-462)
    f->push_back(0xFF);
    f->push_back((uint8_t)marker);
    f->push_back((uint8_t)((payload.size() + 2) >> 8));
    f->push_back((uint8_t)(payload.size() + 2));
    f->insert(f->end(), payload.begin(), payload.end());
}

void quant_tables(int quality_in, uint8_t *ql, uint8_t *qc, RecipTable *rl, RecipTable *rcq) {
    const int quality = quality_in < 1 ? 1 : (quality_in > 100 ? 100 : quality_in); // scaleQuantTables (:464-476)
    const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    for (int i = 0; i < 64; ++i) {
        const int l = (kQLuma[i] * scale + 50) / 100, c = (kQChroma[i] * scale + 50) / 100;
        ql[i] = (uint8_t)(l < 1 ? 1 : (l > 255 ? 255 : l));
        qc[i] = (uint8_t)(c < 1 ? 1 : (c > 255 ? 255 : c));
        rl->r[i] = (uint32_t)round(16777216.0 / ((double)ql[i] * 8.0));   // buildQuantRecipLLM (:749-761)
        rcq->r[i] = (uint32_t)round(16777216.0 / ((double)qc[i] * 8.0));
    }
}

// The entropy-coded segment of the one scan (:862-925, :1020-1037): MCU by MCU, vm x hm luma blocks then Cb then Cr.
// The code is a pure function of the coefficients — the DC predictor of a block is the DC of the block before it in scan
// order, which is known without coding anything — so bands of MCU rows are coded on separate host threads into bit strings
// of their own and spliced: the splice shifts every band to the bit position the one before it ended on, and the 0xFF
// stuffing, which depends on that final byte alignment, happens there. Byte for byte the serial coder's output.
template <class Writer>
void code_mcu_rows(const int16_t *blocks, bool gray, int hm, int vm, unsigned mcus_x, unsigned mcus_y, unsigned row0, unsigned row1, Writer *w) {
    static const uint8_t dc_vals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
    static const HuffmanCodes dc_codes(kBitsDc, dc_vals), ac_luma(kBitsAcLuma, kValAcLuma), ac_chroma(kBitsAcChroma, kValAcChroma);
    const unsigned lbx = mcus_x * hm;
    const size_t luma_blocks = (size_t)lbx * mcus_y * vm, chroma_blocks = gray ? 0 : (size_t)mcus_x * mcus_y;
    const int16_t *cb = blocks + luma_blocks * 64, *cr = cb + chroma_blocks * 64;
    int32_t pred[3] = {0, 0, 0};
    if (row0 > 0) { // the last block of each component in the MCU before this band
        pred[0] = blocks[((size_t)(row0 * vm - 1) * lbx + (lbx - 1)) * 64];
        if (!gray) {
            pred[1] = cb[((size_t)row0 * mcus_x - 1) * 64];
            pred[2] = cr[((size_t)row0 * mcus_x - 1) * 64];
        }
    }
    for (unsigned my = row0; my < row1; ++my)
        for (unsigned mx = 0; mx < mcus_x; ++mx) {
            for (int vy = 0; vy < vm; ++vy)
                for (int hx = 0; hx < hm; ++hx)
                    write_block(blocks + ((size_t)(my * vm + vy) * lbx + (mx * hm + hx)) * 64, w, dc_codes, ac_luma, &pred[0]);
            if (!gray) {
                write_block(cb + ((size_t)my * mcus_x + mx) * 64, w, dc_codes, ac_chroma, &pred[1]);
                write_block(cr + ((size_t)my * mcus_x + mx) * 64, w, dc_codes, ac_chroma, &pred[2]);
            }
        }
}
// Appends n whole bytes to a writer that holds 0..7 pending bits: eight source bytes per step, shifted into place as one
// big-endian word; a word with a 0xFF byte in it (one in thirty or so) goes byte by byte for the stuffing.
void splice(const uint8_t *src, size_t n, EntropyWriter<true> *w) {
    const int k = w->held; // pending bits, unchanged by whole bytes
    uint64_t carry = w->window & ((1u << k) - 1);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t v;
        memcpy(&v, src + i, 8);
        v = __builtin_bswap64(v);
        const uint64_t word = k ? (carry << (64 - k)) | (v >> k) : v;
        carry = v & (((uint64_t)1 << k) - 1);
        const uint64_t inv = ~word; // a zero byte in ~word is a 0xFF byte in word
        if (((inv - 0x0101010101010101ull) & ~inv & 0x8080808080808080ull) == 0) {
            const uint64_t be = __builtin_bswap64(word);
            const size_t at = w->bytes.size();
            w->bytes.resize(at + 8);
            memcpy(w->bytes.data() + at, &be, 8);
        } else {
            for (int b = 56; b >= 0; b -= 8) {
                const uint8_t byte = (uint8_t)(word >> b);
                w->bytes.push_back(byte);
                if (byte == 0xFF) w->bytes.push_back(0x00);
            }
        }
    }
    w->window = (uint32_t)carry;
    for (; i < n; ++i) w->put(src[i], 8);
}
void entropy_code(const int16_t *blocks, bool gray, int hm, int vm, unsigned mcus_x, unsigned mcus_y, std::vector<uint8_t> *file) {
    const size_t total_blocks = (size_t)mcus_x * mcus_y * ((size_t)hm * vm + (gray ? 0 : 2));
    EntropyWriter<true> w;
    w.bytes.reserve(total_blocks * 24);
    const unsigned bands = (unsigned)std::min<size_t>({(size_t)host_threads(), (size_t)mcus_y, total_blocks / 8192});
    if (bands <= 1) {
        code_mcu_rows(blocks, gray, hm, vm, mcus_x, mcus_y, 0, mcus_y, &w);
    } else {
        std::vector<EntropyWriter<false>> piece(bands);
        std::atomic<bool> out_of_memory{false};
        std::atomic<unsigned> next{0};
        auto work = [&]() {
            for (unsigned b = next.fetch_add(1); b < bands; b = next.fetch_add(1)) {
                const unsigned row0 = (unsigned)((uint64_t)mcus_y * b / bands), row1 = (unsigned)((uint64_t)mcus_y * (b + 1) / bands);
                try {
                    EntropyWriter<false> mine; // on this thread's stack: neighbours in `piece` would share cache lines
                    mine.bytes.reserve((size_t)(row1 - row0) * mcus_x * ((size_t)hm * vm + (gray ? 0 : 2)) * 24);
                    code_mcu_rows(blocks, gray, hm, vm, mcus_x, mcus_y, row0, row1, &mine);
                    piece[b] = std::move(mine);
                } catch (const std::bad_alloc &) { // nothing may unwind out of a thread
                    out_of_memory = true;
                }
            }
        };
        std::vector<std::thread> crew;
        crew.reserve(bands - 1);
        try {
            for (unsigned t = 1; t < bands; ++t) crew.emplace_back(work);
        } catch (const std::system_error &) { // no more threads to be had: the ones there are share the bands
        }
        work();
        for (std::thread &t : crew) t.join();
        if (out_of_memory) throw std::bad_alloc();
        size_t room = 16;
        for (const EntropyWriter<false> &p : piece) room += p.bytes.size() + p.bytes.size() / 32 + 16;
        w.bytes.reserve(room);
        for (const EntropyWriter<false> &p : piece) {
            splice(p.bytes.data(), p.bytes.size(), &w);
            w.put(p.window, p.held); // the bits of its last, incomplete byte
        }
    }
    w.finish();
    file->insert(file->end(), w.bytes.begin(), w.bytes.end());
}

// The host half of encode: the container (encodeRgb :929-975, encodeGrayscale :977-1043) around the coefficient blocks the
// device wrote (luma on its (mcus_y vm) x (mcus_x hm) grid, then Cb, then Cr on the MCU grid; natural order within a block).
int write_file(const int16_t *blocks, uint32_t rows, uint32_t cols, bool gray, const zg_jpeg_encode_options &opt, std::vector<uint8_t> *file_out) {
    const int hm = gray || opt.subsampling == 0 ? 1 : 2, vm = !gray && opt.subsampling == 2 ? 2 : 1;
    const unsigned mcus_x = ceil_div(cols, 8u * hm), mcus_y = ceil_div(rows, 8u * vm);
    uint8_t ql[64], qc[64];
    RecipTable rl, rcq;
    quant_tables(opt.quality, ql, qc, &rl, &rcq);
    // the container (encodeRgb :929-975, encodeGrayscale :977-1043)
    std::vector<uint8_t> &file = *file_out;
    file.assign({0xFF, 0xD8});
    const uint8_t dh = (uint8_t)(opt.density_dpi >> 8), dl = (uint8_t)opt.density_dpi;
    push_segment(&file, 0xE0, {'J', 'F', 'I', 'F', 0, 1, 1, 1, dh, dl, dh, dl, 0, 0});
    if (opt.comment) push_segment(&file, 0xFE, std::vector<uint8_t>(opt.comment, opt.comment + opt.comment_len));
    std::vector<uint8_t> seg = {0x00};
    for (int i = 0; i < 64; ++i) seg.push_back(ql[kZigzag[i]]);
    if (!gray) {
        seg.push_back(0x01);
        for (int i = 0; i < 64; ++i) seg.push_back(qc[kZigzag[i]]);
    }
    push_segment(&file, 0xDB, seg);
    const uint8_t hi_r = (uint8_t)(rows >> 8), lo_r = (uint8_t)rows, hi_c = (uint8_t)(cols >> 8), lo_c = (uint8_t)cols;
    if (gray) push_segment(&file, 0xC0, {8, hi_r, lo_r, hi_c, lo_c, 1, 1, 0x11, 0});
    else push_segment(&file, 0xC0, {8, hi_r, lo_r, hi_c, lo_c, 3, 1, (uint8_t)(hm << 4 | vm), 0, 2, 0x11, 1, 3, 0x11, 1});
    const uint8_t dc_vals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
    seg.assign(1, 0x00);
    seg.insert(seg.end(), kBitsDc, kBitsDc + 16); seg.insert(seg.end(), dc_vals, dc_vals + 12);
    seg.push_back(0x10);
    seg.insert(seg.end(), kBitsAcLuma, kBitsAcLuma + 16); seg.insert(seg.end(), kValAcLuma, kValAcLuma + 162);
    if (!gray) {
        seg.push_back(0x01);
        seg.insert(seg.end(), kBitsDc, kBitsDc + 16); seg.insert(seg.end(), dc_vals, dc_vals + 12);
        seg.push_back(0x11);
        seg.insert(seg.end(), kBitsAcChroma, kBitsAcChroma + 16); seg.insert(seg.end(), kValAcChroma, kValAcChroma + 162);
    }
    push_segment(&file, 0xC4, seg);
    if (gray) push_segment(&file, 0xDA, {1, 1, 0x00, 0, 63, 0});
    else push_segment(&file, 0xDA, {3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0});

    entropy_code(blocks, gray, hm, vm, mcus_x, mcus_y, &file);
    file.push_back(0xFF);
    file.push_back(0xD9);
    return ZG_OK;
}

int encode_impl(const zg_image *src, int src_space, const zg_jpeg_encode_options *options, uint8_t **out, size_t *out_len, hipStream_t s) {
    ZG_REQUIRE(out && out_len, ZG_ERR_INVALID_ARGUMENT, "jpeg encode: null output");
    *out = nullptr;
    *out_len = 0;
    zg_jpeg_encode_options opt;
    if (options) opt = *options; else zg_jpeg_default_encode_options(&opt);
    int rc;
    if ((rc = check_image(src, "src"))) return rc;
    if (src->rows == 0 || src->cols == 0) JPEG_FAIL("InvalidImageDimensions");
    if (src->rows > 65535 || src->cols > 65535) JPEG_FAIL("ImageTooLarge");
    ZG_REQUIRE(opt.subsampling >= 0 && opt.subsampling <= 2, ZG_ERR_INVALID_ARGUMENT, "jpeg encode: subsampling %d (0 yuv444, 1 yuv422, 2 yuv420)", opt.subsampling);
    // writeSegment's length is a u16 of payload + 2 (:457-462; the reference traps in @intCast beyond it): refuse, never wrap
    ZG_REQUIRE(!opt.comment || opt.comment_len <= 65533, ZG_ERR_INVALID_ARGUMENT, "jpeg encode: comment of %zu bytes does not fit a COM segment (65533 at most)", opt.comment_len);
    const bool gray = src->pixel == ZG_PIXEL_U8;                         // T == u8 (:321)
    const bool direct = gray || (src->pixel == ZG_PIXEL_RGB_U8 && src_space == ZG_CS_RGB);
    const int hm = gray || opt.subsampling == 0 ? 1 : 2, vm = !gray && opt.subsampling == 2 ? 2 : 1;
    const unsigned mcus_x = ceil_div(src->cols, 8u * hm), mcus_y = ceil_div(src->rows, 8u * vm);
    const unsigned lbx = mcus_x * hm, lby = mcus_y * vm;
    const size_t luma_blocks = (size_t)lbx * lby, chroma_blocks = gray ? 0 : (size_t)mcus_x * mcus_y, total_blocks = luma_blocks + 2 * chroma_blocks;
    const size_t rgb_bytes = direct ? 0 : ((size_t)src->rows * src->cols * 3 + 255) / 256 * 256;
    char *dev = nullptr;
    if ((rc = scratch_alloc((void **)&dev, rgb_bytes + total_blocks * 64 * sizeof(int16_t), s))) return rc;
    zg_image rgb{dev, src->cols, src->rows, src->cols, ZG_PIXEL_RGB_U8};
    if (!direct) rc = zg_convert(src, src_space, &rgb, ZG_CS_RGB, nullptr, (zg_stream)s); // image.convert(Rgb) (:323-327)
    const zg_image *img = direct ? src : &rgb;
    int16_t *coef = (int16_t *)(dev + rgb_bytes);

    uint8_t ql[64], qc[64];
    RecipTable rl, rcq;
    quant_tables(opt.quality, ql, qc, &rl, &rcq);
    if (rc == ZG_OK) {
        for (int c = 0; c < (gray ? 1 : 3); ++c) {
            const ForwardArgs a{dimg(img), c, hm, vm, c == 0 ? lbx : mcus_x, (unsigned)(c == 0 ? luma_blocks : chroma_blocks)};
            int16_t *dstc = coef + (c == 0 ? 0 : (luma_blocks + (size_t)(c - 1) * chroma_blocks)) * 64;
            if (gray) hipLaunchKernelGGL((k_jpeg_forward<true>), dim3(ceil_div(a.nblocks, 32)), dim3(256), 0, s, a, rl, dstc);
            else hipLaunchKernelGGL((k_jpeg_forward<false>), dim3(ceil_div(a.nblocks, 32)), dim3(256), 0, s, a, c == 0 ? rl : rcq, dstc);
        }
        if (hipGetLastError() != hipSuccess) rc = ZG_ERR_HIP;
    }
    std::vector<int16_t> host;
    if (rc == ZG_OK) {
        host.resize(total_blocks * 64);
        rc = download_pageable(host.data(), coef, host.size() * sizeof(int16_t), s);
    }
    scratch_free(dev, s);
    if (rc) return rc;

    std::vector<uint8_t> file;
    if ((rc = write_file(host.data(), src->rows, src->cols, gray, opt, &file))) return rc;
    uint8_t *mem = (uint8_t *)malloc(file.size());
    if (!mem) { set_error("jpeg encode: out of host memory"); return ZG_ERR_OUT_OF_MEMORY; }
    memcpy(mem, file.data(), file.size());
    *out = mem;
    *out_len = file.size();
    return ZG_OK;
}

} // namespace
} // namespace zg

using namespace zg;

// std::vector growth inside the host layers can throw; nothing may unwind through the C ABI
template <typename F> static int no_throw(F &&body) {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        set_error("out of host memory");
        return ZG_ERR_OUT_OF_MEMORY;
    }
}

extern "C" {

void zg_jpeg_default_limits(zg_jpeg_limits *l) { // jpeg.zig:19-33
    l->max_jpeg_bytes = l->max_marker_bytes = (size_t)100 * 1024 * 1024;
    l->max_width = l->max_height = 8192;
    l->max_pixels = 67108864ull;
    l->max_blocks = 1048576;
    l->max_scans = 64;
}

// jpeg.getInfo (:77-179): a forward-only hunt for the first SOFn; running out of bytes anywhere is error.EndOfStream.
int zg_jpeg_info(const uint8_t *d, size_t len, const zg_jpeg_limits *limits, zg_jpeg_header *out) {
    ZG_REQUIRE(d && out, ZG_ERR_INVALID_ARGUMENT, "jpeg info: null argument");
    zg_jpeg_limits lim;
    if (limits) lim = *limits; else zg_jpeg_default_limits(&lim);
    if (len < 2) JPEG_FAIL("EndOfStream");
    if (d[0] != 0xFF || d[1] != 0xD8) JPEG_FAIL("InvalidJpegFile");
    size_t at = 2, seen = 2, markers = 0;
    for (;;) {
        for (;;) { // hunt for 0xFF
            if (at >= len) JPEG_FAIL("EndOfStream");
            const uint8_t byte = d[at++];
            if (++seen > lim.max_jpeg_bytes) JPEG_FAIL("ImageTooLarge");
            if (byte == 0xFF) break;
        }
        if (at >= len) JPEG_FAIL("EndOfStream");
        int m = d[at++];
        ++seen;
        while (m == 0xFF) { // padding
            if (at >= len) JPEG_FAIL("EndOfStream");
            m = d[at++];
            if (++seen > lim.max_jpeg_bytes) JPEG_FAIL("ImageTooLarge");
        }
        if (m == 0x00) continue;
        if (++markers > 10000) JPEG_FAIL("ImageTooLarge");
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD8)) continue; // TEM, RSTn, SOI: no payload
        if (m == 0xD9) JPEG_FAIL("MissingSOF");
        if (len - at < 2) JPEG_FAIL("EndOfStream");
        const unsigned length = be16(d + at);
        at += 2;
        seen += 2;
        if (length < 2) JPEG_FAIL("InvalidMarker");
        const bool sof = m >= 0xC0 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC;
        const unsigned payload = length - 2;
        if (sof) {
            if (payload < 6) JPEG_FAIL("InvalidSOF");
            if (seen + payload > lim.max_jpeg_bytes) JPEG_FAIL("ImageTooLarge");
            if (len - at < 6) JPEG_FAIL("EndOfStream");
            zg_jpeg_header h{};
            h.precision = d[at];
            h.height = be16(d + at + 1);
            h.width = be16(d + at + 3);
            h.num_components = d[at + 5];
            h.progressive = m == 0xC2;
            h.subsampling = -1;
            if (h.num_components == 3 && payload - 6 >= 9) {
                if (len - at < 15) JPEG_FAIL("EndOfStream");
                const uint8_t luma = d[at + 7], cb = d[at + 10], cr = d[at + 13]; // the sampling byte of each component triple
                if (cb == 0x11 && cr == 0x11) h.subsampling = luma == 0x11 ? 0 : (luma == 0x21 ? 1 : (luma == 0x22 ? 2 : -1));
            }
            *out = h;
            return ZG_OK;
        }
        if (seen + payload > lim.max_jpeg_bytes) JPEG_FAIL("ImageTooLarge");
        const size_t skip = len - at < payload ? len - at : payload;
        at += skip;
        seen += skip;
    }
}

void zg_jpeg_default_encode_options(zg_jpeg_encode_options *o) { // EncodeOptions (jpeg.zig:284-290)
    o->quality = 90;
    o->subsampling = 2;
    o->density_dpi = 72;
    o->comment = nullptr;
    o->comment_len = 0;
}
int zg_jpeg_encode(const zg_image *src, int src_space, const zg_jpeg_encode_options *options, uint8_t **out, size_t *out_len, zg_stream stream) {
    return no_throw([&] { return encode_impl(src, src_space, options, out, out_len, as_stream(stream)); });
}
int zg_jpeg_encode_host(const zg_image *src, int src_space, const zg_jpeg_encode_options *options, uint8_t **out, size_t *out_len) {
    HostStage a;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    return no_throw([&] { return encode_impl(&a.dev, src_space, options, out, out_len, nullptr); });
}
int zg_jpeg_encode_blocks(const int16_t *blocks, uint32_t rows, uint32_t cols, int gray, const zg_jpeg_encode_options *options, uint8_t **out, size_t *out_len) {
    ZG_REQUIRE(out && out_len, ZG_ERR_INVALID_ARGUMENT, "jpeg encode blocks: null output");
    *out = nullptr;
    *out_len = 0;
    ZG_REQUIRE(blocks != nullptr, ZG_ERR_INVALID_ARGUMENT, "jpeg encode blocks: null input");
    zg_jpeg_encode_options opt;
    if (options) opt = *options; else zg_jpeg_default_encode_options(&opt);
    if (rows == 0 || cols == 0) JPEG_FAIL("InvalidImageDimensions");
    if (rows > 65535 || cols > 65535) JPEG_FAIL("ImageTooLarge");
    ZG_REQUIRE(opt.subsampling >= 0 && opt.subsampling <= 2, ZG_ERR_INVALID_ARGUMENT, "jpeg encode: subsampling %d (0 yuv444, 1 yuv422, 2 yuv420)", opt.subsampling);
    ZG_REQUIRE(!opt.comment || opt.comment_len <= 65533, ZG_ERR_INVALID_ARGUMENT, "jpeg encode: comment of %zu bytes does not fit a COM segment (65533 at most)", opt.comment_len);
    { // caller-made blocks must stay inside what the baseline Huffman tables can code: AC magnitudes of 10 bits, DC
      // differences of 11 (tables of 11 / 12 size categories; a larger value would index past them)
        const int hm = gray || opt.subsampling == 0 ? 1 : 2, vm = !gray && opt.subsampling == 2 ? 2 : 1;
        const size_t mx = ceil_div(cols, 8u * hm), my = ceil_div(rows, 8u * vm), n_blocks = mx * hm * my * vm + (gray ? 0 : 2 * mx * my);
        for (size_t b = 0; b < n_blocks; ++b) {
            const int16_t *co = blocks + b * 64;
            bool ok = co[0] >= -1024 && co[0] <= 1023;
            for (int i = 1; i < 64; ++i) ok = ok && co[i] >= -1023 && co[i] <= 1023;
            ZG_REQUIRE(ok, ZG_ERR_INVALID_ARGUMENT, "jpeg encode blocks: block %zu holds a coefficient outside the baseline range (DC -1024..1023, AC -1023..1023)", b);
        }
    }
    return no_throw([&]() -> int {
        std::vector<uint8_t> file;
        const int rc = write_file(blocks, rows, cols, gray != 0, opt, &file);
        if (rc) return rc;
        uint8_t *mem = (uint8_t *)malloc(file.size());
        if (!mem) { set_error("jpeg encode: out of host memory"); return ZG_ERR_OUT_OF_MEMORY; }
        memcpy(mem, file.data(), file.size());
        *out = mem;
        *out_len = file.size();
        return ZG_OK;
    });
}
void zg_jpeg_free(void *p) { free(p); }

int zg_jpeg_probe(const uint8_t *jpeg, size_t len, const zg_jpeg_limits *limits, zg_jpeg_header *header_out, int *scan_limit_reached_out) {
    ZG_REQUIRE(jpeg != nullptr, ZG_ERR_INVALID_ARGUMENT, "jpeg probe: null data");
    zg_jpeg_limits lim;
    if (limits) lim = *limits; else zg_jpeg_default_limits(&lim);
    return no_throw([&]() -> int {
        std::vector<Decoder> holder(1);
        holder[0].header.precision = 8;
        const int rc = holder[0].read_stream(jpeg, len, lim);
        if (rc) return rc;
        if (header_out) { *header_out = holder[0].header; header_out->subsampling = -1; }
        if (scan_limit_reached_out) *scan_limit_reached_out = holder[0].scan_limit_reached ? 1 : 0;
        return ZG_OK;
    });
}
int zg_jpeg_coefficient_hash(const uint8_t *jpeg, size_t len, const zg_jpeg_limits *limits, uint64_t *hash_out) {
    ZG_REQUIRE(jpeg && hash_out, ZG_ERR_INVALID_ARGUMENT, "jpeg coefficient hash: null argument");
    zg_jpeg_limits lim;
    if (limits) lim = *limits; else zg_jpeg_default_limits(&lim);
    return no_throw([&]() -> int {
        std::vector<Decoder> holder(1);
        Decoder &d = holder[0];
        d.header.precision = 8;
        int rc = d.read_stream(jpeg, len, lim);
        if (rc == ZG_OK && !d.header.progressive) rc = d.run_baseline_scan();
        if (rc) return rc;
        if (!d.allocated) JPEG_FAIL("BlockStorageNotAllocated");
        uint64_t h = 1469598103934665603ull; // FNV-1a, one 32-bit coefficient per step, component by component
        for (int c = 0; c < d.header.num_components; ++c)
            for (size_t ay = 0; ay < d.bha; ++ay)
                for (size_t ax = 0; ax < d.bwa; ++ax) { // every position of the luma grid, zeros where nothing is stored
                    const int32_t *p = d.block_or_zero(ay, ax, (size_t)c);
                    for (int i = 0; i < 64; ++i) {
                        h ^= (uint32_t)p[i];
                        h *= 1099511628211ull;
                    }
                }
        *hash_out = h;
        return ZG_OK;
    });
}
int zg_jpeg_decode(const uint8_t *jpeg, size_t len, const zg_jpeg_limits *limits, const zg_image *dst, int dst_space, int *scan_limit_reached_out, zg_stream stream) {
    return no_throw([&] { return decode_impl(jpeg, len, limits, dst, dst_space, scan_limit_reached_out, as_stream(stream)); });
}
int zg_jpeg_decode_host(const uint8_t *jpeg, size_t len, const zg_jpeg_limits *limits, const zg_image *dst, int dst_space, int *scan_limit_reached_out) {
    HostStage d;
    int rc;
    if ((rc = d.upload(dst, false, true))) return rc;
    if ((rc = no_throw([&] { return decode_impl(jpeg, len, limits, &d.dev, dst_space, scan_limit_reached_out, nullptr); }))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return d.finish();
}

} // extern "C"
