// png_codec.hip — the host I/O edge of the path: PNG files <-> device images (reference src/codecs/png.zig).
//
// Split by what each side is good at:
//   host    the chunk layer (order rules, CRCs, limits: png.zig:513-794), zlib inflate / deflate (the reference uses Zig's
//           std.compress.flate; here libz), and de-filtering, a byte recurrence along AND across rows (png.zig:1442-1533)
//   device  everything per pixel: sample unpacking for all fifteen colour-type / bit-depth forms, palette and tRNS lookup,
//           Adam7 placement, conversion to the requested Image(T) (png.zig:852-1146, :1805-2053) on the way in; all five
//           row filters, their costs, the adaptive selection and the filtered stream (png.zig:1535-1719) on the way out
// The scan data crosses PCIe once, already de-filtered, at its file bit depth (a 1-bit image uploads 1 bit per pixel).
#include "zg_common.h"

#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <new>
#include <system_error>
#include <thread>
#include <vector>
#include <zlib.h>

namespace zg {
// A byte buffer whose resize() does not clear: the inflater (or the download) overwrites all of it, and 64 MiB of zeros
// first is a tenth of a large file's decode time.
template <class T> struct NoInit : std::allocator<T> {
    template <class U> struct rebind { using other = NoInit<U>; };
    template <class U, class... A> void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U; else ::new ((void *)p) U(std::forward<A>(a)...);
    }
};
using ScanBytes = std::vector<uint8_t, NoInit<uint8_t>>;
// The scanline buffer of the last call stays with the calling thread (up to 256 MiB): the next frame of a similar size
// finds its pages already mapped instead of faulting 4 KiB at a time under the inflater.
struct ScanLease {
    ScanBytes bytes;
    static ScanBytes &kept() {
        static thread_local ScanBytes buffer;
        return buffer;
    }
    ScanLease() { bytes.swap(kept()); }
    ~ScanLease() {
        if (bytes.capacity() <= ((size_t)256 << 20) && bytes.capacity() > kept().capacity()) bytes.swap(kept());
    }
};

namespace {

int png_fail(const char *zig_error, const char *where) {
    set_error("%s (%s)", zig_error, where);
    return ZG_ERR_CODEC;
}
#define PNG_FAIL(name) return png_fail(name, __func__)

const uint8_t kSignature[8] = {137, 80, 78, 71, 13, 10, 26, 10};

inline uint32_t load_be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
inline void store_be32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
inline bool over(uint64_t limit, uint64_t value) { return limit != 0 && value > limit; }
inline bool is_type(const uint8_t *t, const char *name) { return memcmp(t, name, 4) == 0; }

// ---- geometry of the scan data (png.zig:135-272) ---------------------------------------------------------------------------
__host__ __device__ inline int png_channels(int color_type) { return color_type == 2 ? 3 : (color_type == 4 ? 2 : (color_type == 6 ? 4 : 1)); }

struct PassGeom { // one Adam7 pass, or the whole image when not interlaced
    uint32_t x0, y0, dx, dy, w, h;
    size_t row_bytes, offset; // payload bytes per row (a filter byte precedes each), offset of the pass in the scan data
};
struct ScanLayout {
    PassGeom pass[7];
    int npass;
    size_t total;
    bool overflow; // the byte count left usize: error.ImageTooLarge (png.zig:229-245)
};
const uint32_t kAdam7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};

// scanDataLength / adam7TotalSize (png.zig:219-245): every product and sum is checked (std.math.mul / std.math.add in the
// reference); a header whose scan data does not fit a usize sets `overflow`, which the chunk layer reports as
// error.ImageTooLarge before anything is sized from `total`.
ScanLayout scan_layout(const zg_png_header &h) {
    ScanLayout L{};
    const size_t bits = (size_t)png_channels(h.color_type) * h.bit_depth; // <= 64
    auto pass_bytes = [&](size_t row_bytes, uint32_t rows, size_t *out) { // (row_bytes + 1) * rows, checked
        size_t stride;
        return !__builtin_add_overflow(row_bytes, (size_t)1, &stride) && !__builtin_mul_overflow(stride, (size_t)rows, out);
    };
    if (h.interlace_method != 1) {
        L.pass[0] = PassGeom{0, 0, 1, 1, h.width, h.height, ((size_t)h.width * bits + 7) / 8, 0};
        L.npass = 1;
        if (!pass_bytes(L.pass[0].row_bytes, h.height, &L.total)) L.overflow = true, L.total = 0;
        return L;
    }
    for (int p = 0; p < 7; ++p) {
        const uint32_t x0 = kAdam7[p][0], y0 = kAdam7[p][1], dx = kAdam7[p][2], dy = kAdam7[p][3];
        const uint32_t w = h.width > x0 ? (uint32_t)(((uint64_t)h.width - x0 + dx - 1) / dx) : 0,
                       ht = h.height > y0 ? (uint32_t)(((uint64_t)h.height - y0 + dy - 1) / dy) : 0;
        PassGeom g{x0, y0, dx, dy, w, ht, ((size_t)w * bits + 7) / 8, L.total};
        if (w == 0 || ht == 0) g.w = g.h = 0; // an empty pass has no bytes at all
        else {
            size_t pass_total;
            if (!pass_bytes(g.row_bytes, ht, &pass_total) || __builtin_add_overflow(L.total, pass_total, &L.total)) {
                L.overflow = true;
                L.total = 0;
                L.npass = 7;
                return L;
            }
        }
        L.pass[p] = g;
    }
    L.npass = 7;
    return L;
}
// Longest prefix of `len` scan bytes that ends on a row boundary; whole passes are kept (png.zig:254-272).
size_t complete_prefix(size_t len, const ScanLayout &L) {
    size_t kept = 0;
    for (int p = 0; p < L.npass; ++p) {
        const PassGeom &g = L.pass[p];
        if (g.w == 0) continue;
        const size_t stride = g.row_bytes + 1, total = stride * g.h;
        if (len - kept < total) return kept + (len - kept) / stride * stride;
        kept += total;
    }
    return kept;
}

// ---- the chunk layer (png.decode, png.zig:629-794) ------------------------------------------------------------------------
struct PngFile {
    zg_png_header header{};
    uint8_t palette[256 * 3];
    int palette_len = -1; // entries; -1 = no PLTE
    uint8_t trns[256];
    int trns_len = -1;
    std::vector<uint8_t> idat;
    bool truncated = false;
};

int parse_ihdr(const uint8_t *d, uint32_t length, zg_png_header *h) { // parseHeader :561-625
    if (length != 13) PNG_FAIL("InvalidHeaderLength");
    *h = zg_png_header{};
    h->width = load_be32(d);
    h->height = load_be32(d + 4);
    if (h->width == 0 || h->height == 0) PNG_FAIL("InvalidDimensions");
    const int depth = d[8], ct = d[9];
    if (ct != 0 && ct != 2 && ct != 3 && ct != 4 && ct != 6) PNG_FAIL("InvalidColorType");
    const bool small = depth == 1 || depth == 2 || depth == 4;
    const bool ok = ct == 0 ? (small || depth == 8 || depth == 16) : ct == 3 ? (small || depth == 8) : (depth == 8 || depth == 16);
    if (!ok) PNG_FAIL("InvalidBitDepth");
    if (d[10] != 0) PNG_FAIL("UnsupportedCompressionMethod");
    if (d[11] != 0) PNG_FAIL("UnsupportedFilterMethod");
    if (d[12] > 1) PNG_FAIL("UnsupportedInterlaceMethod");
    h->bit_depth = (uint8_t)depth;
    h->color_type = (uint8_t)ct;
    h->interlace_method = d[12];
    return ZG_OK;
}

uint32_t chunk_crc(const uint8_t *type_and_data, size_t n) { return (uint32_t)crc32(crc32(0L, Z_NULL, 0), type_and_data, (uInt)n); }

int read_chunks(const uint8_t *png, size_t len, const zg_png_limits &lim, PngFile *f) {
    if (len < 8 || memcmp(png, kSignature, 8) != 0) PNG_FAIL("InvalidPngSignature");
    if (over(lim.max_png_bytes, len)) PNG_FAIL("PngDataTooLarge");
    const uint8_t *body = png + 8;
    const size_t n = len - 8;
    size_t at = 0, chunk_bytes = 0, idat_bytes = 0, count = 0;
    bool have_header = false, plte = false, trns = false, idat = false, iend = false, iccp = false, srgb = false, idat_closed = false;
    while (at + 8 <= n) {
        // ChunkReader.nextChunk (:521-557): a chunk running past the end is returned cut, its CRC unverifiable
        const uint32_t declared = load_be32(body + at);
        const uint8_t *type = body + at + 4, *data = type + 4;
        at += 8;
        size_t have = declared;
        bool cut = false;
        if ((uint64_t)at + declared + 4 > n) {
            have = (size_t)(declared < n - at ? declared : n - at);
            at = n;
            cut = true;
        } else {
            at += (size_t)declared + 4;
            if (chunk_crc(type, (size_t)declared + 4) != load_be32(data + declared)) PNG_FAIL("InvalidCrc");
        }
        if (over(lim.max_chunks, ++count)) PNG_FAIL("TooManyChunks");
        chunk_bytes += have;
        if (over(lim.max_chunk_bytes, chunk_bytes)) PNG_FAIL("ChunkDataLimitExceeded");
        const bool is_idat = is_type(type, "IDAT"), is_ihdr = is_type(type, "IHDR");
        if (cut && !is_idat) { // cut inside a non-IDAT chunk: fatal before the pixel data, tolerable after (:655-658)
            if (!idat) PNG_FAIL("InvalidChunkLength");
            break;
        }
        if (!have_header && !is_ihdr) PNG_FAIL("ChunkBeforeHeader");
        if (idat && !is_idat) idat_closed = true;
        const uint32_t length = (uint32_t)have;
        int rc;
        if (is_ihdr) {
            if (have_header) PNG_FAIL("MultipleHeaders");
            if ((rc = parse_ihdr(data, length, &f->header))) return rc;
            have_header = true;
            if (over(lim.max_width, f->header.width) || over(lim.max_height, f->header.height)) PNG_FAIL("ImageTooLarge");
            if (over(lim.max_pixels, (uint64_t)f->header.width * f->header.height)) PNG_FAIL("ImageTooLarge");
        } else if (is_type(type, "PLTE")) {
            if (f->header.color_type == 0 || f->header.color_type == 4) PNG_FAIL("PaletteForbiddenForColorType");
            if (idat) PNG_FAIL("PaletteAfterImageData");
            if (f->palette_len >= 0) PNG_FAIL("DuplicatePalette");
            if (length % 3 != 0) PNG_FAIL("InvalidPaletteLength");
            if (length / 3 > 256) PNG_FAIL("PaletteTooLarge");
            f->palette_len = (int)(length / 3);
            memcpy(f->palette, data, length);
            plte = true;
        } else if (is_type(type, "tRNS")) {
            if (trns) PNG_FAIL("MultipleTransparencyChunks");
            if (idat) PNG_FAIL("TransparencyAfterImageData");
            switch (f->header.color_type) {
            case 0: if (length != 2) PNG_FAIL("InvalidTransparencyLength"); break;
            case 2: if (length != 6) PNG_FAIL("InvalidTransparencyLength"); break;
            case 3:
                if (!plte) PNG_FAIL("TransparencyBeforePalette");
                if (f->palette_len < 0) PNG_FAIL("MissingPalette");
                if (length > (uint32_t)f->palette_len) PNG_FAIL("InvalidTransparencyLength");
                break;
            default: PNG_FAIL("InvalidTransparencyForColorType");
            }
            f->trns_len = (int)length;
            memcpy(f->trns, data, length);
            trns = true;
        } else if (is_type(type, "gAMA")) {
            if (plte) PNG_FAIL("GammaAfterPalette");
            if (idat) PNG_FAIL("GammaAfterImageData");
            if (length != 4) PNG_FAIL("InvalidGammaLength");
            f->header.has_gamma = 1;
            f->header.gamma = (float)load_be32(data) / 100000.0f;
        } else if (is_type(type, "sRGB")) {
            if (plte) PNG_FAIL("SrgbAfterPalette");
            if (idat) PNG_FAIL("SrgbAfterImageData");
            if (length != 1) PNG_FAIL("InvalidSrgbLength");
            if (iccp) PNG_FAIL("ColorProfileConflict");
            if (data[0] > 3) PNG_FAIL("InvalidSrgbIntent");
            f->header.has_srgb = 1;
            f->header.srgb_intent = data[0];
            srgb = true;
        } else if (is_type(type, "iCCP")) {
            if (plte) PNG_FAIL("IccpAfterPalette");
            if (idat) PNG_FAIL("IccpAfterImageData");
            if (srgb) PNG_FAIL("ColorProfileConflict");
            iccp = true;
        } else if (is_idat) {
            if (idat_closed) PNG_FAIL("NonConsecutiveIdatChunks");
            if (f->header.color_type == 3 && f->palette_len < 0) PNG_FAIL("MissingPalette");
            idat_bytes += have;
            if (over(lim.max_idat_bytes, idat_bytes)) PNG_FAIL("ImageDataLimitExceeded");
            f->idat.insert(f->idat.end(), data, data + have);
            idat = true;
            if (cut) break;
        } else if (is_type(type, "IEND")) {
            iend = true;
            break;
        } // every other chunk is ignored
    }
    if (!have_header) PNG_FAIL("MissingHeader");
    if (f->idat.empty()) PNG_FAIL("MissingImageData");
    if (!iend) f->truncated = true;
    const ScanLayout L = scan_layout(f->header); // scanDataLength (:787): checked arithmetic, then the limit
    if (L.overflow) PNG_FAIL("ImageTooLarge");
    if (over(lim.max_decompressed_bytes, L.total)) PNG_FAIL("ImageTooLarge");
    return ZG_OK;
}

// The pixel type png.toNativeImage picks (:852-1146). Interlaced grey + alpha without tRNS comes back as Image(u8)
// (the Adam7 branch, :862-870, looks only at tRNS), non-interlaced grey + alpha as Rgba.
int native_pixel(const PngFile &f) {
    const bool t = f.trns_len >= 0;
    switch (f.header.color_type) {
    case 0: return t ? ZG_PIXEL_RGBA_U8 : ZG_PIXEL_U8;
    case 4: return (t || f.header.interlace_method != 1) ? ZG_PIXEL_RGBA_U8 : ZG_PIXEL_U8;
    case 6: return ZG_PIXEL_RGBA_U8;
    default: return t ? ZG_PIXEL_RGBA_U8 : ZG_PIXEL_RGB_U8; // rgb, palette
    }
}

// ---- inflate + de-filter (png.toNativeImage :801-852; :1442-1533, :1721-1803) ----------------------------------------------
int inflate_scan(const PngFile &f, const ScanLayout &L, ScanBytes *scan, bool *truncated) {
    scan->resize(L.total + 1); // not cleared; one spare byte: output past the expected size is error.ImageTooLarge (:829-833)
    z_stream zs{};
    if (inflateInit(&zs) != Z_OK) { set_error("inflateInit failed"); return ZG_ERR_OUT_OF_MEMORY; }
    // zlib counts in 32-bit quantities: both sides are fed in slices. All the input and all the room are on offer, so the
    // loop ends on the end of the stream, an error, no room left (too much data) or nothing left to read (a cut stream).
    size_t in_at = 0, out_given = 0;
    const size_t out_total = L.total + 1, slice = (size_t)1 << 30;
    int zr = Z_OK;
    zs.next_out = scan->data();
    for (;;) {
        if (zs.avail_in == 0 && in_at < f.idat.size()) {
            const size_t take = f.idat.size() - in_at < slice ? f.idat.size() - in_at : slice;
            zs.next_in = const_cast<Bytef *>(f.idat.data() + in_at);
            zs.avail_in = (uInt)take;
            in_at += take;
        }
        if (zs.avail_out == 0 && out_given < out_total) {
            const size_t take = out_total - out_given < slice ? out_total - out_given : slice;
            zs.avail_out = (uInt)take;
            out_given += take;
        }
        if (zs.avail_out == 0) break;
        const bool last_input = in_at >= f.idat.size();
        zr = inflate(&zs, last_input ? Z_SYNC_FLUSH : Z_NO_FLUSH);
        if (zr != Z_OK) break;
        if (last_input && zs.avail_in == 0 && zs.avail_out != 0) break;
    }
    const size_t produced = (size_t)(zs.next_out - scan->data());
    inflateEnd(&zs);
    // The reference reads exactly the expected size and then asks for one more byte (:829-842): a byte arriving is
    // ImageTooLarge whatever comes after it, so that test goes first; zlib, given all the input at once, may already have run
    // into a later error (a checksum that no longer matches, say) in the same call.
    if (produced > L.total) PNG_FAIL("ImageTooLarge");
    if (zr == Z_DATA_ERROR || zr == Z_NEED_DICT || zr == Z_MEM_ERROR || zr == Z_STREAM_ERROR) PNG_FAIL("ReadFailed"); // corruption, not truncation
    if (produced < L.total) { // a short or cut stream: keep whole rows, the rest decodes as zero pixels (:846-851)
        *truncated = true;
        const size_t keep = complete_prefix(produced, L);
        memset(scan->data() + keep, 0, L.total - keep);
    }
    return ZG_OK;
}

inline int paeth_predict(int a, int b, int c) { // png.zig:1442-1448
    const int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c);
    return (pb < pa || pc < pa) ? (pc < pb ? c : b) : a;
}
template <int BPP> void defilter_rows(uint8_t *block, size_t row_bytes, uint32_t rows, bool *bad_filter) {
    const uint8_t *up = nullptr;
    for (uint32_t y = 0; y < rows; ++y) {
        uint8_t *row = block + (size_t)y * (row_bytes + 1);
        const int filter = row[0];
        uint8_t *cur = row + 1;
        const size_t n = row_bytes;
        if (filter > 4) { *bad_filter = true; return; }
        switch (filter) {
        case 1:
            for (size_t i = BPP; i < n; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - BPP]);
            break;
        case 2:
            if (up) for (size_t i = 0; i < n; ++i) cur[i] = (uint8_t)(cur[i] + up[i]);
            break;
        case 3:
            if (up) {
                for (size_t i = 0; i < BPP && i < n; ++i) cur[i] = (uint8_t)(cur[i] + (up[i] >> 1));
                for (size_t i = BPP; i < n; ++i) cur[i] = (uint8_t)(cur[i] + ((cur[i - BPP] + up[i]) >> 1));
            } else {
                for (size_t i = BPP; i < n; ++i) cur[i] = (uint8_t)(cur[i] + (cur[i - BPP] >> 1));
            }
            break;
        case 4:
            if (up) {
                for (size_t i = 0; i < BPP && i < n; ++i) cur[i] = (uint8_t)(cur[i] + up[i]);
                for (size_t i = BPP; i < n; ++i) cur[i] = (uint8_t)(cur[i] + paeth_predict(cur[i - BPP], up[i], up[i - BPP]));
            } else { // first row: Paeth(left, 0, 0) = left
                for (size_t i = BPP; i < n; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - BPP]);
            }
            break;
        default: break;
        }
        up = cur;
    }
}
int defilter_scan(ScanBytes *scan, const zg_png_header &h, const ScanLayout &L) {
    const int bpp = (png_channels(h.color_type) * h.bit_depth + 7) / 8;
    bool bad = false;
    for (int p = 0; p < L.npass && !bad; ++p) {
        const PassGeom &g = L.pass[p];
        if (g.w == 0) continue;
        uint8_t *block = scan->data() + g.offset;
        switch (bpp) {
        case 1: defilter_rows<1>(block, g.row_bytes, g.h, &bad); break;
        case 2: defilter_rows<2>(block, g.row_bytes, g.h, &bad); break;
        case 3: defilter_rows<3>(block, g.row_bytes, g.h, &bad); break;
        case 4: defilter_rows<4>(block, g.row_bytes, g.h, &bad); break;
        case 6: defilter_rows<6>(block, g.row_bytes, g.h, &bad); break;
        default: defilter_rows<8>(block, g.row_bytes, g.h, &bad); break;
        }
    }
    if (bad) PNG_FAIL("InvalidFilterType");
    return ZG_OK;
}
// A palette index past PLTE is error.InvalidPaletteIndex on the non-interlaced path (:1080, :1119); the Adam7 path falls
// back to black instead (:2038-2045) and never looks. Only the `width` real pixels of a row count, not the padding bits.
int check_palette_indices(const ScanBytes &scan, const PngFile &f, const ScanLayout &L) {
    const PassGeom &g = L.pass[0];
    const int depth = f.header.bit_depth, per = 8 / depth, mask = (1 << depth) - 1;
    if ((1 << depth) <= f.palette_len) return ZG_OK;
    for (uint32_t y = 0; y < g.h; ++y) {
        const uint8_t *row = scan.data() + (size_t)y * (g.row_bytes + 1) + 1;
        for (uint32_t x = 0; x < g.w; ++x) {
            const int v = depth == 8 ? row[x] : (row[x / per] >> ((per - 1 - (int)(x % per)) * depth)) & mask;
            if (v >= f.palette_len) PNG_FAIL("InvalidPaletteIndex");
        }
    }
    return ZG_OK;
}

// ---- device: scan data -> pixels -------------------------------------------------------------------------------------------
struct UnpackArgs {
    PassGeom pass[7];
    int interlaced;
    int bit_depth, color_type;
    int palette_len, trns_len; // trns_len < 0: no tRNS
    uint8_t trns[256];
    uint8_t palette[768];
};
// One lane per output pixel. NATIVE is the pixel type png.toNativeImage produces for this file.
template <int NATIVE> __global__ __launch_bounds__(256) void k_png_unpack(const uint8_t *scan, UnpackArgs a, DImg dst) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dst.cols) return;
    int p = 0;
    uint32_t px = (uint32_t)x, py = (uint32_t)y;
    if (a.interlaced) { // the pass a pixel belongs to follows from its position in the 8 x 8 Adam7 cell
        const int xm = x & 7, ym = y & 7;
        p = (ym & 1) ? 6 : (xm & 1) ? 5 : (ym & 2) ? 4 : (xm & 2) ? 3 : (ym & 4) ? 2 : (xm & 4) ? 1 : 0;
        px = ((uint32_t)x - a.pass[p].x0) / a.pass[p].dx;
        py = ((uint32_t)y - a.pass[p].y0) / a.pass[p].dy;
    }
    const uint8_t *row = scan + a.pass[p].offset + (size_t)py * (a.pass[p].row_bytes + 1) + 1;
    const int depth = a.bit_depth, ct = a.color_type;
    const int cs = depth == 16 ? 2 : 1; // a 16-bit sample keeps its high byte, which comes first (readInt(.big) >> 8)
    uint8_t r, g, b, alpha = 255;
    if (ct == 0 || ct == 4) { // extractGrayscalePixel :1855-1909
        uint8_t v;
        if (depth >= 8) {
            const uint8_t *s = row + (size_t)px * cs * (ct == 4 ? 2 : 1);
            v = s[0];
            if (ct == 4) alpha = s[cs];
        } else {
            const int per = 8 / depth, mask = (1 << depth) - 1;
            v = (uint8_t)(((row[px / per] >> ((per - 1 - (int)(px % per)) * depth)) & mask) * (255 / mask));
        }
        // the key is compared with the SCALED 8-bit value: its low byte below 16 bits, its high byte at 16 (:1886-1897)
        if (ct == 0 && a.trns_len >= 2 && v == (depth == 16 ? a.trns[0] : a.trns[1])) alpha = 0;
        r = g = b = v;
    } else if (ct == 3) { // extractPalettePixel :2006-2053 (an index past PLTE was rejected on the host unless interlaced)
        int idx;
        if (depth == 8) idx = row[px];
        else {
            const int per = 8 / depth, mask = (1 << depth) - 1;
            idx = (row[px / per] >> ((per - 1 - (int)(px % per)) * depth)) & mask;
        }
        if (idx < a.palette_len) {
            r = a.palette[idx * 3]; g = a.palette[idx * 3 + 1]; b = a.palette[idx * 3 + 2];
            if (idx < a.trns_len) alpha = a.trns[idx];
        } else {
            r = g = b = 0;
        }
    } else { // extractRgbPixel / extractRgbaPixel :1911-2004
        const uint8_t *s = row + (size_t)px * cs * (ct == 6 ? 4 : 3);
        r = s[0]; g = s[cs]; b = s[2 * cs];
        if (ct == 6) alpha = s[3 * cs];
        else if (a.trns_len >= 6) {
            const int k = depth == 16 ? 0 : 1;
            if (r == a.trns[k] && g == a.trns[2 + k] && b == a.trns[4 + k]) alpha = 0;
        }
    }
    uint8_t *out = (uint8_t *)dst.data + ((size_t)y * dst.stride + x) * Px<NATIVE>::BYTES;
    if constexpr (NATIVE == ZG_PIXEL_U8) out[0] = r;
    else if constexpr (NATIVE == ZG_PIXEL_RGB_U8) { out[0] = r; out[1] = g; out[2] = b; }
    else *(uint32_t *)out = (uint32_t)r | (uint32_t)g << 8 | (uint32_t)b << 16 | (uint32_t)alpha << 24;
}

// 8-bit grey / rgb / rgba without tRNS, not interlaced: the de-filtered rows ARE the pixels; only the filter byte in front of
// every row has to go. A byte-stream copy, 16 bytes per lane; the source rows start at odd addresses (row * (n + 1) + 1), so
// they are read with unaligned dword loads, the destination is written with the widest aligned store its layout allows.
struct __attribute__((packed)) U32U { uint32_t v; };
__device__ inline uint32_t load_u32_unaligned(const uint8_t *p) { return ((const U32U *)p)->v; }
__device__ inline void store_u32_unaligned(uint8_t *p, uint32_t v) { ((U32U *)p)->v = v; }
template <int ALIGN> __global__ __launch_bounds__(256) void k_png_copy_rows(const uint8_t *scan, size_t row_bytes, uint8_t *dst, size_t dst_pitch) {
    const size_t off = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (off >= row_bytes) return;
    const uint8_t *src = scan + (size_t)blockIdx.y * (row_bytes + 1) + 1 + off;
    uint8_t *out = dst + (size_t)blockIdx.y * dst_pitch + off;
    if (off + 16 <= row_bytes) {
        const uint32_t a = load_u32_unaligned(src), b = load_u32_unaligned(src + 4), c = load_u32_unaligned(src + 8), d = load_u32_unaligned(src + 12);
        if constexpr (ALIGN == 16) *(uint4 *)out = make_uint4(a, b, c, d);
        else if constexpr (ALIGN == 4) { ((uint32_t *)out)[0] = a; ((uint32_t *)out)[1] = b; ((uint32_t *)out)[2] = c; ((uint32_t *)out)[3] = d; }
        else { store_u32_unaligned(out, a); store_u32_unaligned(out + 4, b); store_u32_unaligned(out + 8, c); store_u32_unaligned(out + 12, d); }
    } else {
        for (size_t i = 0; off + i < row_bytes; ++i) out[i] = src[i];
    }
}

int natural_space(int pixel) { return pixel_channels(pixel) == 1 ? ZG_CS_GRAY : (pixel_channels(pixel) == 3 ? ZG_CS_RGB : ZG_CS_RGBA); }

int decode_impl(const uint8_t *png, size_t len, const zg_png_limits *limits, const zg_image *dst, int dst_space, int *truncated_out, hipStream_t s) {
    ZG_REQUIRE(png != nullptr, ZG_ERR_INVALID_ARGUMENT, "png: null data");
    zg_png_limits lim;
    if (limits) lim = *limits; else zg_png_default_limits(&lim);
    int rc;
    if ((rc = check_image(dst, "dst"))) return rc;
    PngFile f;
    if ((rc = read_chunks(png, len, lim, &f))) return rc;
    ZG_REQUIRE(dst->rows == f.header.height && dst->cols == f.header.width, ZG_ERR_DIMENSION_MISMATCH, "png: the file is %ux%u, dst is %ux%u",
               f.header.height, f.header.width, dst->rows, dst->cols);
    const ScanLayout L = scan_layout(f.header);
    ScanLease lease;
    ScanBytes &scan = lease.bytes;
    bool truncated = f.truncated;
    if ((rc = inflate_scan(f, L, &scan, &truncated))) return rc;
    if ((rc = defilter_scan(&scan, f.header, L))) return rc;
    if (f.header.color_type == 3 && f.header.interlace_method != 1 && (rc = check_palette_indices(scan, f, L))) return rc;
    if (truncated_out) *truncated_out = truncated ? 1 : 0;

    const int native = native_pixel(f);
    const bool direct = dst->pixel == native && dst_space == natural_space(native);
    const size_t native_bytes = direct ? 0 : (size_t)f.header.width * f.header.height * pixel_size(native);
    uint8_t *dev = nullptr;
    if ((rc = scratch_alloc((void **)&dev, L.total + 64 + native_bytes, s))) return rc;
    // the scan data is pageable host memory that dies with this call: a synchronised copy into scratch
    if ((rc = upload_pageable(dev, scan.data(), L.total, s))) { scratch_free(dev, s); return rc; }

    UnpackArgs a{};
    for (int p = 0; p < 7; ++p) a.pass[p] = L.pass[p < L.npass ? p : 0];
    a.interlaced = f.header.interlace_method == 1;
    a.bit_depth = f.header.bit_depth;
    a.color_type = f.header.color_type;
    a.palette_len = f.palette_len < 0 ? 0 : f.palette_len;
    a.trns_len = f.trns_len;
    if (f.trns_len > 0) memcpy(a.trns, f.trns, (size_t)f.trns_len);
    if (f.palette_len > 0) memcpy(a.palette, f.palette, (size_t)f.palette_len * 3);

    zg_image native_img{dev + (L.total + 63) / 64 * 64, f.header.width, f.header.height, f.header.width, native};
    const zg_image *target = direct ? dst : &native_img;
    const dim3 grid(ceil_div(f.header.width, 256), f.header.height);
    const bool plain_rows = !a.interlaced && a.bit_depth == 8 && a.trns_len < 0 && (a.color_type == 0 || a.color_type == 2 || a.color_type == 6);
    if (plain_rows) {
        const size_t row_bytes = L.pass[0].row_bytes, pitch = target->stride * pixel_size(native);
        const dim3 cgrid(ceil_div((unsigned)((row_bytes + 15) / 16), 256), f.header.height);
        const uintptr_t base = (uintptr_t)target->data;
        if (base % 16 == 0 && pitch % 16 == 0) hipLaunchKernelGGL((k_png_copy_rows<16>), cgrid, dim3(256), 0, s, (const uint8_t *)dev, row_bytes, (uint8_t *)target->data, pitch);
        else if (base % 4 == 0 && pitch % 4 == 0) hipLaunchKernelGGL((k_png_copy_rows<4>), cgrid, dim3(256), 0, s, (const uint8_t *)dev, row_bytes, (uint8_t *)target->data, pitch);
        else hipLaunchKernelGGL((k_png_copy_rows<1>), cgrid, dim3(256), 0, s, (const uint8_t *)dev, row_bytes, (uint8_t *)target->data, pitch);
    } else switch (native) {
    case ZG_PIXEL_U8: hipLaunchKernelGGL((k_png_unpack<ZG_PIXEL_U8>), grid, dim3(256), 0, s, (const uint8_t *)dev, a, dimg(target)); break;
    case ZG_PIXEL_RGB_U8: hipLaunchKernelGGL((k_png_unpack<ZG_PIXEL_RGB_U8>), grid, dim3(256), 0, s, (const uint8_t *)dev, a, dimg(target)); break;
    default: hipLaunchKernelGGL((k_png_unpack<ZG_PIXEL_RGBA_U8>), grid, dim3(256), 0, s, (const uint8_t *)dev, a, dimg(target)); break;
    }
    rc = hipGetLastError() == hipSuccess ? ZG_OK : ZG_ERR_HIP;
    if (rc == ZG_OK && !direct) rc = zg_convert(&native_img, natural_space(native), dst, dst_space, nullptr, (zg_stream)s); // Image.convert (:1160-1184)
    scratch_free(dev, s);
    return rc;
}

// ---- device: pixels -> filtered scan data (png.zig:1535-1719) ----------------------------------------------------------------
__device__ inline int d_paeth(int a, int b, int c) {
    const int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c);
    return (pb < pa || pc < pa) ? (pc < pb ? c : b) : a;
}
__device__ inline int d_predict(int filter, int left, int above, int ul, bool first_row, bool first_px) {
    switch (filter) {
    case 1: return left;
    case 2: return above;
    case 3: return (left + above) >> 1;
    case 4: return first_row ? left : (first_px ? above : d_paeth(left, above, ul)); // :1585-1606
    default: return 0;
    }
}
__device__ inline uint32_t abs_i8(int residual) { const int v = (int8_t)(uint8_t)residual; return (uint32_t)(v < 0 ? -v : v); } // calculateFilterCost :1621-1631

// Four consecutive bytes of the row byte stream per lane (left / upper-left come from BPP bytes earlier, read as unaligned
// dwords; the first dword of a row, which has no complete left neighbour, and a ragged tail go byte by byte).
template <int BPP> struct RowQuad {
    uint32_t cur, left, above, ul;
    int count;        // valid bytes (4, fewer in the tail)
    bool first_row;
    __device__ RowQuad(const uint8_t *cur_row, const uint8_t *up_row, int i, int n) {
        first_row = up_row == nullptr;
        count = n - i < 4 ? n - i : 4;
        if (count == 4 && i >= BPP) {
            cur = load_u32_unaligned(cur_row + i);
            left = load_u32_unaligned(cur_row + i - BPP);
            above = up_row ? load_u32_unaligned(up_row + i) : 0u;
            ul = up_row ? load_u32_unaligned(up_row + i - BPP) : 0u;
        } else {
            cur = left = above = ul = 0;
            for (int k = 0; k < count; ++k) {
                const int j = i + k;
                cur |= (uint32_t)cur_row[j] << (8 * k);
                if (j >= BPP) left |= (uint32_t)cur_row[j - BPP] << (8 * k);
                if (up_row) {
                    above |= (uint32_t)up_row[j] << (8 * k);
                    if (j >= BPP) ul |= (uint32_t)up_row[j - BPP] << (8 * k);
                }
            }
        }
    }
    __device__ int residual(int filter, int k, int i) const {
        const int sh = 8 * k;
        return (int)((cur >> sh) & 0xff) - d_predict(filter, (left >> sh) & 0xff, (above >> sh) & 0xff, (ul >> sh) & 0xff, first_row, i + k < BPP);
    }
};

// One workgroup per row: the five filter costs (sum of |signed residual|) in one pass over the row and the one above, then
// selectBestFilter's pick (:1634-1658): the cheapest, ties to the lowest ordinal; a first row has no up / average / Paeth
// candidates.
template <int BPP> __global__ __launch_bounds__(256) void k_png_row_costs(DImg src, uint8_t *best) {
    const int y = blockIdx.x, n = src.cols * BPP;
    const uint8_t *cur = (const uint8_t *)src.data + (size_t)y * src.stride * BPP;
    const uint8_t *up = y ? cur - src.stride * BPP : nullptr;
    uint32_t c[5] = {0, 0, 0, 0, 0};
    for (int i = threadIdx.x * 4; i < n; i += 1024) {
        const RowQuad<BPP> q(cur, up, i, n);
        for (int k = 0; k < q.count; ++k) {
#pragma unroll
            for (int f = 0; f < 5; ++f) c[f] += abs_i8(q.residual(f, k, i));
        }
    }
    __shared__ uint32_t part[5][4];
#pragma unroll
    for (int f = 0; f < 5; ++f) {
        uint32_t v = c[f];
        for (int off = 32; off; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & 63) == 0) part[f][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t lowest = 0xffffffffu;
        int pick = 0;
        for (int f = 0; f < (y == 0 ? 2 : 5); ++f) {
            const uint32_t total = part[f][0] + part[f][1] + part[f][2] + part[f][3];
            if (total < lowest) { lowest = total; pick = f; } // strict: ties keep the lower ordinal (:1650-1654)
        }
        best[y] = (uint8_t)pick;
    }
}
// filterScanlinesAdaptive's row loop (:1675-1716): rows are analysed every `sample_rate` rows, while the choice is still
// changing (streak == 0) and in the first / last three rows; otherwise the last choice is reused. Sequential by nature and
// tiny. One wave walks it: each lane fetches four rows' picks (one coalesced load per 256 rows), the chain itself runs on
// wave-uniform values (readlane in, a lane-select out), i.e. on the scalar unit with no memory access inside it.
__global__ __launch_bounds__(64) void k_png_select(const uint8_t *best, int rows, uint8_t *filters) {
    const int lane = threadIdx.x;
    const int sample_rate = rows > 512 ? 8 : 1, mask = sample_rate - 1;
    int last = 0, streak = 0;
    for (int base = 0; base < rows; base += 256) {
        const int r0 = base + lane * 4;
        uint32_t w = 0;
        if (r0 + 4 <= rows) w = load_u32_unaligned(best + r0);
        else for (int k = 0; r0 + k < rows; ++k) w |= (uint32_t)best[r0 + k] << (8 * k);
        uint32_t mine = 0;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const uint32_t wj = __builtin_amdgcn_readlane(w, j);
            uint32_t r = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { // rows past the end (last block only) run on zeros; their results are never stored
                const int y = base + j * 4 + k, pick = (int)((wj >> (8 * k)) & 0xff);
                const bool analyze = (y & mask) == 0 || streak == 0 || y < 3 || y >= rows - 3;
                const int grown = streak + 1 < sample_rate ? streak + 1 : sample_rate;
                if (analyze) {
                    streak = pick == last ? grown : 0;
                    last = pick;
                }
                r |= (uint32_t)last << (8 * k);
            }
            mine = lane == j ? r : mine;
        }
        if (r0 + 4 <= rows) store_u32_unaligned(filters + r0, mine);
        else for (int k = 0; r0 + k < rows; ++k) filters[r0 + k] = (uint8_t)(mine >> (8 * k));
    }
}
// filterRow (:1535-1618) for every byte of every row; `filters` == nullptr applies `fixed` to all rows.
template <int BPP> __global__ __launch_bounds__(256) void k_png_filter_rows(DImg src, const uint8_t *filters, int fixed, uint8_t *out) {
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y, n = src.cols * BPP;
    if (i >= n) return;
    const uint8_t *cur = (const uint8_t *)src.data + (size_t)y * src.stride * BPP;
    const uint8_t *up = y ? cur - src.stride * BPP : nullptr;
    const int f = filters ? filters[y] : fixed;
    uint8_t *row = out + (size_t)y * (n + 1);
    if (i == 0) row[0] = (uint8_t)f;
    const RowQuad<BPP> q(cur, up, i, n);
    uint32_t packed = 0;
    for (int k = 0; k < q.count; ++k) packed |= (uint32_t)(uint8_t)q.residual(f, k, i) << (8 * k);
    if (q.count == 4) store_u32_unaligned(row + 1 + i, packed);
    else for (int k = 0; k < q.count; ++k) row[1 + i + k] = (uint8_t)(packed >> (8 * k));
}

template <int BPP> int filter_launch(const zg_image *src, int filter, uint8_t *filtered, uint8_t *best, uint8_t *choice, hipStream_t s) {
    const unsigned n = src->cols * BPP;
    if (filter == ZG_PNG_FILTER_ADAPTIVE) {
        hipLaunchKernelGGL((k_png_row_costs<BPP>), dim3(src->rows), dim3(256), 0, s, dimg(src), best);
        hipLaunchKernelGGL(k_png_select, dim3(1), dim3(64), 0, s, (const uint8_t *)best, (int)src->rows, choice);
    }
    hipLaunchKernelGGL((k_png_filter_rows<BPP>), dim3(ceil_div(ceil_div(n, 4), 256), src->rows), dim3(256), 0, s, dimg(src),
                       filter == ZG_PNG_FILTER_ADAPTIVE ? (const uint8_t *)choice : nullptr, filter, filtered);
    return hipGetLastError() == hipSuccess ? ZG_OK : ZG_ERR_HIP;
}
int filter_impl(const zg_image *src, int filter, uint8_t *filtered, hipStream_t s) {
    int rc;
    if ((rc = check_image(src, "src"))) return rc;
    ZG_REQUIRE(src->pixel == ZG_PIXEL_U8 || src->pixel == ZG_PIXEL_RGB_U8 || src->pixel == ZG_PIXEL_RGBA_U8, ZG_ERR_UNSUPPORTED,
               "png filter: 8-bit grey / rgb / rgba scanlines only (pixel %d)", src->pixel);
    ZG_REQUIRE(filter >= ZG_PNG_FILTER_ADAPTIVE && filter <= 4, ZG_ERR_INVALID_ARGUMENT, "png filter: unknown filter %d", filter);
    ZG_REQUIRE(filtered != nullptr, ZG_ERR_INVALID_ARGUMENT, "png filter: null output");
    if (src->rows == 0 || src->cols == 0) return ZG_OK;
    uint8_t *work = nullptr; // per row: selectBestFilter's pick, then the filter the state machine settles on
    if (filter == ZG_PNG_FILTER_ADAPTIVE && (rc = scratch_alloc((void **)&work, (size_t)src->rows * 2, s))) return rc;
    uint8_t *best = work, *choice = work + src->rows;
    switch (src->pixel) {
    case ZG_PIXEL_U8: rc = filter_launch<1>(src, filter, filtered, best, choice, s); break;
    case ZG_PIXEL_RGB_U8: rc = filter_launch<3>(src, filter, filtered, best, choice, s); break;
    default: rc = filter_launch<4>(src, filter, filtered, best, choice, s); break;
    }
    if (work) scratch_free(work, s);
    return rc;
}

// ---- the IDAT stream (png.zig:1297-1306 + std.compress.flate) -----------------------------------------------------------
// Level 5 / Z_FILTERED is zlib's own "filtered" configuration (good 8, lazy 16, nice 32), the one the reference's preset
// names. Compressed bytes differ between deflaters; what they decode to does not — which is also what lets a large image be
// deflated on several cores: the scanlines are cut into 1 MiB pieces, every piece is a raw deflate stream primed with the
// 32 KiB before it (so matches still reach back across the cut) and flushed to a byte boundary, and the pieces are
// concatenated under one zlib header with the Adler-32 of the whole. Any inflater reads it as one stream.
int deflate_piece(const uint8_t *in, size_t n, const uint8_t *dict, size_t dict_len, int level, int window_bits, bool last, std::vector<uint8_t> *out) {
    z_stream zs{};
    if (deflateInit2(&zs, level, Z_DEFLATED, window_bits, 8, Z_FILTERED) != Z_OK) { set_error("deflateInit2 failed"); return ZG_ERR_OUT_OF_MEMORY; }
    if (dict_len && deflateSetDictionary(&zs, dict, (uInt)dict_len) != Z_OK) { deflateEnd(&zs); set_error("deflateSetDictionary failed"); return ZG_ERR_HIP; }
    out->resize(n + n / 1000 + 1024);
    size_t in_at = 0, out_at = 0;
    for (;;) {
        if (zs.avail_in == 0 && in_at < n) {
            const size_t take = n - in_at < (1u << 30) ? n - in_at : (1u << 30);
            zs.next_in = const_cast<uint8_t *>(in) + in_at;
            zs.avail_in = (uInt)take;
            in_at += take;
        }
        if (out_at + (1u << 16) > out->size()) out->resize(out->size() * 2);
        const size_t room = out->size() - out_at < (1u << 30) ? out->size() - out_at : (1u << 30);
        zs.next_out = out->data() + out_at;
        zs.avail_out = (uInt)room;
        const int zr = deflate(&zs, in_at < n ? Z_NO_FLUSH : (last ? Z_FINISH : Z_SYNC_FLUSH));
        out_at += room - zs.avail_out;
        if (zr == Z_STREAM_END) break;
        if (zr != Z_OK && zr != Z_BUF_ERROR) { deflateEnd(&zs); set_error("deflate failed (%d)", zr); return ZG_ERR_HIP; }
        if (!last && in_at >= n && zs.avail_in == 0 && zs.avail_out != 0) break; // the sync flush is complete
    }
    deflateEnd(&zs);
    out->resize(out_at);
    return ZG_OK;
}
int deflate_scanlines(const uint8_t *scan, size_t scan_bytes, int compression_level, std::vector<uint8_t> *z) {
    const int level = compression_level < 0 ? 5 : (compression_level > 9 ? 9 : compression_level);
    const size_t piece = (size_t)1 << 20, pieces = (scan_bytes + piece - 1) / piece;
    const int threads = (int)std::min<size_t>((size_t)host_threads(), pieces);
    if (threads <= 1 || pieces < 4) return deflate_piece(scan, scan_bytes, nullptr, 0, level, 15, true, z);

    std::vector<std::vector<uint8_t>> part(pieces);
    std::vector<uLong> sum(pieces);
    std::atomic<size_t> next{0};
    std::atomic<int> status{ZG_OK};
    char message[512] = "";
    std::mutex message_mu;
    auto work = [&]() {
        for (size_t i = next.fetch_add(1); i < pieces && status.load() == ZG_OK; i = next.fetch_add(1)) {
            const size_t at = i * piece, n = std::min(piece, scan_bytes - at), back = std::min<size_t>(at, 32768);
            int rc;
            try {
                rc = deflate_piece(scan + at, n, scan + at - back, back, level, -15, i + 1 == pieces, &part[i]);
            } catch (const std::bad_alloc &) {
                set_error("out of host memory");
                rc = ZG_ERR_OUT_OF_MEMORY;
            }
            sum[i] = adler32(adler32(0L, Z_NULL, 0), scan + at, (uInt)n);
            if (rc != ZG_OK) { // the error text is per thread: carry it to the caller's
                std::lock_guard<std::mutex> lock(message_mu);
                if (status.exchange(rc) == ZG_OK) snprintf(message, sizeof message, "%s", zg_last_error());
            }
        }
    };
    {
        std::vector<std::thread> crew;
        crew.reserve((size_t)threads - 1);
        try {
            for (int t = 1; t < threads; ++t) crew.emplace_back(work);
        } catch (const std::system_error &) { // no more threads to be had: the ones there are share the pieces
        }
        work();
        for (std::thread &t : crew) t.join();
    }
    if (status.load() != ZG_OK) { set_error("%s", message); return status.load(); }

    size_t total = 2 + 4;
    for (const auto &v : part) total += v.size();
    z->clear();
    z->reserve(total);
    const unsigned cmf = 0x78, flevel = level < 2 ? 0 : (level < 6 ? 1 : (level == 6 ? 2 : 3));
    unsigned flg = flevel << 6;
    flg += 31 - (cmf * 256 + flg) % 31;
    z->push_back((uint8_t)cmf);
    z->push_back((uint8_t)flg);
    uLong adler = adler32(0L, Z_NULL, 0);
    for (size_t i = 0; i < pieces; ++i) {
        z->insert(z->end(), part[i].begin(), part[i].end());
        adler = adler32_combine(adler, sum[i], (z_off_t)std::min(piece, scan_bytes - i * piece));
    }
    uint8_t tail[4];
    store_be32(tail, (uint32_t)adler);
    z->insert(z->end(), tail, tail + 4);
    return ZG_OK;
}

// ---- the container writer (encodeRaw, png.zig:1335-1398) ------------------------------------------------------------------
void append_chunk(std::vector<uint8_t> *out, const char *type, const uint8_t *data, size_t n) {
    const size_t at = out->size();
    out->resize(at + 12 + n);
    uint8_t *p = out->data() + at;
    store_be32(p, (uint32_t)n);
    memcpy(p + 4, type, 4);
    if (n) memcpy(p + 8, data, n);
    store_be32(p + 8 + n, chunk_crc(p + 4, n + 4));
}

int encode_impl(const zg_image *src, int src_space, const zg_png_encode_options *options, uint8_t **out, size_t *out_len, hipStream_t s) {
    ZG_REQUIRE(out && out_len, ZG_ERR_INVALID_ARGUMENT, "png encode: null output");
    *out = nullptr;
    *out_len = 0;
    zg_png_encode_options opt;
    if (options) opt = *options; else zg_png_default_encode_options(&opt);
    int rc;
    if ((rc = check_image(src, "src"))) return rc;
    ZG_REQUIRE(src->rows > 0 && src->cols > 0, ZG_ERR_INVALID_ARGUMENT, "png encode: empty image");
    ZG_REQUIRE(opt.filter >= ZG_PNG_FILTER_ADAPTIVE && opt.filter <= 4, ZG_ERR_INVALID_ARGUMENT, "png encode: unknown filter %d", opt.filter);
    ZG_REQUIRE(opt.srgb_intent <= 3, ZG_ERR_INVALID_ARGUMENT, "png encode: sRGB intent %d", opt.srgb_intent);
    const bool direct = (src->pixel == ZG_PIXEL_U8 && src_space == ZG_CS_GRAY) || (src->pixel == ZG_PIXEL_RGB_U8 && src_space == ZG_CS_RGB) ||
                        (src->pixel == ZG_PIXEL_RGBA_U8 && src_space == ZG_CS_RGBA);
    const int enc_pixel = direct ? src->pixel : ZG_PIXEL_RGB_U8; // any other T is converted to Rgb first (:1409-1423)
    const size_t row_bytes = (size_t)src->cols * pixel_size(enc_pixel), scan_bytes = (row_bytes + 1) * src->rows;
    const size_t rgb_bytes = direct ? 0 : (row_bytes * src->rows + 63) / 64 * 64;
    uint8_t *dev = nullptr;
    if ((rc = scratch_alloc((void **)&dev, rgb_bytes + scan_bytes, s))) return rc;
    zg_image rgb{dev, src->cols, src->rows, src->cols, ZG_PIXEL_RGB_U8};
    if (!direct) rc = zg_convert(src, src_space, &rgb, ZG_CS_RGB, nullptr, (zg_stream)s);
    if (rc == ZG_OK) rc = filter_impl(direct ? src : &rgb, opt.filter, dev + rgb_bytes, s);
    ScanLease lease;
    ScanBytes &scan = lease.bytes;
    if (rc == ZG_OK) {
        scan.resize(scan_bytes);
        rc = download_pageable(scan.data(), dev + rgb_bytes, scan_bytes, s);
    }
    scratch_free(dev, s);
    if (rc) return rc;

    std::vector<uint8_t> z;
    if ((rc = deflate_scanlines(scan.data(), scan_bytes, opt.compression_level, &z))) return rc;
    const size_t out_at = z.size();

    std::vector<uint8_t> file(kSignature, kSignature + 8);
    uint8_t ihdr[13] = {0};
    store_be32(ihdr, src->cols);
    store_be32(ihdr + 4, src->rows);
    ihdr[8] = 8;
    ihdr[9] = enc_pixel == ZG_PIXEL_U8 ? 0 : (enc_pixel == ZG_PIXEL_RGB_U8 ? 2 : 6);
    append_chunk(&file, "IHDR", ihdr, 13);
    if (opt.srgb_intent >= 0) { // sRGB wins over gAMA (:1357-1370)
        const uint8_t intent = (uint8_t)opt.srgb_intent;
        append_chunk(&file, "sRGB", &intent, 1);
    } else if (opt.has_gamma) {
        uint8_t g[4];
        store_be32(g, (uint32_t)(opt.gamma * 100000.0f)); // @trunc(g * 100000.0) in f32
        append_chunk(&file, "gAMA", g, 4);
    }
    append_chunk(&file, "IDAT", z.data(), out_at);
    append_chunk(&file, "IEND", nullptr, 0);
    uint8_t *mem = (uint8_t *)malloc(file.size());
    if (!mem) { set_error("png encode: out of host memory"); return ZG_ERR_OUT_OF_MEMORY; }
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

void zg_png_default_limits(zg_png_limits *l) { // png.zig:16-41
    const size_t max_file = (size_t)100 * 1024 * 1024;
    l->max_png_bytes = l->max_chunk_bytes = l->max_idat_bytes = max_file;
    l->max_chunks = 8192;
    l->max_width = l->max_height = 8192;
    l->max_pixels = 67108864ull;
    l->max_decompressed_bytes = 536886272u;
}
void zg_png_default_encode_options(zg_png_encode_options *o) {
    o->filter = ZG_PNG_FILTER_ADAPTIVE;
    o->compression_level = -1;
    o->has_gamma = 0;
    o->gamma = 0.0f;
    o->srgb_intent = -1;
}

// png.getInfo (:308-410) is a forward-only reader: no CRC checks, stops at the first IDAT / IEND, running out of bytes
// between chunks ends the scan quietly, running out inside a chunk's type or payload is error.EndOfStream.
int zg_png_info(const uint8_t *png, size_t len, const zg_png_limits *limits, zg_png_header *out) {
    ZG_REQUIRE(png && out, ZG_ERR_INVALID_ARGUMENT, "png info: null argument");
    zg_png_limits lim;
    if (limits) lim = *limits; else zg_png_default_limits(&lim);
    if (len < 8) PNG_FAIL("EndOfStream");
    if (memcmp(png, kSignature, 8) != 0) PNG_FAIL("InvalidPngSignature");
    size_t at = 8, count = 0;
    zg_png_header h{};
    bool found = false;
    auto skip = [&](uint64_t want) { at += (size_t)((uint64_t)(len - at) < want ? len - at : want); };
    for (;;) {
        if (over(lim.max_png_bytes, at)) PNG_FAIL("PngDataTooLarge");
        if (len - at < 4) break;
        const uint32_t length = load_be32(png + at);
        at += 4;
        if (len - at < 4) PNG_FAIL("EndOfStream");
        const uint8_t *type = png + at;
        at += 4;
        if (over(lim.max_chunks, ++count)) PNG_FAIL("TooManyChunks");
        if (lim.max_png_bytes != 0 && at + (size_t)length + 4 > lim.max_png_bytes) PNG_FAIL("PngDataTooLarge");
        if (is_type(type, "IDAT") || is_type(type, "IEND")) break;
        if (is_type(type, "IHDR")) {
            if (found) PNG_FAIL("MultipleHeaders");
            if (length != 13) PNG_FAIL("InvalidHeaderLength");
            if (len - at < 13) PNG_FAIL("EndOfStream");
            const uint8_t *d = png + at;
            h = zg_png_header{};
            h.width = load_be32(d);
            h.height = load_be32(d + 4);
            if (h.width == 0 || h.height == 0) PNG_FAIL("InvalidDimensions");
            if (d[9] != 0 && d[9] != 2 && d[9] != 3 && d[9] != 4 && d[9] != 6) PNG_FAIL("InvalidColorType");
            h.bit_depth = d[8]; h.color_type = d[9]; h.compression_method = d[10]; h.filter_method = d[11]; h.interlace_method = d[12];
            found = true;
            at += 13;
            skip(4);
        } else if (is_type(type, "gAMA") && found) {
            if (length != 4) PNG_FAIL("InvalidGammaLength");
            if (len - at < 4) PNG_FAIL("EndOfStream");
            h.has_gamma = 1;
            h.gamma = (float)load_be32(png + at) / 100000.0f;
            at += 4;
            skip(4);
        } else if (is_type(type, "sRGB") && found) {
            if (length != 1) PNG_FAIL("InvalidSrgbLength");
            if (len - at < 1) PNG_FAIL("EndOfStream");
            if (png[at] > 3) PNG_FAIL("InvalidSrgbIntent");
            h.has_srgb = 1;
            h.srgb_intent = png[at];
            at += 1;
            skip(4);
        } else {
            skip((uint64_t)length + 4);
        }
    }
    if (!found) PNG_FAIL("MissingHeader");
    *out = h;
    return ZG_OK;
}

int zg_png_probe(const uint8_t *png, size_t len, const zg_png_limits *limits, zg_png_header *header_out, int *native_pixel_out, int *truncated_out) {
    ZG_REQUIRE(png != nullptr, ZG_ERR_INVALID_ARGUMENT, "png probe: null data");
    zg_png_limits lim;
    if (limits) lim = *limits; else zg_png_default_limits(&lim);
    return no_throw([&]() -> int {
        PngFile f;
        const int rc = read_chunks(png, len, lim, &f);
        if (rc) return rc;
        if (header_out) *header_out = f.header;
        if (native_pixel_out) *native_pixel_out = native_pixel(f);
        if (truncated_out) *truncated_out = f.truncated ? 1 : 0;
        return ZG_OK;
    });
}

int zg_png_scan_hash(const uint8_t *png, size_t len, const zg_png_limits *limits, uint64_t *hash_out, int *truncated_out) {
    ZG_REQUIRE(png && hash_out, ZG_ERR_INVALID_ARGUMENT, "png scan hash: null argument");
    zg_png_limits lim;
    if (limits) lim = *limits; else zg_png_default_limits(&lim);
    return no_throw([&]() -> int {
        PngFile f;
        int rc;
        if ((rc = read_chunks(png, len, lim, &f))) return rc;
        const ScanLayout L = scan_layout(f.header);
        ScanLease lease;
        ScanBytes &scan = lease.bytes;
        bool truncated = f.truncated;
        if ((rc = inflate_scan(f, L, &scan, &truncated))) return rc;
        if ((rc = defilter_scan(&scan, f.header, L))) return rc;
        if (f.header.color_type == 3 && f.header.interlace_method != 1 && (rc = check_palette_indices(scan, f, L))) return rc;
        uint64_t h = 1469598103934665603ull; // FNV-1a over the de-filtered scan data, filter bytes included
        for (size_t i = 0; i < L.total; ++i) { h ^= scan[i]; h *= 1099511628211ull; }
        *hash_out = h;
        if (truncated_out) *truncated_out = truncated ? 1 : 0;
        return ZG_OK;
    });
}
int zg_png_decode(const uint8_t *png, size_t len, const zg_png_limits *limits, const zg_image *dst, int dst_space, int *truncated_out, zg_stream stream) {
    return no_throw([&] { return decode_impl(png, len, limits, dst, dst_space, truncated_out, as_stream(stream)); });
}
int zg_png_decode_host(const uint8_t *png, size_t len, const zg_png_limits *limits, const zg_image *dst, int dst_space, int *truncated_out) {
    HostStage d;
    int rc;
    if ((rc = d.upload(dst, false, true))) return rc;
    if ((rc = no_throw([&] { return decode_impl(png, len, limits, &d.dev, dst_space, truncated_out, nullptr); }))) return rc;
    ZG_HIP(hipStreamSynchronize(nullptr));
    return d.finish();
}
int zg_png_filter(const zg_image *src, int filter, uint8_t *filtered, zg_stream stream) { return filter_impl(src, filter, filtered, as_stream(stream)); }
int zg_png_encode(const zg_image *src, int src_space, const zg_png_encode_options *options, uint8_t **out, size_t *out_len, zg_stream stream) {
    return no_throw([&] { return encode_impl(src, src_space, options, out, out_len, as_stream(stream)); });
}
int zg_png_encode_host(const zg_image *src, int src_space, const zg_png_encode_options *options, uint8_t **out, size_t *out_len) {
    HostStage a;
    int rc;
    if ((rc = a.upload(src, true, false))) return rc;
    return no_throw([&] { return encode_impl(&a.dev, src_space, options, out, out_len, nullptr); });
}
int zg_png_compress(const uint8_t *scanlines, size_t len, int compression_level, uint8_t **out, size_t *out_len) {
    ZG_REQUIRE(out && out_len, ZG_ERR_INVALID_ARGUMENT, "png compress: null output");
    *out = nullptr;
    *out_len = 0;
    ZG_REQUIRE(scanlines != nullptr || len == 0, ZG_ERR_INVALID_ARGUMENT, "png compress: null input");
    return no_throw([&]() -> int {
        std::vector<uint8_t> z;
        const uint8_t none = 0;
        const int rc = deflate_scanlines(len ? scanlines : &none, len, compression_level, &z);
        if (rc) return rc;
        uint8_t *mem = (uint8_t *)malloc(z.size());
        if (!mem) { set_error("png compress: out of host memory"); return ZG_ERR_OUT_OF_MEMORY; }
        memcpy(mem, z.data(), z.size());
        *out = mem;
        *out_len = z.size();
        return ZG_OK;
    });
}
void zg_png_free(void *p) { free(p); }

} // extern "C"
