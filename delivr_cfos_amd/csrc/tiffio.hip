// tiffio.hip - z-plane ingest: TIFF planes -> uint16 volume in HBM.
//
// The reference reads its raw stacks plane by plane with cv2.imread / skimage.io / tifffile
// (downsample/downsample_and_mask.py:25-30, :36-41, :396-404) - codecs from third-party libraries, one plane at a
// time on one core.  Once inference takes seconds, that dominates the wall clock.  Here: a TIFF reader (classic and
// BigTIFF, little/big endian, strips or tiles, 8/16-bit single channel, compression 1 = none, 5 = LZW, 8 / 32946 =
// deflate through zlib, each with or without the horizontal predictor - what light-sheet stitchers and the reference's
// own writers emit, blob_highlighter.py:131, and what its libtiff-based readers accept) decodes planes on a pool of host
// threads straight into pinned staging buffers, from which they are copied to their z offset in the device volume while
// the next planes are being decoded.
// Decoder pinned by fixtures written with libtiff or decoded by it when they were written by hand (tiles, compressed
// BigTIFF): tests/golden/tiff_*.tif, oracle/make_goldens.py, tests/test_host_cpu.py.
#include "common.h"

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct TiffInfo {
    bool big = false;      // big-endian ("MM")
    bool big_tiff = false; // BigTIFF (magic 43: 8-byte offsets, 20-byte IFD entries)
    bool tiled = false;    // TileWidth / TileLength / TileOffsets / TileByteCounts instead of strips
    uint32_t width = 0, height = 0, bits = 1, compression = 1, samples = 1, rows_per_strip = 0xffffffffu, predictor = 1,
             photometric = 1, planar = 1, sample_format = 1, tile_w = 0, tile_h = 0;
    std::vector<uint64_t> strip_off, strip_len;  // strips, or tiles in row-major tile order
    std::vector<uint64_t> tile_off, tile_len;
};

struct Reader {
    const uint8_t* p;
    size_t n;
    bool big;
    bool ok(size_t off, size_t len) const { return off <= n && len <= n - off; }
    uint16_t u16(size_t o) const { return big ? (uint16_t)(p[o] << 8 | p[o + 1]) : (uint16_t)(p[o] | p[o + 1] << 8); }
    uint32_t u32(size_t o) const {
        return big ? ((uint32_t)p[o] << 24 | (uint32_t)p[o + 1] << 16 | (uint32_t)p[o + 2] << 8 | p[o + 3])
                   : ((uint32_t)p[o] | (uint32_t)p[o + 1] << 8 | (uint32_t)p[o + 2] << 16 | (uint32_t)p[o + 3] << 24);
    }
    uint64_t u64(size_t o) const { return big ? ((uint64_t)u32(o) << 32 | u32(o + 4)) : ((uint64_t)u32(o + 4) << 32 | u32(o)); }
};

// values of an IFD entry (types BYTE 1, SHORT 3, LONG 4, and in BigTIFF LONG8 16) as u64.  Classic entries are 12 bytes
// (count u32, 4 bytes of value / offset), BigTIFF entries 20 bytes (count u64, 8 bytes of value / offset)
bool entry_values(const Reader& r, size_t e, bool big_tiff, std::vector<uint64_t>& out) {
    const uint16_t type = r.u16(e + 2);
    const uint64_t count = big_tiff ? r.u64(e + 4) : r.u32(e + 4);
    const size_t sz = type == 1 ? 1 : type == 3 ? 2 : type == 4 ? 4 : (type == 16 && big_tiff) ? 8 : 0;
    if (!sz || count > (1u << 28)) return false;
    const size_t inline_bytes = big_tiff ? 8 : 4;
    size_t off = e + (big_tiff ? 12 : 8);
    if ((size_t)count * sz > inline_bytes) {
        off = big_tiff ? (size_t)r.u64(off) : r.u32(off);
        if (!r.ok(off, (size_t)count * sz)) return false;
    }
    out.resize(count);
    for (uint64_t i = 0; i < count; ++i)
        out[i] = sz == 1 ? r.p[off + i] : sz == 2 ? r.u16(off + 2 * i) : sz == 4 ? r.u32(off + 4 * i) : r.u64(off + 8 * i);
    return true;
}

const char* parse_ifd(const uint8_t* data, size_t n, TiffInfo& t) {
    if (n < 8) return "file shorter than a TIFF header";
    if (data[0] == 'I' && data[1] == 'I') t.big = false;
    else if (data[0] == 'M' && data[1] == 'M') t.big = true;
    else return "not a TIFF file";
    Reader r{data, n, t.big};
    const uint16_t magic = r.u16(2);
    if (magic != 42 && magic != 43) return "not a TIFF file (magic is neither 42 nor 43)";
    t.big_tiff = magic == 43;
    size_t ifd, ne, esz, e0;
    if (t.big_tiff) {
        if (n < 16 || r.u16(4) != 8 || r.u16(6) != 0) return "BigTIFF header with an offset size other than 8";
        ifd = (size_t)r.u64(8);
        if (!r.ok(ifd, 8)) return "IFD offset outside the file";
        const uint64_t cnt = r.u64(ifd);
        if (cnt > 65535) return "implausible IFD entry count";
        ne = (size_t)cnt;
        esz = 20;
        e0 = ifd + 8;
    } else {
        ifd = r.u32(4);
        if (!r.ok(ifd, 2)) return "IFD offset outside the file";
        ne = r.u16(ifd);
        esz = 12;
        e0 = ifd + 2;
    }
    if (!r.ok(e0, ne * esz)) return "IFD truncated";
    std::vector<uint64_t> v;
    for (size_t i = 0; i < ne; ++i) {
        const size_t e = e0 + i * esz;
        const uint16_t tag = r.u16(e);
        if (tag != 256 && tag != 257 && tag != 258 && tag != 259 && tag != 262 && tag != 273 && tag != 277 && tag != 278 &&
            tag != 279 && tag != 284 && tag != 317 && tag != 322 && tag != 323 && tag != 324 && tag != 325 && tag != 339)
            continue;
        if (!entry_values(r, e, t.big_tiff, v) || v.empty()) return "unsupported IFD entry type";
        switch (tag) {
            case 256: t.width = (uint32_t)v[0]; break;
            case 257: t.height = (uint32_t)v[0]; break;
            case 258: t.bits = (uint32_t)v[0]; if (v.size() != 1) return "multi-sample planes are not supported"; break;
            case 259: t.compression = (uint32_t)v[0]; break;
            case 262: t.photometric = (uint32_t)v[0]; break;
            case 273: t.strip_off = v; break;
            case 277: t.samples = (uint32_t)v[0]; break;
            case 278: t.rows_per_strip = (uint32_t)v[0]; break;
            case 279: t.strip_len = v; break;
            case 284: t.planar = (uint32_t)v[0]; break;
            case 317: t.predictor = (uint32_t)v[0]; break;
            case 322: t.tile_w = (uint32_t)v[0]; break;
            case 323: t.tile_h = (uint32_t)v[0]; break;
            case 324: t.tile_off = v; break;
            case 325: t.tile_len = v; break;
            case 339: t.sample_format = (uint32_t)v[0]; break;
        }
    }
    if (!t.width || !t.height) return "image size missing";
    if (t.width > (1u << 20) || t.height > (1u << 20) || (uint64_t)t.width * t.height > (1ull << 34)) return "implausible image size";
    if (t.samples != 1) return "only single-channel planes are supported";
    if (t.bits != 8 && t.bits != 16) return "only 8- and 16-bit samples are supported";
    if (t.sample_format != 1) return "only unsigned integer samples are supported";
    if (t.compression != 1 && t.compression != 5 && t.compression != 8 && t.compression != 32946)
        return "only uncompressed, LZW and deflate planes are supported";
    if (t.predictor != 1 && t.predictor != 2) return "unsupported predictor";
    if (!t.tile_off.empty() || t.tile_w || t.tile_h) {  // tiled: the tile tables take the place of the strip tables
        if (!t.tile_w || !t.tile_h || t.tile_w > (1u << 16) || t.tile_h > (1u << 16)) return "tile size missing or implausible";
        if (t.tile_off.empty() || t.tile_off.size() != t.tile_len.size()) return "tile tables missing";
        t.tiled = true;
        t.strip_off = t.tile_off;
        t.strip_len = t.tile_len;
    }
    if (t.strip_off.empty() || t.strip_off.size() != t.strip_len.size()) return "strip tables missing";
    if (!t.tiled && t.rows_per_strip == 0) return "RowsPerStrip is 0";
    for (size_t i = 0; i < t.strip_off.size(); ++i)
        if (!r.ok((size_t)t.strip_off[i], (size_t)t.strip_len[i])) return "strip or tile outside the file";
    return nullptr;
}

// TIFF 6.0 LZW: MSB-first codes of 9..12 bits, ClearCode 256, EndOfInformation 257, the code width grows one code
// early ("early change").  Every table string is a substring of the output written so far - entry `next` is the
// previous string plus the byte that follows it in the output - so the table holds (offset, length) pairs and a code
// is decoded with one forward copy inside the output buffer.  Returns the number of bytes written (<= cap).
// Camera noise compresses to strings of ~1.4 bytes, so what is paid is the work per CODE: a string of up to 8 bytes is copied as
// ONE unaligned 8-byte load + store whatever its length (what lands behind it is overwritten by the next code; `dst` must have
// LZW_SLACK writable bytes behind `cap`), the bit buffer is refilled four bytes at a time, an entry's offset and length share a
// word.  A 2048 x 2048 uint16 plane of dense synthetic tissue on one core of the build container: 100 -> 58 ms (uniform random
// 16-bit values 44 ms, brain with background 24 ms).  Tried and slower: the 256 literals as table entries pointing into a prefix
// of the buffer, which makes literal-or-string no branch but every literal an 8-byte store - the overlapping wide stores defeat
// store-to-load forwarding for the strings that follow (random values 44 -> 77 ms).
constexpr size_t LZW_SLACK = 16;
size_t lzw_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
    uint64_t ent[4096];  // offset << 16 | length (a string is at most 4096 - 258 + 1 bytes long)
    int width = 9, next = 258;
    long long prev_pos = -1;  // where the previous code's string starts in dst
    uint32_t prev_len = 0;
    uint64_t acc = 0;
    int nbits = 0;
    size_t ip = 0, op = 0;
    for (;;) {
        if (nbits <= 32) {
            if (ip + 4 <= n) {
                uint32_t w;
                memcpy(&w, src + ip, 4);
                acc = (acc << 32) | __builtin_bswap32(w);
                ip += 4;
                nbits += 32;
            } else {
                while (nbits <= 56 && ip < n) {
                    acc = (acc << 8) | src[ip++];
                    nbits += 8;
                }
            }
        }
        if (nbits < width) break;
        const int code = (int)((acc >> (nbits - width)) & ((1u << width) - 1));
        nbits -= width;
        const size_t start = op;
        uint32_t l;
        if (code < 256) {
            if (op >= cap) break;
            dst[op++] = (uint8_t)code;
            l = 1;
        } else if (code >= 258 && code < next) {
            const uint64_t e = ent[code];
            l = (uint32_t)(e & 0xffffu);
            const size_t o = (size_t)(e >> 16);
            if (op + l > cap) break;
            const uint8_t* from = dst + o;
            uint8_t* to = dst + op;
            // (the string ends at or before op: its l bytes are final; what an 8-byte load reads behind them is not, and lands
            // behind the string in the output, where the next code writes)
            if (l <= 8) {
                uint64_t v;
                memcpy(&v, from, 8);
                memcpy(to, &v, 8);
            } else if (o + l + 8 <= op) {  // far enough back for 8-byte steps that never read what this copy writes
                for (uint32_t k = 0; k < l; k += 8) {
                    uint64_t v;
                    memcpy(&v, from + k, 8);
                    memcpy(to + k, &v, 8);
                }
            } else {
                for (uint32_t k = 0; k < l; ++k) to[k] = from[k];
            }
            op += l;
        } else if (code == 256) {
            width = 9;
            next = 258;
            prev_pos = -1;
            continue;
        } else if (code == 257) {
            break;
        } else if (code == next && prev_pos >= 0) {  // KwKwK: previous string + its own first byte (the source ends where the copy starts)
            l = prev_len + 1;
            if (op + l > cap) break;
            const uint8_t* from = dst + prev_pos;
            uint8_t* to = dst + op;
            for (uint32_t k = 0; k < l; ++k) to[k] = from[k];
            op += l;
        } else {
            break;  // corrupt stream
        }
        if (prev_pos >= 0 && next < 4096) {
            ent[next] = ((uint64_t)prev_pos << 16) | (prev_len + 1);
            ++next;
        }
        prev_pos = (long long)start;
        prev_len = l;
        if (next >= (1 << width) - 1 && width < 12) ++width;  // early change
    }
    return op < cap ? op : cap;
}

// decodes one plane into out (row-major uint16, 8-bit samples are widened); returns nullptr or an error text.
// Segments are strips (full image width, RowsPerStrip rows, the last one short) or tiles (TileWidth x TileLength, always
// stored whole, row-major tile order); every segment is decompressed on its own, the horizontal predictor runs along the
// rows of the segment, and the part of it that lies inside the image is copied out.
const char* decode_plane(const uint8_t* data, size_t n, const TiffInfo& t, uint16_t* out, std::vector<uint8_t>& scratch) {
    const size_t bps = t.bits / 8;
    const uint32_t seg_w = t.tiled ? t.tile_w : t.width;
    const uint32_t seg_h = t.tiled ? t.tile_h : (t.rows_per_strip < t.height ? t.rows_per_strip : t.height);
    const size_t nx = ((size_t)t.width + seg_w - 1) / seg_w, ny = ((size_t)t.height + seg_h - 1) / seg_h;
    if (t.strip_off.size() < nx * ny) return t.tiled ? "fewer tiles than the image needs" : "fewer strips than the image needs";
    const size_t seg_row_bytes = (size_t)seg_w * bps;
    scratch.resize((size_t)seg_h * seg_row_bytes + LZW_SLACK);  // (the LZW decoder's 8-byte copies run past the end of a string)
    for (size_t sy = 0; sy < ny; ++sy)
        for (size_t sx = 0; sx < nx; ++sx) {
            const size_t s = sy * nx + sx;
            const uint32_t r0 = (uint32_t)(sy * seg_h), c0 = (uint32_t)(sx * seg_w);
            const uint32_t rows_in = r0 + seg_h <= t.height ? seg_h : t.height - r0;  // rows of the segment inside the image
            const uint32_t cols_in = c0 + seg_w <= t.width ? seg_w : t.width - c0;
            const uint32_t rows_stored = t.tiled ? seg_h : rows_in;                    // tiles are stored whole
            const size_t want = (size_t)rows_stored * seg_row_bytes;
            const uint8_t* src = data + t.strip_off[s];
            const uint8_t* raw;
            if (t.compression == 1) {
                if (t.strip_len[s] < want) return "uncompressed segment shorter than its rows";
                raw = src;
            } else if (t.compression == 5) {
                if (lzw_decode(src, (size_t)t.strip_len[s], scratch.data(), want) != want) return "LZW segment does not decode to its rows";
                raw = scratch.data();
            } else {  // 8 / 32946: zlib stream
                uLongf got = (uLongf)want;
                if (uncompress(scratch.data(), &got, src, (uLong)t.strip_len[s]) != Z_OK || (size_t)got != want)
                    return "deflate segment does not decode to its rows";
                raw = scratch.data();
            }
            for (uint32_t r = 0; r < rows_in; ++r) {
                const uint8_t* in = raw + (size_t)r * seg_row_bytes;
                uint16_t* o = out + (size_t)(r0 + r) * t.width + c0;
                if (bps == 1) {
                    uint8_t acc = 0;
                    for (uint32_t x = 0; x < cols_in; ++x) {
                        const uint8_t v = t.predictor == 2 ? (uint8_t)(acc + in[x]) : in[x];
                        acc = v;
                        o[x] = v;
                    }
                } else if (!t.big && t.predictor == 1) {
                    memcpy(o, in, (size_t)cols_in * 2);  // little-endian samples are already in host order
                } else {
                    uint16_t acc = 0;
                    for (uint32_t x = 0; x < cols_in; ++x) {
                        const uint16_t w = t.big ? (uint16_t)(in[2 * x] << 8 | in[2 * x + 1]) : (uint16_t)(in[2 * x] | in[2 * x + 1] << 8);
                        const uint16_t v = t.predictor == 2 ? (uint16_t)(acc + w) : w;
                        acc = v;
                        o[x] = v;
                    }
                }
            }
        }
    if (t.photometric == 0) {  // WhiteIsZero
        const uint16_t top = t.bits == 8 ? 255 : 65535;
        for (size_t i = 0; i < (size_t)t.width * t.height; ++i) out[i] = (uint16_t)(top - out[i]);
    }
    return nullptr;
}

// TIFF 6.0 LZW encoder (the counterpart of lzw_decode; same conventions as libtiff's writer: every strip starts with
// ClearCode, the code width grows when the entry that fills the current width has been added - "early change" - the
// table is reset with a ClearCode when entry 4093 has been added, EndOfInformation closes the strip).
void lzw_encode(const uint8_t* src, size_t n, std::vector<uint8_t>& out) {
    constexpr int HSIZE = 16384;  // open addressing, keys (prefix << 8 | byte) + 1
    static thread_local std::vector<uint32_t> keys;
    static thread_local std::vector<uint16_t> vals;
    keys.assign(HSIZE, 0);
    vals.resize(HSIZE);
    uint64_t acc = 0;
    int nbits = 0, width = 9, next = 258;
    auto put = [&](int code) {
        acc = (acc << width) | (uint32_t)code;
        nbits += width;
        while (nbits >= 8) {
            out.push_back((uint8_t)(acc >> (nbits - 8)));
            nbits -= 8;
        }
    };
    put(256);
    if (n == 0) {
        put(257);
        if (nbits) out.push_back((uint8_t)(acc << (8 - nbits)));
        return;
    }
    int prefix = src[0];
    for (size_t i = 1; i < n; ++i) {
        const uint8_t c = src[i];
        const uint32_t key = ((uint32_t)prefix << 8 | c) + 1;
        uint32_t hsh = (key * 2654435761u) >> 18;  // 14 bits
        int found = -1;
        while (keys[hsh]) {
            if (keys[hsh] == key) {
                found = vals[hsh];
                break;
            }
            hsh = (hsh + 1) & (HSIZE - 1);
        }
        if (found >= 0) {
            prefix = found;
            continue;
        }
        put(prefix);
        keys[hsh] = key;
        vals[hsh] = (uint16_t)next;
        ++next;
        if (next == 4094) {  // table full: ClearCode at the current (12-bit) width, start over
            put(256);
            keys.assign(HSIZE, 0);
            width = 9;
            next = 258;
        } else if (next > (1 << width) - 1) {
            ++width;
        }
        prefix = c;
    }
    put(prefix);
    ++next;
    if (next == 4094) {
        put(256);
        width = 9;
    } else if (next > (1 << width) - 1) {
        ++width;
    }
    put(257);
    if (nbits) out.push_back((uint8_t)(acc << (8 - nbits)));
}

bool read_file(const char* path, std::vector<uint8_t>& buf) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz < 0) {
        fclose(f);
        return false;
    }
    buf.resize((size_t)sz);
    const size_t got = sz ? fread(buf.data(), 1, (size_t)sz, f) : 0;
    fclose(f);
    return got == (size_t)sz;
}

thread_local std::string g_tiff_error;

}  // namespace

extern "C" {

const char* dlv_tiff_last_error(void) { return g_tiff_error.c_str(); }

int dlv_tiff_plane_size(const char* path, int* height, int* width, int* bits) {
    try {
        if (!path) return DLV_EINVAL;
        std::vector<uint8_t> buf;
        if (!read_file(path, buf)) {
            g_tiff_error = std::string("cannot read ") + path;
            return DLV_EINVAL;
        }
        TiffInfo t;
        if (const char* e = parse_ifd(buf.data(), buf.size(), t)) {
            g_tiff_error = std::string(path) + ": " + e;
            return DLV_EUNSUP;
        }
        if (height) *height = (int)t.height;
        if (width) *width = (int)t.width;
        if (bits) *bits = (int)t.bits;
        return DLV_OK;
    
    } catch (const std::exception& e) {  // nothing throws across the C ABI
        g_tiff_error = std::string("dlv_tiff_plane_size: ") + e.what();
        return DLV_EUNSUP;
    } catch (...) {
        g_tiff_error = "dlv_tiff_plane_size: unknown exception";
        return DLV_EUNSUP;
    }
}

int dlv_tiff_read_plane_u16(const char* path, uint16_t* out_host, int height, int width) {
    try {
        if (!path || !out_host) return DLV_EINVAL;
        std::vector<uint8_t> buf, scratch;
        if (!read_file(path, buf)) {
            g_tiff_error = std::string("cannot read ") + path;
            return DLV_EINVAL;
        }
        TiffInfo t;
        const char* e = parse_ifd(buf.data(), buf.size(), t);
        if (!e && ((int)t.height != height || (int)t.width != width)) e = "plane size differs from the expected one";
        if (!e) e = decode_plane(buf.data(), buf.size(), t, out_host, scratch);
        if (e) {
            g_tiff_error = std::string(path) + ": " + e;
            return DLV_EUNSUP;
        }
        return DLV_OK;
    
    } catch (const std::exception& e) {  // nothing throws across the C ABI
        g_tiff_error = std::string("dlv_tiff_read_plane_u16: ") + e.what();
        return DLV_EUNSUP;
    } catch (...) {
        g_tiff_error = "dlv_tiff_read_plane_u16: unknown exception";
        return DLV_EUNSUP;
    }
}

// one 8/16-bit single-channel plane -> little-endian classic TIFF, strips of ~64 KB, compression 1 (none) or 5 (LZW,
// no predictor) - what tifffile.imwrite(..., compression='lzw') produces for the reference's plane files
int dlv_tiff_write_plane(const char* path, const void* data_host, int height, int width, int bits, int compression) {
    try {
        if (!path || !data_host) return DLV_EINVAL;
        if (height <= 0 || width <= 0 || (bits != 8 && bits != 16) || (compression != 1 && compression != 5)) {
            g_tiff_error = "dlv_tiff_write_plane: 8/16-bit planes, compression 1 or 5";
            return DLV_EINVAL;
        }
        const size_t row_bytes = (size_t)width * (bits / 8);
        const uint32_t rps = (uint32_t)std::min<size_t>((size_t)height, std::max<size_t>(1, ((size_t)64 << 10) / row_bytes));
        const uint32_t nstrips = ((uint32_t)height + rps - 1) / rps;
        std::vector<uint8_t> file(8);
        std::vector<uint32_t> offs(nstrips), lens(nstrips);
        const uint8_t* src = (const uint8_t*)data_host;  // host is little-endian, like the file
        std::vector<uint8_t> enc;
        for (uint32_t st = 0; st < nstrips; ++st) {
            const uint32_t r0 = st * rps, rows = std::min(rps, (uint32_t)height - r0);
            const uint8_t* p = src + (size_t)r0 * row_bytes;
            const size_t nb = (size_t)rows * row_bytes;
            offs[st] = (uint32_t)file.size();
            if (compression == 5) {
                enc.clear();
                lzw_encode(p, nb, enc);
                file.insert(file.end(), enc.begin(), enc.end());
                lens[st] = (uint32_t)enc.size();
            } else {
                file.insert(file.end(), p, p + nb);
                lens[st] = (uint32_t)nb;
            }
            if (file.size() & 1) file.push_back(0);
            if (file.size() > 0xfff00000ull) {
                g_tiff_error = "plane too large for classic TIFF";
                return DLV_EUNSUP;
            }
        }
        auto put16 = [&](uint16_t v) { file.push_back((uint8_t)v); file.push_back((uint8_t)(v >> 8)); };
        auto put32 = [&](uint32_t v) { put16((uint16_t)v); put16((uint16_t)(v >> 16)); };
        uint32_t off_tab = 0, len_tab = 0;
        if (nstrips > 1) {
            off_tab = (uint32_t)file.size();
            for (uint32_t v : offs) put32(v);
            len_tab = (uint32_t)file.size();
            for (uint32_t v : lens) put32(v);
        }
        const uint32_t ifd = (uint32_t)file.size();
        struct Tag { uint16_t tag, type; uint32_t count, value; };
        const Tag tags[] = {{256, 4, 1, (uint32_t)width}, {257, 4, 1, (uint32_t)height}, {258, 3, 1, (uint32_t)bits},
                            {259, 3, 1, (uint32_t)compression}, {262, 3, 1, 1}, {273, 4, nstrips, nstrips > 1 ? off_tab : offs[0]},
                            {277, 3, 1, 1}, {278, 4, 1, rps}, {279, 4, nstrips, nstrips > 1 ? len_tab : lens[0]}, {339, 3, 1, 1}};
        put16((uint16_t)(sizeof(tags) / sizeof(tags[0])));
        for (const Tag& t : tags) {
            put16(t.tag);
            put16(t.type);
            put32(t.count);
            if (t.type == 3 && t.count == 1) {
                put16((uint16_t)t.value);
                put16(0);
            } else {
                put32(t.value);
            }
        }
        put32(0);
        file[0] = 'I';
        file[1] = 'I';
        file[2] = 42;
        file[3] = 0;
        file[4] = (uint8_t)ifd;
        file[5] = (uint8_t)(ifd >> 8);
        file[6] = (uint8_t)(ifd >> 16);
        file[7] = (uint8_t)(ifd >> 24);
        FILE* f = fopen(path, "wb");
        if (!f || fwrite(file.data(), 1, file.size(), f) != file.size()) {
            if (f) fclose(f);
            g_tiff_error = std::string("cannot write ") + path;
            return DLV_EINVAL;
        }
        fclose(f);
        return DLV_OK;
    
    } catch (const std::exception& e) {  // nothing throws across the C ABI
        g_tiff_error = std::string("dlv_tiff_write_plane: ") + e.what();
        return DLV_EUNSUP;
    } catch (...) {
        g_tiff_error = "dlv_tiff_write_plane: unknown exception";
        return DLV_EUNSUP;
    }
}

// planes paths[0..n_planes) -> vol_dev[(z0 + i) * plane_stride + y * row_stride + x], decoded by n_threads host threads
// into two pinned staging chunks that alternate between "being filled" and "being copied".  The threads live for the whole
// call and take planes from one queue (their file / scratch buffers keep their capacity: a thread per chunk paid an mmap, the
// page faults and a munmap of ~13 MB for every plane - profiles/README.md, r06final_tiff_threads); a plane of chunk c may be
// decoded once the copy that read the staging buffer of chunk c - 2 has finished.
int dlv_tiff_stack_to_device(dlv_ctx* ctx, const char* const* paths, int n_planes, int height, int width, uint16_t* vol_dev,
                             long long plane_stride, long long row_stride, int n_threads) {
    if (!ctx || !paths || !vol_dev) return DLV_EINVAL;
    if (n_planes <= 0 || height <= 0 || width <= 0) return dlv_fail(ctx, DLV_EINVAL, "empty stack");
    if (row_stride < width || plane_stride < (long long)height * row_stride) return dlv_fail(ctx, DLV_EINVAL, "strides smaller than the plane");
    DLV_HIP(ctx, hipSetDevice(ctx->device));
    // no exception crosses the C ABI: allocation failures of the vectors / strings / threads below are reported as codes
    try {
    if (n_threads <= 0) n_threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
    n_threads = std::min(n_threads, n_planes);
    const size_t plane_elems = (size_t)height * width;
    // planes per staging chunk: at least one per thread, about 256 MB
    const int by_size = (int)std::max<size_t>(1, ((size_t)256 << 20) / (plane_elems * 2));
    const int chunk = ctx->tiff_chunk > 0 ? std::min(n_planes, ctx->tiff_chunk) : std::max(1, std::min(n_planes, std::max(n_threads, by_size)));
    const int n_chunks = (n_planes + chunk - 1) / chunk;
    // staging buffers and events are released on every exit path
    struct Staging {
        uint16_t* buf[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
        ~Staging() {
            for (int b = 0; b < 2; ++b) {
                if (ev[b]) (void)hipEventDestroy(ev[b]);
                if (buf[b]) (void)hipHostFree(buf[b]);
            }
        }
    } stg;
    for (int b = 0; b < std::min(2, n_chunks); ++b) {
        DLV_HIP(ctx, hipHostMalloc((void**)&stg.buf[b], (size_t)chunk * plane_elems * 2, hipHostMallocDefault));
        DLV_HIP(ctx, hipEventCreateWithFlags(&stg.ev[b], hipEventDisableTiming));
    }
    uint16_t** stage = stg.buf;
    hipEvent_t* done = stg.ev;
    std::string err;
    std::atomic<bool> failed{false}, oom{false};
    std::atomic<int> next{0};
    std::mutex mu;
    std::condition_variable cv;
    int writable = std::min(2, n_chunks);          // chunks [0, writable) may be decoded into their staging buffer (guarded by mu)
    std::vector<int> decoded(n_chunks, 0);          // planes of a chunk that are in the staging buffer (guarded by mu)
    std::vector<std::string> errs(n_threads);
    std::vector<std::thread> pool;
    pool.reserve(n_threads);
    struct Joiner {  // every exit path (also a failed thread start: std::system_error) stops and joins what is running
        std::vector<std::thread>& p;
        std::atomic<bool>& stop;
        std::mutex& mu;
        std::condition_variable& cv;
        ~Joiner() {
            bool any = false;
            for (auto& th : p) any = any || th.joinable();
            if (!any) return;
            {
                std::lock_guard<std::mutex> lk(mu);
                stop = true;
            }
            cv.notify_all();
            for (auto& th : p)
                if (th.joinable()) th.join();
        }
    };
    std::atomic<bool> stop{false};
    Joiner joiner{pool, stop, mu, cv};
    for (int t = 0; t < n_threads; ++t)
        pool.emplace_back([&, t]() {
            try {  // an exception escaping a thread function would call std::terminate
            std::vector<uint8_t> buf, scratch;
            for (;;) {
                const int i = next.fetch_add(1);
                if (i >= n_planes || failed || stop) break;
                const int c = i / chunk;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return c < writable || failed || stop; });
                }
                if (failed || stop) break;
                const char* path = paths[i];
                const char* e = nullptr;
                TiffInfo ti;
                if (!read_file(path, buf)) e = "cannot read the file";
                if (!e) e = parse_ifd(buf.data(), buf.size(), ti);
                if (!e && ((int)ti.height != height || (int)ti.width != width)) e = "plane size differs from the first plane";
                if (!e) e = decode_plane(buf.data(), buf.size(), ti, stage[c & 1] + (size_t)(i - c * chunk) * plane_elems, scratch);
                if (e) errs[t] = std::string(path) + ": " + e;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (e) failed = true;
                    ++decoded[c];
                }
                cv.notify_all();
            }
            } catch (...) {
                {
                    std::lock_guard<std::mutex> lk(mu);
                    failed = true;  // (no allocation here: the message is set by the caller)
                    oom = true;
                }
                cv.notify_all();
            }
        });
    int rc = DLV_OK;
    for (int c = 0; c < n_chunks && !failed; ++c) {
        const int b = c & 1, c0 = c * chunk;
        const int cn = std::min(chunk, n_planes - c0);
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return decoded[c] == cn || failed; });
        }
        if (failed) break;
        hipError_t he = hipSuccess;
        if (plane_stride == (long long)height * row_stride) {  // planes back to back: one 2-D copy for the chunk
            he = hipMemcpy2DAsync(vol_dev + (long long)c0 * plane_stride, (size_t)row_stride * 2, stage[b], (size_t)width * 2,
                                  (size_t)width * 2, (size_t)cn * height, hipMemcpyHostToDevice, ctx->stream);
        } else {
            for (int i = 0; i < cn && he == hipSuccess; ++i)
                he = hipMemcpy2DAsync(vol_dev + (long long)(c0 + i) * plane_stride, (size_t)row_stride * 2,
                                      stage[b] + (size_t)i * plane_elems, (size_t)width * 2, (size_t)width * 2, (size_t)height,
                                      hipMemcpyHostToDevice, ctx->stream);
        }
        if (he != hipSuccess) {
            err = std::string("hipMemcpy2DAsync: ") + hipGetErrorString(he);
            std::lock_guard<std::mutex> lk(mu);
            failed = true;
            break;
        }
        (void)hipEventRecord(done[b], ctx->stream);
        if (c + 2 < n_chunks) {  // chunk c + 2 shares this staging buffer: writable once the copy has read it (chunk c + 1 decodes meanwhile)
            (void)hipEventSynchronize(done[b]);
            {
                std::lock_guard<std::mutex> lk(mu);
                writable = c + 3;
            }
            cv.notify_all();
        }
    }
    if (failed) cv.notify_all();
    for (auto& th : pool) th.join();
    (void)hipStreamSynchronize(ctx->stream);  // before the staging buffers go away
    if (failed) {
        for (auto& e : errs)
            if (!e.empty()) err = e;
        if (err.empty() && oom) err = "out of host memory while decoding";
        rc = dlv_fail(ctx, oom ? DLV_ENOMEM : DLV_EUNSUP, "%s", err.c_str());
    }
    return rc;
    } catch (const std::bad_alloc&) {
        (void)hipStreamSynchronize(ctx->stream);
        return dlv_fail(ctx, DLV_ENOMEM, "dlv_tiff_stack_to_device: out of host memory");
    } catch (...) {
        (void)hipStreamSynchronize(ctx->stream);
        return dlv_fail(ctx, DLV_EHIP, "dlv_tiff_stack_to_device: host-side failure (thread creation?)");
    }
}

}  // extern "C"
