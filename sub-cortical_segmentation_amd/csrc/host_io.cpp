// libscs_io - host-side file helper of the inference path (include/scs_io.h): values of a gzip-compressed file at given byte
// offsets.  A library of its own, without any HIP dependency: the Python layer reads volumes before it has touched a GPU
// runtime, and loading the HIP runtime of /opt/rocm through libscs.so ahead of torch's bundled one leaves the process with two.
//
// test_scan needs, per subject, 15 prior values for each of the ~3e5 ROI voxels out of a 755 MB (inflated) NIfTI volume
// that an external registration tool wrote as ONE gzip stream (/root/reference/cnn_cort/base.py:388-394 reads it with
// nibabel's get_data()).  Inflating that stream is the largest single item of the host's work per subject (0.6 CPU-seconds
// with the system zlib, 1.2 GB/s: its match copy moves a byte at a time, and a prior volume is almost all long runs).
// This is a small inflate (RFC 1951 / 1952) written for that shape of data: distance-1 matches are memsets, other matches
// word copies, the output goes through a 1 MB window that stays in cache and is never written to memory as a whole, and only
// the requested elements are copied out.  It stops as soon as the last requested element has been produced.
//
// The member's ISIZE is always checked.  Its CRC-32 is checked whenever a member is inflated to its end with the check on:
// scs_gz_read (T1 / mask volumes: whole-file reads) and scs_gz_gather calls that ask for the inflated length (whole stream);
// SCS_GZ_NOCRC=1 switches it off.  A gather that stops at its last requested element (the 755 MB prior: the ROI ends long
// before the stream does) never sees the trailer and takes no CRC - INTEGRATION.md says so.  The CRC of the produced bytes
// is slicing-by-8 with zero runs folded by a precomputed operator (the tables of the writer below).
#include <cerrno>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/scs_io.h"

namespace scs {
static thread_local char g_io_err[512] = "";
static void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_io_err, sizeof(g_io_err), fmt, ap);
    va_end(ap);
}
}  // namespace scs
static const int SCS_OK = 0, SCS_ERR_ARG = -1;

extern "C" const char* scs_io_last_error(void) { return scs::g_io_err; }

namespace {

struct Bits {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc = 0;
    int cnt = 0;
    inline void refill() {
        while (cnt <= 56 && p < end) { acc |= (uint64_t)(*p++) << cnt; cnt += 8; }
    }
    inline uint32_t peek(int n) const { return (uint32_t)(acc & ((1ull << n) - 1)); }
    inline void drop(int n) { acc >>= n; cnt -= n; }
    inline uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
};

struct Entry { uint16_t sym; uint8_t len; uint8_t sub; };      // sub > 0: `sym` is the offset of a second-level table of 2^sub entries

// canonical Huffman decoding table, two levels: P primary bits, longer codes through per-prefix sub-tables
struct Table {
    std::vector<Entry> e;
    int P = 0;
    bool build(const uint8_t* lens, int n, int primary) {
        P = primary;
        int count[16] = {0};
        for (int i = 0; i < n; ++i) count[lens[i]]++;
        count[0] = 0;
        int code = 0, next[16], used = 0;
        long kraft = 0;
        for (int l = 1; l <= 15; ++l) { code = (code + count[l - 1]) << 1; next[l] = code; used += count[l]; kraft += (long)count[l] << (15 - l); }
        if (used == 0) { e.assign((size_t)1 << P, Entry{0, 0, 0}); return true; }
        if (kraft > (1L << 15)) return false;                               // over-subscribed
        if (kraft < (1L << 15) && !(used == 1 && count[1] == 1)) return false;   // incomplete (only a single 1-bit code may be)
        e.assign((size_t)1 << P, Entry{0, 0, 0});
        // pass 1: longest code behind every primary prefix
        std::vector<uint8_t> maxl((size_t)1 << P, 0);
        std::vector<uint32_t> rev(n, 0);
        for (int i = 0; i < n; ++i) {
            const int L = lens[i];
            if (!L) continue;
            uint32_t c = (uint32_t)next[L]++, r = 0;
            for (int b = 0; b < L; ++b) r |= ((c >> b) & 1u) << (L - 1 - b);
            rev[i] = r;
            if (L > P) { uint8_t& m = maxl[r & ((1u << P) - 1)]; if (L > m) m = (uint8_t)L; }
        }
        for (size_t q = 0; q < maxl.size(); ++q)
            if (maxl[q]) {
                const int sb = maxl[q] - P;
                if (e.size() + ((size_t)1 << sb) > 65535) return false;
                e[q] = Entry{(uint16_t)e.size(), 0, (uint8_t)sb};
                e.resize(e.size() + ((size_t)1 << sb), Entry{0, 0, 0});
            }
        for (int i = 0; i < n; ++i) {
            const int L = lens[i];
            if (!L) continue;
            const uint32_t r = rev[i];
            if (L <= P) {
                for (uint32_t k = r; k < (1u << P); k += 1u << L) e[k] = Entry{(uint16_t)i, (uint8_t)L, 0};
            } else {
                const Entry pe = e[r & ((1u << P) - 1)];
                const uint32_t hi = r >> P;
                for (uint32_t k = hi; k < (1u << pe.sub); k += 1u << (L - P)) e[pe.sym + k] = Entry{(uint16_t)i, (uint8_t)L, 0};
            }
        }
        return true;
    }
    inline int decode(Bits& b) const {
        Entry x = e[b.peek(P)];
        if (x.sub) x = e[x.sym + ((b.acc >> P) & ((1u << x.sub) - 1))];
        if (!x.len) return -1;
        b.drop(x.len);
        return x.sym;
    }
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

constexpr size_t kHist = 32768, kChunk = 1 << 20, kSlack = 512;

struct Gather {
    const int64_t* off;     // byte offsets into the inflated stream, ascending
    int64_t n, i = 0;
    int elem;
    uint8_t* out;
    // range mode (scs_gz_read): bytes [range_lo, range_hi) of the stream go to range_out
    uint8_t* range_out = nullptr;
    int64_t range_lo = 0, range_hi = 0, copied = 0;
    bool finished() const { return i >= n && (!range_out || copied >= range_hi); }
};

uint32_t crc_feed(uint32_t reg, const uint8_t* p, size_t n);      // CRC-32 register update (defined with the writer's tables below)

struct Window {
    std::vector<uint8_t> buf;
    size_t w = 0;           // bytes in the window
    int64_t base = 0;       // stream offset of buf[0]
    // CRC-32 of the current member's output (crc_on): register, and the stream offset up to which it has been taken
    bool crc_on = false;
    uint32_t crc_reg = 0xFFFFFFFFu;
    int64_t crc_pos = 0;
    Window() : buf(kHist + kChunk + kSlack) {}
    void crc_catch_up() {
        const int64_t have = base + (int64_t)w;
        if (crc_on && have > crc_pos) { crc_reg = crc_feed(crc_reg, buf.data() + (crc_pos - base), (size_t)(have - crc_pos)); crc_pos = have; }
    }
    // hand the produced bytes to the gather, keep the last kHist bytes as match history
    void flush(Gather& g) {
        const int64_t have = base + (int64_t)w;
        if (g.range_out) {
            const int64_t lo = g.copied > g.range_lo ? g.copied : g.range_lo, hi = have < g.range_hi ? have : g.range_hi;
            if (hi > lo) memcpy(g.range_out + (lo - g.range_lo), buf.data() + (lo - base), (size_t)(hi - lo));
            g.copied = have;
        }
        while (g.i < g.n && g.off[g.i] + g.elem <= have) {
            memcpy(g.out + (size_t)g.i * g.elem, buf.data() + (g.off[g.i] - base), (size_t)g.elem);
            ++g.i;
        }
        crc_catch_up();                                             // before the bytes leave the window
        if (w > kHist) {
            const size_t keep = kHist;
            memmove(buf.data(), buf.data() + (w - keep), keep);
            base += (int64_t)(w - keep);
            w = keep;
        }
    }
};

// one deflate stream; returns 0, or -1 (corrupt / truncated).  *done: every requested element has been delivered.
int inflate_stream(Bits& b, Window& win, Gather& g, bool* done) {
    static const uint8_t kOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    Table lit, dist;
    uint8_t lens[320];
    for (;;) {
        b.refill();
        if (b.cnt < 3) return -1;
        const uint32_t final_block = b.take(1), type = b.take(2);
        if (type == 0) {
            b.drop(b.cnt & 7);
            b.refill();
            if (b.cnt < 32) return -1;
            const uint32_t len = b.take(16), nlen = b.take(16);
            if ((len ^ 0xFFFFu) != nlen) return -1;
            uint32_t left = len;
            while (left && b.cnt >= 8) {                                    // whole bytes still in the accumulator
                if (win.w >= kHist + kChunk) { win.flush(g); if (g.finished()) { *done = true; return 0; } }
                win.buf[win.w++] = (uint8_t)b.take(8); --left;
            }
            while (left) {                                                  // the rest straight from the input (accumulator empty)
                if (win.w >= kHist + kChunk) { win.flush(g); if (g.finished()) { *done = true; return 0; } }
                const size_t room = kHist + kChunk - win.w, avail = (size_t)(b.end - b.p);
                const size_t m = left < room ? left : room;
                if (m > avail) return -1;
                memcpy(win.buf.data() + win.w, b.p, m);
                b.p += m; win.w += m; left -= (uint32_t)m;
            }
        } else if (type == 1 || type == 2) {
            if (type == 1) {
                int k = 0;
                for (; k < 144; ++k) lens[k] = 8;
                for (; k < 256; ++k) lens[k] = 9;
                for (; k < 280; ++k) lens[k] = 7;
                for (; k < 288; ++k) lens[k] = 8;
                if (!lit.build(lens, 288, 10)) return -1;
                for (k = 0; k < 32; ++k) lens[k] = 5;                       // (codes 30, 31 exist and are invalid)
                if (!dist.build(lens, 32, 8)) return -1;
            } else {
                b.refill();
                if (b.cnt < 14) return -1;
                const int hlit = (int)b.take(5) + 257, hdist = (int)b.take(5) + 1, hclen = (int)b.take(4) + 4;
                if (hlit > 286 || hdist > 30) return -1;
                uint8_t cl[19] = {0};
                for (int k = 0; k < hclen; ++k) { b.refill(); if (b.cnt < 3) return -1; cl[kOrder[k]] = (uint8_t)b.take(3); }
                Table ct;
                if (!ct.build(cl, 19, 7)) return -1;
                int k = 0;
                while (k < hlit + hdist) {
                    b.refill();
                    const int s = ct.decode(b);
                    if (s < 0 || b.cnt < 0) return -1;
                    if (s < 16) lens[k++] = (uint8_t)s;
                    else {
                        int rep; uint8_t v = 0;
                        if (s == 16) { if (k == 0) return -1; v = lens[k - 1]; rep = 3 + (int)b.take(2); }
                        else if (s == 17) rep = 3 + (int)b.take(3);
                        else rep = 11 + (int)b.take(7);
                        if (b.cnt < 0 || k + rep > hlit + hdist) return -1;
                        while (rep--) lens[k++] = v;
                    }
                }
                if (lens[256] == 0) return -1;
                if (!lit.build(lens, hlit, 10) || !dist.build(lens + hlit, hdist, 8)) return -1;
            }
            uint8_t* buf = win.buf.data();
            for (;;) {
                if (win.w >= kHist + kChunk) { win.flush(g); if (g.finished()) { *done = true; return 0; } }
                b.refill();
                int s = lit.decode(b);
                if (s < 0 || b.cnt < 0) return -1;                            // (cnt < 0: read past the end of the input)
                if (s < 256) { buf[win.w++] = (uint8_t)s; continue; }
                if (s == 256) break;
                s -= 257;
                if (s >= 29) return -1;
                const int len = kLenBase[s] + (int)b.take(kLenExtra[s]);
                b.refill();
                const int ds = dist.decode(b);
                if (ds < 0 || ds >= 30) return -1;
                const size_t d = kDistBase[ds] + b.take(kDistExtra[ds]);
                if (b.cnt < 0 || d > win.w) return -1;                       // (d > bytes produced so far: corrupt)
                uint8_t* dst = buf + win.w;
                const uint8_t* src = dst - d;
                if (d == 1) memset(dst, *src, (size_t)len);
                else if (d >= (size_t)len) memcpy(dst, src, (size_t)len);
                else if (d >= 8) { for (int k = 0; k < len; k += 8) memcpy(dst + k, src + k, 8); }   // (slack behind the window)
                else { for (int k = 0; k < len; ++k) dst[k] = src[k]; }
                win.w += (size_t)len;
            }
            if (b.cnt < 0) return -1;
        } else {
            return -1;
        }
        if (final_block) return 0;
    }
}

}  // namespace

static int gz_drive(const char* path, Gather& g, int64_t* inflated_bytes, bool want_crc) {
    using scs::set_error;
    FILE* f = fopen(path, "rb");
    if (!f) { set_error("%s: %s", path, strerror(errno)); return SCS_ERR_ARG; }
    std::vector<uint8_t> in;
    {
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (sz < 18) { fclose(f); set_error("%s: too short for a gzip file", path); return SCS_ERR_ARG; }
        in.resize((size_t)sz);
        const size_t got = fread(in.data(), 1, in.size(), f);
        fclose(f);
        if (got != in.size()) { set_error("%s: short read", path); return SCS_ERR_ARG; }
    }
    Window win;
    win.crc_on = want_crc && getenv("SCS_GZ_NOCRC") == nullptr;
    const uint8_t* p = in.data();
    const uint8_t* end = p + in.size();
    bool done = g.finished() && inflated_bytes == nullptr;
    while (!done && p < end) {
        while (p < end && *p == 0) ++p;                                     // zero padding between members is legal
        if (p == end) break;
        if (end - p < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8) { set_error("%s: not a gzip member", path); return SCS_ERR_ARG; }
        const int flg = p[3];
        p += 10;
        if (flg & 4) { if (end - p < 2) goto trunc; const size_t xl = p[0] | (p[1] << 8); p += 2; if ((size_t)(end - p) < xl) goto trunc; p += xl; }
        if (flg & 8) { while (p < end && *p) ++p; if (p == end) goto trunc; ++p; }
        if (flg & 16) { while (p < end && *p) ++p; if (p == end) goto trunc; ++p; }
        if (flg & 2) { if (end - p < 2) goto trunc; p += 2; }
        {
            Bits b{p, end};
            const int64_t before = win.base + (int64_t)win.w;
            win.crc_reg = 0xFFFFFFFFu; win.crc_pos = before;                 // the CRC is per member
            if (inflate_stream(b, win, g, &done) != 0) { set_error("%s: corrupt or truncated deflate stream", path); return SCS_ERR_ARG; }
            if (done) break;
            // unread whole bytes in the accumulator go back to the input
            b.drop(b.cnt & 7);
            p = b.p - b.cnt / 8;
            if (end - p < 8) goto trunc;
            const uint32_t isize = (uint32_t)p[4] | ((uint32_t)p[5] << 8) | ((uint32_t)p[6] << 16) | ((uint32_t)p[7] << 24);
            const int64_t produced = win.base + (int64_t)win.w - before;
            if ((uint32_t)produced != isize) { set_error("%s: gzip member length mismatch", path); return SCS_ERR_ARG; }
            if (win.crc_on) {
                win.crc_catch_up();
                const uint32_t want = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
                if (~win.crc_reg != want) { set_error("%s: gzip member CRC-32 mismatch (corrupted file)", path); return SCS_ERR_ARG; }
            }
            p += 8;
        }
        win.flush(g);
        if (g.finished() && inflated_bytes == nullptr) done = true;
    }
    win.flush(g);
    if (inflated_bytes) *inflated_bytes = win.base + (int64_t)win.w;
    if (g.i < g.n) { set_error("%s: voxel data truncated (%lld of %lld elements)", path, (long long)g.i, (long long)g.n); return SCS_ERR_ARG; }
    if (g.range_out && g.copied < g.range_hi) { set_error("%s: voxel data truncated (%lld of %lld bytes)", path, (long long)g.copied, (long long)g.range_hi); return SCS_ERR_ARG; }
    return SCS_OK;
trunc:
    set_error("%s: truncated gzip stream", path);
    return SCS_ERR_ARG;
}

extern "C" int scs_gz_gather(const char* path, const int64_t* byte_offsets, int64_t n, int elem_bytes, void* out, int64_t* inflated_bytes) {
    using scs::set_error;
    if (!path || (!byte_offsets && n) || (!out && n) || n < 0 || elem_bytes < 1 || elem_bytes > 16) { set_error("bad argument: scs_gz_gather"); return SCS_ERR_ARG; }
    for (int64_t k = 1; k < n; ++k)
        if (byte_offsets[k] < byte_offsets[k - 1]) { set_error("bad argument: offsets must ascend"); return SCS_ERR_ARG; }
    if (n && byte_offsets[0] < 0) { set_error("bad argument: negative offset"); return SCS_ERR_ARG; }
    Gather g{byte_offsets, n, 0, elem_bytes, static_cast<uint8_t*>(out)};
    return gz_drive(path, g, inflated_bytes, inflated_bytes != nullptr);
}

extern "C" int scs_gz_read(const char* path, int64_t offset, int64_t nbytes, void* out) {
    using scs::set_error;
    if (!path || offset < 0 || nbytes < 0 || (!out && nbytes)) { set_error("bad argument: scs_gz_read"); return SCS_ERR_ARG; }
    Gather g{nullptr, 0, 0, 1, nullptr};
    g.range_out = static_cast<uint8_t*>(out); g.range_lo = offset; g.range_hi = offset + nbytes;
    if (nbytes == 0) return SCS_OK;
    return gz_drive(path, g, nullptr, true);
}

// ===================================================================================== a deflate WRITER for sparse volumes
// One complete gzip member for a buffer that is mostly runs: zero words become distance-1 matches of 258 bytes (13 bits each),
// anything that repeats the bytes 8 back (the low mantissa bytes of float64 values that came from float32, a constant fill,
// equal neighbours) becomes a distance-8 match, the rest are literals - all in one fixed-Huffman block (RFC 1951, 3.2.6).
// No hash chains, no lazy matching: ~1 ns per zero-run byte, ~3 ns per data byte; zlib level 1 spends 3 ns on EVERY byte, and
// the float64 [256,256,192,15] probability volume of test_scan is 330 MB of rows of which 39 MB are data.  The CRC-32 of a
// zero run is one multiplication of the register by a precomputed 32x32 bit matrix per set bit of the run length (the
// register update for a zero byte is linear), the CRC of data is slicing-by-8.
namespace scs {

struct CrcTables {
    uint32_t sl[8][256];                 // slicing-by-8
    uint32_t zp[36][4][256];             // zp[k]: register after 2^k zero bytes, as four byte-indexed tables
    CrcTables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            sl[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) sl[t][i] = sl[0][sl[t - 1][i] & 0xFF] ^ (sl[t - 1][i] >> 8);
        uint32_t col[32];                // operator "one zero byte" as images of the 32 basis vectors
        for (int b = 0; b < 32; ++b) { const uint32_t r = 1u << b; col[b] = sl[0][r & 0xFF] ^ (r >> 8); }
        for (int k = 0; k < 36; ++k) {
            for (int j = 0; j < 4; ++j)
                for (uint32_t v = 0; v < 256; ++v) {
                    uint32_t x = 0;
                    for (int i = 0; i < 8; ++i) if (v & (1u << i)) x ^= col[8 * j + i];
                    zp[k][j][v] = x;
                }
            uint32_t sq[32];             // square the operator: image of a basis vector under two applications
            for (int b = 0; b < 32; ++b) {
                const uint32_t r = col[b];
                sq[b] = zp[k][0][r & 0xFF] ^ zp[k][1][(r >> 8) & 0xFF] ^ zp[k][2][(r >> 16) & 0xFF] ^ zp[k][3][r >> 24];
            }
            for (int b = 0; b < 32; ++b) col[b] = sq[b];
        }
    }
    uint32_t zeros(uint32_t r, uint64_t n) const {
        for (int k = 0; n; ++k, n >>= 1)
            if (n & 1) r = zp[k][0][r & 0xFF] ^ zp[k][1][(r >> 8) & 0xFF] ^ zp[k][2][(r >> 16) & 0xFF] ^ zp[k][3][r >> 24];
        return r;
    }
    uint32_t bytes(uint32_t r, const uint8_t* p, size_t n) const {
        while (n >= 8) {
            uint32_t a, b;
            memcpy(&a, p, 4); memcpy(&b, p + 4, 4);
            a ^= r;
            r = sl[7][a & 0xFF] ^ sl[6][(a >> 8) & 0xFF] ^ sl[5][(a >> 16) & 0xFF] ^ sl[4][a >> 24] ^
                sl[3][b & 0xFF] ^ sl[2][(b >> 8) & 0xFF] ^ sl[1][(b >> 16) & 0xFF] ^ sl[0][b >> 24];
            p += 8; n -= 8;
        }
        while (n--) r = sl[0][(r ^ *p++) & 0xFF] ^ (r >> 8);
        return r;
    }
};

struct FixedCodes {
    uint32_t lit[256]; uint8_t lit_n[256];       // literal: code (bit-reversed, ready for an LSB-first writer), bits
    uint32_t len[259]; uint8_t len_n[259];       // match length 3..258: length symbol + its extra bits, total bits
    uint32_t eob; uint8_t eob_n;
    static uint32_t rev(uint32_t c, int n) { uint32_t r = 0; for (int i = 0; i < n; ++i) r |= ((c >> i) & 1u) << (n - 1 - i); return r; }
    static void sym(int s, uint32_t& code, int& n) {
        if (s < 144) { code = 0x30 + s; n = 8; }
        else if (s < 256) { code = 0x190 + (s - 144); n = 9; }
        else if (s < 280) { code = s - 256; n = 7; }
        else { code = 0xC0 + (s - 280); n = 8; }
        code = rev(code, n);
    }
    FixedCodes() {
        for (int b = 0; b < 256; ++b) { uint32_t c; int n; sym(b, c, n); lit[b] = c; lit_n[b] = (uint8_t)n; }
        static const int base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const int extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        for (int L = 3; L <= 258; ++L) {
            int s = 28;
            while (base[s] > L) --s;
            if (L == 258) s = 28;
            uint32_t c; int n;
            sym(257 + s, c, n);
            len[L] = c | ((uint32_t)(L - base[s]) << n);
            len_n[L] = (uint8_t)(n + extra[s]);
        }
        uint32_t c; int n; sym(256, c, n); eob = c; eob_n = (uint8_t)n;
    }
};

static const CrcTables& crc_tables() { static const CrcTables t; return t; }
}  // namespace scs
namespace {
// register update over n produced bytes: runs of zero words through the zero-run operator, the rest slicing-by-8
uint32_t crc_feed(uint32_t reg, const uint8_t* p, size_t n) {
    const scs::CrcTables& C = scs::crc_tables();
    size_t i = 0, span = 0;
    while (i + 8 <= n) {
        uint64_t v;
        memcpy(&v, p + i, 8);
        if (v != 0) { i += 8; continue; }
        size_t q = i + 8;
        while (q + 8 <= n) { memcpy(&v, p + q, 8); if (v != 0) break; q += 8; }
        if (q - i >= 64) {                                           // short runs are not worth two operator applications
            reg = C.zeros(C.bytes(reg, p + span, i - span), q - i);
            span = q;
        }
        i = q;
    }
    return C.bytes(reg, p + span, n - span);
}
}  // namespace
namespace scs {
static const FixedCodes& fixed_codes() { static const FixedCodes t; return t; }

struct BitWriter {
    uint8_t* o; size_t cap, len = 0; uint64_t acc = 0; int nb = 0; bool overflow = false;
    BitWriter(uint8_t* out, size_t c) : o(out), cap(c) {}
    inline void put(uint32_t code, int n) {
        acc |= (uint64_t)code << nb; nb += n;
        if (nb >= 32) {
            if (len + 4 <= cap) { const uint32_t w = (uint32_t)acc; memcpy(o + len, &w, 4); len += 4; } else overflow = true;
            acc >>= 32; nb -= 32;
        }
    }
    inline void byte(uint8_t b) { if (len < cap) o[len++] = b; else overflow = true; }
    void align() { while (nb > 0) { byte((uint8_t)acc); acc >>= 8; nb -= 8; } nb = 0; acc = 0; }
};

static inline uint64_t load64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

}  // namespace scs

extern "C" int scs_gz_member(const void* data, int64_t nbytes, void* out, int64_t cap, int64_t* out_len) {
    using namespace scs;
    if ((!data && nbytes) || !out || !out_len || nbytes < 0 || cap < 32) { set_error("bad argument: scs_gz_member"); return SCS_ERR_ARG; }
    const CrcTables& C = crc_tables();
    const FixedCodes& F = fixed_codes();
    const uint8_t* d = static_cast<const uint8_t*>(data);
    const size_t n = (size_t)nbytes;
    BitWriter w(static_cast<uint8_t*>(out), (size_t)cap);
    static const uint8_t head[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff};
    for (int i = 0; i < 10; ++i) w.byte(head[i]);
    w.put(3, 3);                                                    // BFINAL = 1, BTYPE = 01 (fixed Huffman codes)
    const uint32_t d1 = FixedCodes::rev(0, 5), d8 = FixedCodes::rev(5, 5) | (1u << 5);   // distance 1; distance 8 = code 5 + extra bit 1
    uint32_t reg = 0xFFFFFFFFu;
    size_t p = 0, span = 0;                                          // [span, p): bytes whose CRC is still to be taken
    while (p < n) {
        if (p + 8 <= n && load64(d + p) == 0) {                      // a run of zero words
            size_t q = p + 8;
            while (q + 8 <= n && load64(d + q) == 0) q += 8;
            while (q < n && d[q] == 0) ++q;
            reg = C.zeros(C.bytes(reg, d + span, p - span), q - p);
            span = q;
            size_t left = q - p;
            if (p == 0 || d[p - 1] != 0) { w.put(F.lit[0], F.lit_n[0]); --left; }   // the byte a distance-1 match copies
            while (left >= 258 + 3 || left == 258) { w.put(F.len[258], F.len_n[258]); w.put(d1, 5); left -= 258; }
            if (left > 258) { w.put(F.len[left - 3], F.len_n[left - 3]); w.put(d1, 5); left = 3; }    // 259, 260: leave a match of 3
            if (left >= 3) { w.put(F.len[left], F.len_n[left]); w.put(d1, 5); left = 0; }
            while (left--) w.put(F.lit[0], F.lit_n[0]);
            p = q;
            continue;
        }
        if (p >= 8 && d[p] == d[p - 8]) {                            // bytes that repeat the bytes 8 back
            size_t m = 1;
            const size_t lim = n - p < 258 ? n - p : 258;
            while (m + 8 <= lim && load64(d + p + m) == load64(d + p + m - 8)) m += 8;
            while (m < lim && d[p + m] == d[p + m - 8]) ++m;
            if (m >= 3) { w.put(F.len[m], F.len_n[m]); w.put(d8, 6); p += m; continue; }
        }
        w.put(F.lit[d[p]], F.lit_n[d[p]]);
        ++p;
    }
    reg = C.bytes(reg, d + span, n - span);
    w.put(F.eob, F.eob_n);
    w.align();
    const uint32_t crc = ~reg, isize = (uint32_t)(n & 0xFFFFFFFFu);
    for (int i = 0; i < 4; ++i) w.byte((uint8_t)(crc >> (8 * i)));
    for (int i = 0; i < 4; ++i) w.byte((uint8_t)(isize >> (8 * i)));
    if (w.overflow) { set_error("scs_gz_member: output buffer too small (%lld bytes for %lld of input)", (long long)cap, (long long)nbytes); return SCS_ERR_ARG; }
    *out_len = (int64_t)w.len;
    return SCS_OK;
}

// The same member for a buffer given SPARSELY: nbytes of `fill` elements (item bytes each) with values[k] at element
// elem_index[k] (later duplicates win) - the buffer is built here, so that the caller's threads do not hold an interpreter
// lock for it.
extern "C" int scs_gz_member_scatter(int64_t nbytes, const void* fill, int item, const int64_t* elem_index, const void* values, int64_t nvals,
                                     void* out, int64_t cap, int64_t* out_len) {
    using scs::set_error;
    if (nbytes < 0 || item < 1 || item > 16 || nbytes % item != 0 || !fill || nvals < 0 || (nvals && (!elem_index || !values)) || !out || !out_len) {
        set_error("bad argument: scs_gz_member_scatter");
        return SCS_ERR_ARG;
    }
    const int64_t nelem = nbytes / item;
    uint8_t* buf = static_cast<uint8_t*>(malloc((size_t)nbytes + 8));
    if (!buf) { set_error("scs_gz_member_scatter: out of memory (%lld bytes)", (long long)nbytes); return SCS_ERR_ARG; }
    const uint8_t* f = static_cast<const uint8_t*>(fill);
    bool zero = true;
    for (int i = 0; i < item; ++i) zero = zero && f[i] == 0;
    if (zero) memset(buf, 0, (size_t)nbytes);
    else for (int64_t e = 0; e < nelem; ++e) memcpy(buf + e * item, f, (size_t)item);
    const uint8_t* v = static_cast<const uint8_t*>(values);
    for (int64_t k = 0; k < nvals; ++k) {
        const int64_t e = elem_index[k];
        if (e < 0 || e >= nelem) { free(buf); set_error("scs_gz_member_scatter: element %lld outside the member (%lld elements)", (long long)e, (long long)nelem); return SCS_ERR_ARG; }
        memcpy(buf + e * item, v + k * item, (size_t)item);
    }
    const int rc = scs_gz_member(buf, nbytes, out, cap, out_len);
    free(buf);
    return rc;
}
