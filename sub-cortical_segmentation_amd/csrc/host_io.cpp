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
// The member's ISIZE is checked; its CRC-32 is NOT (that alone costs as much as the system zlib's whole inflate).
#include <cerrno>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
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

struct Window {
    std::vector<uint8_t> buf;
    size_t w = 0;           // bytes in the window
    int64_t base = 0;       // stream offset of buf[0]
    Window() : buf(kHist + kChunk + kSlack) {}
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

static int gz_drive(const char* path, Gather& g, int64_t* inflated_bytes) {
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
            if (inflate_stream(b, win, g, &done) != 0) { set_error("%s: corrupt or truncated deflate stream", path); return SCS_ERR_ARG; }
            if (done) break;
            // unread whole bytes in the accumulator go back to the input
            b.drop(b.cnt & 7);
            p = b.p - b.cnt / 8;
            if (end - p < 8) goto trunc;
            const uint32_t isize = (uint32_t)p[4] | ((uint32_t)p[5] << 8) | ((uint32_t)p[6] << 16) | ((uint32_t)p[7] << 24);
            const int64_t produced = win.base + (int64_t)win.w - before;
            if ((uint32_t)produced != isize) { set_error("%s: gzip member length mismatch", path); return SCS_ERR_ARG; }
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
    return gz_drive(path, g, inflated_bytes);
}

extern "C" int scs_gz_read(const char* path, int64_t offset, int64_t nbytes, void* out) {
    using scs::set_error;
    if (!path || offset < 0 || nbytes < 0 || (!out && nbytes)) { set_error("bad argument: scs_gz_read"); return SCS_ERR_ARG; }
    Gather g{nullptr, 0, 0, 1, nullptr};
    g.range_out = static_cast<uint8_t*>(out); g.range_lo = offset; g.range_hi = offset + nbytes;
    if (nbytes == 0) return SCS_OK;
    return gz_drive(path, g, nullptr);
}
