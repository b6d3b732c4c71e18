// fx_pgzip.hpp -- the FIRST open of a single gzip stream on all cores of the host (round 3).
//
// A deflate stream is one chain: zlib inflates it with one core (7 s for the 3 Gbp file; the reference's first touch,
// gzread + zran_build_index, util.c:728-742, is the same chain twice).  The second open has restart points and runs in
// parallel (gzip_indexed_to_blob); this file gives the first one the same shape, the way pugz / rapidgzip do it:
//
//   1  the compressed bytes are cut into T pieces; thread t looks for the first place behind its cut where a DYNAMIC
//      deflate block begins -- it tries bit position after bit position: header fields in range, a complete code-length
//      code, complete literal/length and distance codes, an end-of-block code, and then the WHOLE block decodes without an
//      invalid symbol or an impossible distance and is followed by a plausible header.  (Stored and fixed blocks are not
//      looked for: the piece's predecessor simply decodes through them.)
//   2  every thread decodes from its block start on.  What lies before that is unknown to it, so it decodes into 16-bit
//      symbols: a byte, or a MARKER "byte k of the 32 KiB in front of my start" where a match reaches back past the start
//      (matches that copy markers copy them).  It stops where the next thread's start is -- exactly there, or the found
//      start was no true block boundary: then that piece is dropped and the thread decodes on to the one behind.
//   3  pieces in order: the last 32 KiB of everything before piece t resolve piece t's markers (only the 32 KiB tail has to
//      be done in order: 64 KiB of look-ups per piece; the bodies are resolved in parallel), bytes go out through `sink`.
//      Block boundaries met on the way (>= `spacing` bytes of output apart) become zran-style restart points with their
//      windows, so the index file gets the same table a serial first open captures.
//
// The CRC-32 / ISIZE of the trailer are checked against the folded CRCs of the pieces; any doubt anywhere (no block start
// found, a decode error on the true path, sizes that do not add up, a second gzip member) -> return false and the caller
// inflates serially with zlib, as before.  Host code only (plain C++17 + zlib for crc32): tests on any machine.
#pragma once
#include <zlib.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

namespace fx {
namespace pgz {

constexpr int WIN = 32768;
constexpr int LROOT = 10, DROOT = 8;
constexpr uint32_t LINKB = 1u << 15;

// growing array of 16-bit symbols without value-initialisation (a std::vector would zero what a block has just written).
// Its pages come straight from mmap, 2 MiB-aligned and advised as huge pages: a piece's output is tens of MB written once
// by one thread while a hundred others do the same -- with 4 KiB pages the first touches queue up behind the process's
// mmap lock, and giving 3 GB of them back took longer than inflating them (0.34 of 0.62 s).
template <class E> struct BufT {
    E *p = nullptr;
    size_t n = 0, cap = 0;
    BufT() = default;
    BufT(const BufT &) = delete;
    BufT &operator=(const BufT &) = delete;
    ~BufT() { release(); }
    static size_t bytes_of(size_t c) { return (c * sizeof(E) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1); }
    bool reserve(size_t c) {
        if (c <= cap) return true;
        if (p) c = std::max(c, cap + cap / 2);             // (a buffer that has to grow grows by half)
        const size_t nb = bytes_of(c);
        void *q = p ? mremap(p, bytes_of(cap), nb, MREMAP_MAYMOVE) : mmap(nullptr, nb, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (q == MAP_FAILED) return false;
        (void)madvise(q, nb, MADV_HUGEPAGE);
        p = (E *)q; cap = nb / sizeof(E);
        return true;
    }
    void release() { if (p) (void)munmap(p, bytes_of(cap)); p = nullptr; n = cap = 0; }
    size_t size() const { return n; }
    E operator[](size_t i) const { return p[i]; }
};
using Buf16 = BufT<uint16_t>;
using Buf8 = BufT<uint8_t>;

struct Bits {                      // LSB-first bit reader over a buffer readable 8 bytes past `n`
    const uint8_t *p;
    uint64_t n_bits, pos;
    uint64_t peek() const {        // >= 57 bits from pos on
        uint64_t w;
        memcpy(&w, p + (pos >> 3), 8);
        return w >> (pos & 7);
    }
};

// entries as in fx_inflate_par.hpp: L | extra << 4 | kind << 8 | payload << 16; LINK: sub-table of 2^L entries at payload
struct Tables {
    uint32_t l[1 << LROOT], d[1 << DROOT];
    std::vector<uint32_t> lp, dp;
};
static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t CLORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

static inline uint32_t rev_bits(uint32_t v, int n) {
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}
template <bool DIST> static inline uint32_t entry_of(int s, int L) {
    if (DIST) return s >= 30 ? (1u | (1u << 8)) : (uint32_t)L | ((uint32_t)DEXT[s] << 4) | ((uint32_t)DBASE[s] << 16);
    if (s < 256) return (uint32_t)L | ((uint32_t)s << 16);
    if (s == 256) return (uint32_t)L | (2u << 8);
    if (s >= 286) return 1u | (3u << 8);
    return (uint32_t)L | ((uint32_t)LEXT[s - 257] << 4) | (1u << 8) | ((uint32_t)LBASE[s - 257] << 16);
}
// canonical code from lengths; returns false for over-subscribed or (strict) incomplete codes.  allow_single: a code of one
// symbol of length 1 is accepted (the distance code of a block with one distance, RFC 1951 3.2.7)
template <bool DIST> static bool build(const uint8_t *len, int n, int root, uint32_t *lut, std::vector<uint32_t> &pool, bool allow_single) {
    int count[16] = {0};
    for (int s = 0; s < n; ++s) count[len[s]]++;
    const uint32_t inv = DIST ? (1u | (1u << 8)) : (1u | (3u << 8));
    const int size = 1 << root;
    for (int i = 0; i < size; ++i) lut[i] = inv;
    pool.clear();
    if (count[0] == n) return DIST;                           // no distance codes at all: a block of literals only
    int left = 1, used = 0;
    for (int L = 1; L <= 15; ++L) { left = (left << 1) - count[L]; used += count[L]; if (left < 0) return false; }
    if (left > 0 && !(allow_single && used == 1 && count[1] == 1)) return false;
    uint32_t next[16];
    next[1] = 0;
    for (int L = 1; L < 15; ++L) next[L + 1] = (next[L] + (uint32_t)count[L]) << 1;
    uint32_t code_of[320];
    for (int s = 0; s < n; ++s) if (len[s]) code_of[s] = rev_bits(next[len[s]]++, len[s]);
    for (int s = 0; s < n; ++s) {                             // codes that fit the root; what the longer ones need under theirs
        const int L = len[s];
        if (!L) continue;
        if (L <= root) { const uint32_t e = entry_of<DIST>(s, L); for (uint32_t i = code_of[s]; i < (uint32_t)size; i += 1u << L) lut[i] = e; }
        else {
            const uint32_t p = code_of[s] & (uint32_t)(size - 1), need = (uint32_t)(L - root), e = lut[p];
            if (!(e & LINKB) || need > (e & 15u)) lut[p] = LINKB | need;
        }
    }
    for (int p = 0; p < size; ++p)
        if (lut[p] & LINKB) {
            const uint32_t k = lut[p] & 15u, off = (uint32_t)pool.size();
            pool.resize(off + (1u << k), inv);
            lut[p] = LINKB | (off << 16) | k;
        }
    for (int s = 0; s < n; ++s) {
        const int L = len[s];
        if (L <= root) continue;
        const uint32_t e = lut[code_of[s] & (uint32_t)(size - 1)], k = e & 15u, off = e >> 16, rest = (uint32_t)(L - root);
        const uint32_t v = entry_of<DIST>(s, L);
        for (uint32_t j = code_of[s] >> root; j < (1u << k); j += 1u << rest) pool[off + j] = v;
    }
    return true;
}

enum { H_BAD = -1, H_STORED = 0, H_FIXED = 1, H_DYNAMIC = 2 };
// block header at b.pos: the type, `last`; a dynamic block's tables are built (strict checks: what the block search relies on)
static int header(Bits &b, Tables &T, int &last) {
    if (b.pos + 3 > b.n_bits) return H_BAD;
    uint64_t w = b.peek();
    last = (int)(w & 1u);
    const int type = (int)((w >> 1) & 3u);
    b.pos += 3;
    if (type == 3) return H_BAD;
    if (type == 0) return H_STORED;
    uint8_t lens[320];
    if (type == 1) {
        int s = 0;
        for (; s < 144; ++s) lens[s] = 8;
        for (; s < 256; ++s) lens[s] = 9;
        for (; s < 280; ++s) lens[s] = 7;
        for (; s < 288; ++s) lens[s] = 8;
        build<false>(lens, 288, LROOT, T.l, T.lp, false);
        for (s = 0; s < 30; ++s) lens[s] = 5;
        build<true>(lens, 30, DROOT, T.d, T.dp, false);
        return H_FIXED;
    }
    if (b.pos + 14 > b.n_bits) return H_BAD;
    w = b.peek();
    const int nlen = (int)(w & 31u) + 257, ndist = (int)((w >> 5) & 31u) + 1, ncode = (int)((w >> 10) & 15u) + 4;
    b.pos += 14;
    if (nlen > 286 || ndist > 30) return H_BAD;
    uint8_t cll[19] = {0};
    w = b.peek();
    for (int i = 0; i < ncode; ++i) cll[CLORD[i]] = (uint8_t)((w >> (3 * i)) & 7u);
    b.pos += 3 * (uint64_t)ncode;
    int count[8] = {0}, left = 1;
    for (int i = 0; i < 19; ++i) count[cll[i]]++;
    for (int L = 1; L <= 7; ++L) left = (left << 1) - count[L];
    if (left != 0) return H_BAD;                              // the code-length code must be complete
    uint32_t next[8];
    uint16_t cl[128];
    next[1] = 0;
    for (int L = 1; L < 7; ++L) next[L + 1] = (next[L] + (uint32_t)count[L]) << 1;
    for (int s = 0; s < 19; ++s) {
        const int L = cll[s];
        if (!L) continue;
        const uint32_t r = rev_bits(next[L]++, L);
        for (uint32_t i = r; i < 128u; i += 1u << L) cl[i] = (uint16_t)((s << 4) | L);
    }
    int idx = 0;
    while (idx < nlen + ndist) {
        if (b.pos + 14 > b.n_bits) return H_BAD;
        w = b.peek();
        const uint32_t e = cl[(uint32_t)w & 127u];
        const int L = (int)(e & 15u), sym = (int)(e >> 4);
        w >>= L; b.pos += (uint64_t)L;
        if (sym < 16) lens[idx++] = (uint8_t)sym;
        else {
            int v = 0, rep;
            if (sym == 16) { if (!idx) return H_BAD; v = lens[idx - 1]; rep = 3 + (int)(w & 3u); b.pos += 2; }
            else if (sym == 17) { rep = 3 + (int)(w & 7u); b.pos += 3; }
            else { rep = 11 + (int)(w & 127u); b.pos += 7; }
            if (idx + rep > nlen + ndist) return H_BAD;
            while (rep--) lens[idx++] = (uint8_t)v;
        }
    }
    if (lens[256] == 0) return H_BAD;
    if (!build<false>(lens, nlen, LROOT, T.l, T.lp, false)) return H_BAD;
    if (!build<true>(lens + nlen, ndist, DROOT, T.d, T.dp, true)) return H_BAD;
    return H_DYNAMIC;
}

// the symbols of one block into out16 (bytes, or 0x8000 | k = byte k of the unknown window in front of the piece);
// `have` = symbols the piece has produced so far (out16.size()).  check_only: nothing is stored, distances are only checked
// against what could exist (the block search).  Returns false on an invalid symbol / distance / the end of the input.
static bool block_codes(Bits &b, const Tables &T, Buf16 *out16, uint64_t &have, bool check_only) {
    uint16_t *p = check_only ? nullptr : out16->p;
    size_t o = check_only ? 0 : out16->n, cap = check_only ? 0 : out16->cap;
    const uint8_t *in = b.p;
    uint64_t pos = b.pos;
    const uint64_t n_bits = b.n_bits;
    const uint32_t *lt = T.l, *dt = T.d;
    const uint32_t *lp = T.lp.data(), *dp = T.dp.data();
    bool ok = false;
    for (;;) {
        if (pos >= n_bits) break;
        if (!check_only && o + 264 > cap) {                   // room for one more symbol (at most 258 entries)
            if (!out16->reserve(cap ? cap * 2 : ((size_t)1 << 20))) break;
            p = out16->p; cap = out16->cap;
        }
        uint64_t w;
        memcpy(&w, in + (pos >> 3), 8);
        w >>= (pos & 7);
        const uint32_t w32 = (uint32_t)w;
        uint32_t e = lt[w32 & ((1u << LROOT) - 1u)];
        if (e & LINKB) e = lp[(e >> 16) + ((w32 >> LROOT) & ((1u << (e & 15u)) - 1u))];
        const uint32_t L = e & 15u, kind = (e >> 8) & 3u;
        if (kind == 0) {
            pos += L;
            if (!check_only) p[o++] = (uint16_t)(e >> 16);
            ++have;
            continue;
        }
        if (kind == 2) { pos += L; ok = pos <= n_bits; break; }
        if (kind == 3) break;
        const uint32_t le = (e >> 4) & 15u;
        const uint32_t mlen = (e >> 16) + ((w32 >> L) & ((1u << le) - 1u));
        const uint32_t v = (uint32_t)(w >> (L + le));
        uint32_t d = dt[v & ((1u << DROOT) - 1u)];
        if (d & LINKB) d = dp[(d >> 16) + ((v >> DROOT) & ((1u << (d & 15u)) - 1u))];
        if (d & 0x100u) break;
        const uint32_t dl = d & 15u, de = (d >> 4) & 15u;
        const uint32_t dist = (d >> 16) + ((v >> dl) & ((1u << de) - 1u));
        pos += L + le + dl + de;
        if (pos > n_bits) break;
        if (dist > have + WIN) break;                         // further back than anything that can exist
        if (!check_only) {
            if (dist <= o) {                                  // the usual case: everything copied lies in this piece
                const uint16_t *src = p + o - dist;
                uint16_t *dst = p + o;
                if (dist >= 8) { for (uint32_t i = 0; i < mlen; i += 8) memcpy(dst + i, src + i, 16); }    // (room behind the match: 264 entries)
                else for (uint32_t i = 0; i < mlen; ++i) dst[i] = src[i];
            } else {
                for (uint32_t i = 0; i < mlen; ++i) {
                    const int64_t src = (int64_t)(o + i) - (int64_t)dist;
                    p[o + i] = src >= 0 ? p[src] : (uint16_t)(0x8000u | (uint32_t)(WIN + src));
                }
            }
            o += mlen;
        }
        have += mlen;
    }
    b.pos = pos;
    if (!check_only) out16->n = o;
    return ok;
}

// The same for a piece whose last 32 KiB hold no marker any more: from there on nothing it produces can depend on the unknown
// window, so the symbols are plain BYTES (half the memory, and nothing to resolve later).  out8 holds those 32 KiB at its start;
// every distance reaches at most that far back.
static bool block_codes8(Bits &b, const Tables &T, Buf8 *out8, uint64_t &have) {
    uint8_t *p = out8->p;
    size_t o = out8->n, cap = out8->cap;
    const uint8_t *in = b.p;
    uint64_t pos = b.pos;
    const uint64_t n_bits = b.n_bits;
    const uint32_t *lt = T.l, *dt = T.d;
    const uint32_t *lp = T.lp.data(), *dp = T.dp.data();
    bool ok = false;
    for (;;) {
        if (pos >= n_bits) break;
        if (o + 264 + 16 > cap) {
            if (!out8->reserve(cap * 2)) break;
            p = out8->p; cap = out8->cap;
        }
        uint64_t w;
        memcpy(&w, in + (pos >> 3), 8);
        w >>= (pos & 7);
        const uint32_t w32 = (uint32_t)w;
        uint32_t e = lt[w32 & ((1u << LROOT) - 1u)];
        if (e & LINKB) e = lp[(e >> 16) + ((w32 >> LROOT) & ((1u << (e & 15u)) - 1u))];
        const uint32_t L = e & 15u, kind = (e >> 8) & 3u;
        if (kind == 0) {
            pos += L;
            p[o++] = (uint8_t)(e >> 16);
            ++have;
            continue;
        }
        if (kind == 2) { pos += L; ok = pos <= n_bits; break; }
        if (kind == 3) break;
        const uint32_t le = (e >> 4) & 15u;
        const uint32_t mlen = (e >> 16) + ((w32 >> L) & ((1u << le) - 1u));
        const uint32_t v = (uint32_t)(w >> (L + le));
        uint32_t d = dt[v & ((1u << DROOT) - 1u)];
        if (d & LINKB) d = dp[(d >> 16) + ((v >> DROOT) & ((1u << (d & 15u)) - 1u))];
        if (d & 0x100u) break;
        const uint32_t dl = d & 15u, de = (d >> 4) & 15u;
        const uint32_t dist = (d >> 16) + ((v >> dl) & ((1u << de) - 1u));
        pos += L + le + dl + de;
        if (pos > n_bits) break;
        if (dist > o) break;                                  // (o >= WIN here: never true for a valid stream)
        const uint8_t *src = p + o - dist;
        uint8_t *dst = p + o;
        if (dist >= 16) { for (uint32_t i = 0; i < mlen; i += 16) memcpy(dst + i, src + i, 16); }       // (room behind the match: 264 + 16 bytes)
        else for (uint32_t i = 0; i < mlen; ++i) dst[i] = src[i];
        o += mlen;
        have += mlen;
    }
    b.pos = pos;
    out8->n = o;
    return ok;
}

// first bit position >= from (and < until) at which a dynamic block begins that decodes to its end and is followed by a
// plausible header; ~0 when there is none
static uint64_t find_block(const uint8_t *in, uint64_t n_bits, uint64_t from, uint64_t until) {
    Tables T;
    for (uint64_t p = from; p < until && p + 80 < n_bits; ++p) {
        // cheap rejections first: not the last block, dynamic, HLIT <= 29, HDIST <= 29
        uint64_t w;
        memcpy(&w, in + (p >> 3), 8);
        w >>= (p & 7);
        if ((w & 7u) != 4u) continue;                         // BFINAL = 0, BTYPE = 2 (bits: 0, then 0 1)
        if (((w >> 3) & 31u) > 29u || ((w >> 8) & 31u) > 29u) continue;
        Bits b{in, n_bits, p};
        int last = 0;
        if (header(b, T, last) != H_DYNAMIC) continue;
        uint64_t have = 0;
        if (!block_codes(b, T, nullptr, have, true)) continue;
        if (have < 1024) continue;                            // real blocks of a large stream are not tiny (a false start often is)
        if (b.pos + 3 <= n_bits) { const uint64_t nx = b.peek(); if (((nx >> 1) & 3u) == 3u) continue; }
        return p;
    }
    return ~0ull;
}

struct Point { uint64_t bit, out; };                          // a block boundary: bit position, output offset (of the whole stream)
struct Piece {
    uint64_t start_bit = ~0ull, end_bit = 0;                   // [start, end): where this piece's decode began and stopped
    Buf16 sym;                                                 // its output, markers included ...
    Buf8 byt;                                                  // ... until its last 32 KiB held none: those 32 KiB again, then plain bytes
    size_t out_len() const { return sym.n + (byt.n ? byt.n - (size_t)WIN : 0); }
    // outputs [a, a + len) with the markers looked up in Wn (the window in front of the piece, base = WIN - Wn.size());
    // false when a marker points in front of what exists
    bool resolve(const std::vector<uint8_t> &Wn, size_t a, size_t len, uint8_t *dst) const {
        const size_t base = (size_t)WIN - Wn.size();
        size_t i = 0;
        for (; i < len && a + i < sym.n; ++i) {
            const uint16_t s = sym.p[a + i];
            if (s & 0x8000u) { const uint32_t idx = s & 0x7FFFu; if (idx < base) return false; dst[i] = Wn[idx - base]; }
            else dst[i] = (uint8_t)s;
        }
        if (i < len) memcpy(dst + i, byt.p + WIN + (a + i - sym.n), len - i);
        return true;
    }
    void release() { sym.release(); byt.release(); }
    std::vector<Point> marks;                                  // block boundaries inside it (out: relative to the piece)
    bool dropped = false, failed = false, reached_end = false;
    uint32_t crc = 0;
    uint64_t out_base = 0;
};

// Result: the inflated bytes through sink(offset, data, len) (called from many threads, disjoint ranges), restart points
struct Piece;
struct Result {
    uint64_t out_bytes = 0;
    std::vector<uint64_t> pt_cin, pt_cout;
    std::vector<uint8_t> pt_bits, pt_has, pt_win;
    // The pieces' symbol buffers (two bytes per byte of output: GBs), still mapped: unmapping them takes about as long as
    // the inflate itself and holds the process's mmap lock on the way, so the CALLER says when -- release_later() hands them
    // to a detached thread (after its own stream / pinned-buffer teardown, which needs that lock too); the destructor
    // releases them on the spot if nobody did.
    std::function<void()> release_later;
    ~Result() { if (release_later) release_later(); }
};
using Sink = std::function<bool(int worker, uint64_t off, const uint8_t *data, size_t len)>;   // called from `workers` threads, disjoint ranges

// in: the WHOLE file (gzip wrapper included), readable 8 bytes past n.  alloc(total) is called once, before the first sink.
static bool inflate_parallel(const uint8_t *in, uint64_t n, int threads, uint64_t spacing, const std::function<bool(uint64_t total, int workers)> &alloc,
                             const Sink &sink, Result &res) {
    // ---- the gzip header (RFC 1952)
    if (n < 18 + 64 || in[0] != 0x1f || in[1] != 0x8b || in[2] != 8) return false;
    const int flg = in[3];
    uint64_t hp = 10;
    if (flg & 4) { if (hp + 2 > n) return false; hp += 2 + (in[hp] | (in[hp + 1] << 8)); }
    if (flg & 8) { while (hp < n && in[hp]) ++hp; ++hp; }
    if (flg & 16) { while (hp < n && in[hp]) ++hp; ++hp; }
    if (flg & 2) hp += 2;
    if (hp + 8 >= n) return false;
    const uint64_t dbits0 = hp * 8, dend = (n - 8) * 8;        // deflate data: [hp, n - 8) if the file is ONE member
    const uint64_t total_bits = dend - dbits0;
    int T = threads;
    const uint64_t min_piece = (uint64_t)4 << 23;              // 4 MiB of compressed bytes per piece at least
    if ((uint64_t)T * min_piece > total_bits) T = (int)std::max<uint64_t>(1, total_bits / min_piece);
    if (T < 2) return false;                                   // small files: the serial path is fine
    std::vector<Piece> pc((size_t)T);
    static const bool trace = [] { const char *e = getenv("FX_TRACE_PGZ"); return e && atoi(e) != 0; }();
    const auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (trace) fprintf(stderr, "[fxgpu] pgzip %-28s %8.1f ms (%d pieces)\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count(), T); };
    // ---- 1: block starts
    {
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t)
            th.emplace_back([&, t]() {
                const uint64_t from = dbits0 + total_bits * (uint64_t)t / (uint64_t)T, until = dbits0 + total_bits * (uint64_t)(t + 1) / (uint64_t)T;
                pc[(size_t)t].start_bit = find_block(in, dend, from, until);
            });
        pc[0].start_bit = dbits0;
        for (auto &x : th) x.join();
    }
    lap("block starts found");
    // ---- 2: decode.  A piece stops at the first LATER piece start it lands on exactly (block boundaries are only ever met at the
    // top of the loop); a start it passes without landing on it was no true boundary, and it just decodes on.
    std::vector<uint64_t> starts((size_t)T);
    for (int t = 0; t < T; ++t) starts[(size_t)t] = pc[(size_t)t].start_bit;
    {
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t]() {
                Piece &P = pc[(size_t)t];
                if (P.start_bit == ~0ull) { P.failed = true; return; }
                Tables Tb;
                Bits b{in, dend, P.start_bit};
                uint64_t have = 0, last_mark = 0;
                bool mode8 = false;                                         // the piece has left its markers behind: plain bytes from here on
                const size_t est = (size_t)(total_bits / 8 / (uint64_t)T * 4) + ((size_t)1 << 20);    // bytes this piece is likely to produce
                if (!P.sym.reserve(std::min<size_t>(est, (size_t)4 << 20))) { P.failed = true; return; }
                int nxt = t + 1;
                for (;;) {
                    while (nxt < T && (starts[(size_t)nxt] == ~0ull || starts[(size_t)nxt] < b.pos)) ++nxt;
                    if (nxt < T && starts[(size_t)nxt] == b.pos) break;         // exactly at a later piece's block: hand over
                    if (have - last_mark >= spacing && have > 0) { P.marks.push_back(Point{b.pos, have}); last_mark = have; }
                    if (!mode8 && P.sym.n >= (size_t)WIN) {
                        // Does anything in the last 32 KiB still stand for a byte of the unknown window?  If not, nothing
                        // produced from here on can (every distance stays within those 32 KiB): on with bytes.
                        const uint16_t *tl = P.sym.p + P.sym.n - WIN;
                        uint16_t any = 0;
                        for (int i = 0; i < WIN; ++i) any |= tl[i];
                        if (!(any & 0x8000u)) {
                            if (!P.byt.reserve(est + (size_t)WIN + 1024)) { P.failed = true; break; }
                            for (int i = 0; i < WIN; ++i) P.byt.p[i] = (uint8_t)tl[i];
                            P.byt.n = (size_t)WIN;
                            mode8 = true;
                        }
                    }
                    int last = 0;
                    const int type = header(b, Tb, last);
                    if (type == H_BAD) { P.failed = true; break; }
                    if (type == H_STORED) {
                        const uint64_t bp = (b.pos + 7) >> 3;
                        if ((bp + 4) * 8 > dend) { P.failed = true; break; }
                        const uint32_t len = in[bp] | ((uint32_t)in[bp + 1] << 8), nl = in[bp + 2] | ((uint32_t)in[bp + 3] << 8);
                        if ((len ^ 0xFFFFu) != nl || (bp + 4 + len) * 8 > dend) { P.failed = true; break; }
                        if (mode8) {
                            const size_t o = P.byt.n;
                            if (!P.byt.reserve(o + len + 264 + 16)) { P.failed = true; break; }
                            memcpy(P.byt.p + o, in + bp + 4, len);
                            P.byt.n = o + len;
                        } else {
                            const size_t o = P.sym.n;
                            if (!P.sym.reserve(o + len + 264)) { P.failed = true; break; }
                            for (uint32_t i = 0; i < len; ++i) P.sym.p[o + i] = in[bp + 4 + i];
                            P.sym.n = o + len;
                        }
                        have += len;
                        b.pos = (bp + 4 + len) * 8;
                    } else if (mode8 ? !block_codes8(b, Tb, &P.byt, have) : !block_codes(b, Tb, &P.sym, have, false)) { P.failed = true; break; }
                    if (last) { P.reached_end = true; break; }
                }
                P.end_bit = b.pos;
            });
        for (auto &x : th) x.join();
    }
    lap("pieces decoded");
    if (trace) {
        uint64_t s16 = 0, s8 = 0;
        for (const Piece &P : pc) { s16 += P.sym.n; s8 += P.byt.n; }
        fprintf(stderr, "[fxgpu] pgzip symbols kept with markers: %.1f MB of output, as plain bytes: %.1f MB\n", s16 / 1e6, s8 / 1e6);
    }
    // ---- the chain from piece 0: every piece hands over to the piece whose start it stopped at; the others are dropped
    std::vector<int> live;
    uint64_t total = 0;
    for (int t = 0;;) {
        Piece &P = pc[(size_t)t];
        if (P.failed) return false;
        live.push_back(t);
        P.out_base = total;
        total += P.out_len();
        if (P.reached_end) break;
        int j = t + 1;
        while (j < T && starts[(size_t)j] != P.end_bit) ++j;
        if (j >= T) return false;
        t = j;
    }
    for (int t = 0; t < T; ++t)
        if (std::find(live.begin(), live.end(), t) == live.end()) { pc[(size_t)t].release(); pc[(size_t)t].dropped = true; }
    // the stream must end at the trailer (up to 7 bits of padding) and ISIZE must agree
    const Piece &Lp = pc[(size_t)live.back()];
    if (((Lp.end_bit + 7) >> 3) != n - 8) return false;
    const uint8_t *tr = in + n - 8;
    const uint32_t want_crc = tr[0] | (tr[1] << 8) | (tr[2] << 16) | ((uint32_t)tr[3] << 24);
    const uint32_t isz = tr[4] | (tr[5] << 8) | (tr[6] << 16) | ((uint32_t)tr[7] << 24);
    if (isz != (uint32_t)total || total == 0) return false;
    const int W = std::min<int>(threads, (int)live.size());
    if (!alloc(total, W)) return false;
    // ---- 3: windows in order (tails only), then the bodies in parallel
    std::vector<std::vector<uint8_t>> win(live.size());       // win[k]: the (up to) 32 KiB in front of live piece k
    {
        std::vector<uint8_t> cur;                              // the last <= 32 KiB of everything resolved so far
        for (size_t k = 0; k < live.size(); ++k) {
            const Piece &P = pc[(size_t)live[k]];
            win[k] = cur;
            const size_t m = P.out_len(), take = std::min<size_t>(m, WIN);
            std::vector<uint8_t> tail(take);
            if (!P.resolve(cur, m - take, take, tail.data())) return false;      // a marker that reaches back before the start of the stream
            if (take == (size_t)WIN) cur.swap(tail);
            else {
                std::vector<uint8_t> nc;
                const size_t keep = std::min(cur.size(), (size_t)WIN - take);
                nc.insert(nc.end(), cur.end() - (std::ptrdiff_t)keep, cur.end());
                nc.insert(nc.end(), tail.begin(), tail.end());
                cur.swap(nc);
            }
        }
    }
    lap("windows chained");
    std::atomic<size_t> next_piece(0);
    std::atomic<int> bad(0);
    {
        std::vector<std::thread> th;
        for (int wkr = 0; wkr < W; ++wkr)
            th.emplace_back([&, wkr]() {
                std::vector<uint8_t> buf;
                for (size_t k; (k = next_piece.fetch_add(1)) < live.size() && !bad.load();) {
                    Piece &P = pc[(size_t)live[k]];
                    const std::vector<uint8_t> &Wn = win[k];
                    const size_t m = P.out_len();
                    uint32_t crc = 0;
                    const size_t STEP = 8u << 20;
                    for (size_t a = 0; a < m && !bad.load(); a += STEP) {
                        const size_t len = std::min(STEP, m - a);
                        const uint8_t *data;
                        if (a >= P.sym.n) data = P.byt.p + WIN + (a - P.sym.n);        // plain bytes already: nothing to resolve, nothing to copy
                        else {
                            buf.resize(len);
                            if (!P.resolve(Wn, a, len, buf.data())) { bad.store(1); break; }
                            data = buf.data();
                        }
                        crc = (uint32_t)crc32(crc, data, (uInt)len);
                        if (!sink(wkr, P.out_base + a, data, len)) { bad.store(2); break; }
                    }
                    P.crc = crc;
                }
            });
        for (auto &x : th) x.join();
    }
    lap("bodies resolved, sunk");
    if (bad.load()) return false;
    uint32_t crc = pc[(size_t)live[0]].crc;
    for (size_t k = 1; k < live.size(); ++k) crc = (uint32_t)crc32_combine(crc, pc[(size_t)live[k]].crc, (z_off_t)pc[(size_t)live[k]].out_len());
    if (crc != want_crc) return false;
    // ---- restart points: the start of the deflate data, piece starts and the block boundaries marked inside the pieces, at
    // least `spacing` bytes of output apart (which ones: in order; their 32 KiB windows: in parallel)
    res.out_bytes = total;
    struct Acc { size_t k; uint64_t bit, out_rel; bool at_start; };
    std::vector<Acc> acc;
    {
        uint64_t last_out = 0;
        for (size_t k = 0; k < live.size(); ++k) {
            const Piece &P = pc[(size_t)live[k]];
            if (k == 0) acc.push_back(Acc{0, P.start_bit, 0, true});
            else if (P.out_base - last_out >= spacing && win[k].size() == (size_t)WIN) { acc.push_back(Acc{k, P.start_bit, 0, true}); last_out = P.out_base; }
            for (const Point &mk : P.marks) {
                if (P.out_base + mk.out - last_out < spacing || P.out_base + mk.out < (uint64_t)WIN) continue;
                acc.push_back(Acc{k, mk.bit, mk.out, false});
                last_out = P.out_base + mk.out;
            }
        }
    }
    const size_t np = acc.size();
    res.pt_cin.resize(np); res.pt_cout.resize(np); res.pt_bits.resize(np); res.pt_has.resize(np);
    res.pt_win.resize((np - 1) * (size_t)WIN);
    {
        std::atomic<size_t> nextp(0);
        std::vector<std::thread> th;
        for (int wkr = 0; wkr < W; ++wkr)
            th.emplace_back([&]() {
                for (size_t i; (i = nextp.fetch_add(1)) < np;) {
                    const Acc &A = acc[i];
                    const Piece &P = pc[(size_t)live[A.k]];
                    const std::vector<uint8_t> &Wn = win[A.k];
                    res.pt_cin[i] = (A.bit + 7) >> 3;
                    res.pt_bits[i] = (uint8_t)((8 - (A.bit & 7)) & 7);
                    res.pt_cout[i] = P.out_base + A.out_rel;
                    res.pt_has[i] = i ? 1 : 0;
                    if (!i) continue;
                    uint8_t *w = res.pt_win.data() + (i - 1) * (size_t)WIN;
                    // bytes out_rel - WIN .. out_rel - 1 of the piece's output (< 0: the window in front of it)
                    const int64_t q0 = (int64_t)A.out_rel - WIN;
                    int j = 0;
                    for (; j < WIN && q0 + j < 0; ++j) { const int64_t jj = (int64_t)Wn.size() + q0 + j; w[j] = jj >= 0 ? Wn[(size_t)jj] : 0; }
                    if (j < WIN) (void)P.resolve(Wn, (size_t)(q0 + j), (size_t)(WIN - j), w + j);
                }
            });
        for (auto &x : th) x.join();
    }
    lap("restart points");
    {
        auto *junk = new std::vector<Piece>(std::move(pc));
        res.release_later = [junk]() {
            std::thread([junk]() {
                // ... and not at once: unmapping GBs keeps the process's mmap lock busy, and what the caller does next -- device
                // allocations for the scan of the bytes it has just got -- needs that lock (measured: fx_fasta_build 177 ms
                // instead of 1 ms).  The pages are idle memory for that long (FX_PGZ_RELEASE_DELAY_MS, default 1500).
                static const int delay_ms = [] { const char *e = getenv("FX_PGZ_RELEASE_DELAY_MS"); return e ? atoi(e) : 1500; }();
                if (delay_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(delay_ms));
                for (Piece &P : *junk) P.release();
                delete junk;
            }).detach();
        };
    }
    return true;
}

}  // namespace pgz
}  // namespace fx
