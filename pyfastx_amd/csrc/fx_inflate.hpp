// fx_inflate.hpp -- BGZF member inflate on the GPU (K9).
//
// A BGZF file (bgzip) is a concatenation of gzip members, each holding <= 64 KiB
// of data compressed independently (no LZ77 window crosses a member) and
// carrying its own compressed size in a 'BC' extra field, so the host can
// build the member table with one cheap header walk and every member can be
// inflated in parallel.  This replaces, for BGZF inputs, the serial
// gzread() inflate that feeds the reference's scan (kseq.c:70) and the
// zran_seek/zran_read random access (index.c:685-686): the inflated stream
// becomes the resident blob the other kernels work on.
//
// One work-item per member.  DEFLATE decoding is bit-serial, so the parallelism
// is across members (a 3 Gbp genome is ~47 k of them).  The decoder is the
// canonical-code "count/symbol" scheme (no 2 KiB fast tables): per work-item
// state is 16+288+16+32 16-bit words, kept in LDS (44 KiB per 64-lane
// workgroup), code lengths for dynamic blocks in private memory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fx {

constexpr int INFL_BLOCK = 64;           // one wave per workgroup: 44 KiB of LDS tables
constexpr int MAXBITS = 15, MAXLCODES = 286, MAXDCODES = 30, FIXLCODES = 288;

enum InflStatus { INFL_OK = 0, INFL_EINPUT = 1, INFL_EOUTPUT = 2, INFL_EBLOCK = 3, INFL_ECODES = 4, INFL_EDIST = 5,
                  INFL_ESIZE = 6 };

struct BitIn {
    const uint8_t *p, *end;
    uint64_t buf;
    int cnt;
    int err;
};

typedef uint64_t __attribute__((aligned(1))) uint64_u;    // 8 bytes at any address (gfx9 unaligned access mode)

// One unaligned 8-byte load tops the bit buffer up to >= 56 bits: every input byte costs 1/7 of a memory round
// trip instead of one (the decoder is a chain of dependent round trips, so this is what it runs at).  The bits of
// a partially consumed byte are OR-ed in again by the next refill at the same position: same data, harmless.
__device__ __forceinline__ void refill(BitIn &b) {
    if (b.p + 8 <= b.end) {
        b.buf |= *reinterpret_cast<const uint64_u *>(b.p) << b.cnt;
        b.p += (63 - b.cnt) >> 3;
        b.cnt |= 56;
        return;
    }
    while (b.cnt <= 56 && b.p < b.end) { b.buf |= (uint64_t)(*b.p++) << b.cnt; b.cnt += 8; }
}
__device__ __forceinline__ uint32_t getbits(BitIn &b, int n) {
    if (b.cnt < n) { refill(b); if (b.cnt < n) { b.err = 1; return 0; } }
    const uint32_t v = (uint32_t)(b.buf & ((1ull << n) - 1ull));
    b.buf >>= n; b.cnt -= n;
    return v;
}

// LDS-resident canonical Huffman table of one work-item: column `lane` of cnt[][64] / sym[][64].
struct Huff { uint16_t *cnt; uint16_t *sym; };     // element i of this lane is ptr[i * INFL_BLOCK]

__device__ __forceinline__ int decode(BitIn &b, const Huff &h) {
    if (b.cnt < MAXBITS) refill(b);
    uint32_t bits = (uint32_t)b.buf;
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= MAXBITS; ++len) {
        code |= (int)(bits & 1u);
        bits >>= 1;
        const int count = h.cnt[len * INFL_BLOCK];
        if (code - count < first) {
            if (b.cnt < len) { b.err = 1; return -1; }
            b.buf >>= len; b.cnt -= len;
            return h.sym[(index + (code - first)) * INFL_BLOCK];
        }
        index += count; first += count;
        first <<= 1; code <<= 1;
    }
    b.err = 1;
    return -1;
}

// Build count/symbol from code lengths (canonical codes).  Returns <0 for an over-subscribed set.
__device__ inline int construct(const Huff &h, const uint8_t *length, int n) {
    for (int len = 0; len <= MAXBITS; ++len) h.cnt[len * INFL_BLOCK] = 0;
    for (int s = 0; s < n; ++s) h.cnt[length[s] * INFL_BLOCK]++;
    if (h.cnt[0] == n) return 0;
    int left = 1;
    for (int len = 1; len <= MAXBITS; ++len) {
        left <<= 1;
        left -= h.cnt[len * INFL_BLOCK];
        if (left < 0) return left;
    }
    uint16_t offs[MAXBITS + 1];
    offs[1] = 0;
    for (int len = 1; len < MAXBITS; ++len) offs[len + 1] = offs[len] + h.cnt[len * INFL_BLOCK];
    for (int s = 0; s < n; ++s)
        if (length[s] != 0) h.sym[(offs[length[s]]++) * INFL_BLOCK] = (uint16_t)s;
    return left;
}

__device__ const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115,
                                       131, 163, 195, 227, 258};
__device__ const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025,
                                       1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t CLORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// Literal/length + distance codes of one block -> bytes at out[o...].
__device__ inline int inflate_codes(BitIn &b, const Huff &lc, const Huff &dc, uint8_t *out, int64_t &o, int64_t cap) {
    for (;;) {
        int sym = decode(b, lc);
        if (sym < 0) return INFL_EINPUT;
        if (sym < 256) {
            if (o >= cap) return INFL_EOUTPUT;
            out[o++] = (uint8_t)sym;
        } else if (sym == 256) {
            return INFL_OK;
        } else {
            sym -= 257;
            if (sym >= 29) return INFL_ECODES;
            const int len = LBASE[sym] + (int)getbits(b, LEXT[sym]);
            const int ds = decode(b, dc);
            if (ds < 0 || ds >= 30) return INFL_EINPUT;
            const int64_t dist = DBASE[ds] + (int64_t)getbits(b, DEXT[ds]);
            if (b.err) return INFL_EINPUT;
            if (dist > o) return INFL_EDIST;                 // BGZF members never reference outside themselves
            if (o + len > cap) return INFL_EOUTPUT;
            const uint8_t *src = out + o - dist;
            uint8_t *dst = out + o;
            int i = 0;
            if (dist >= 8) {            // an 8-byte step never reads a byte written in the same step, so the loads of
                                        // a 32-byte group are independent: one round trip per group, not per byte
                for (; i + 32 <= len && dist >= 32; i += 32) {
                    const uint64_t a = *reinterpret_cast<const uint64_u *>(src + i), c = *reinterpret_cast<const uint64_u *>(src + i + 8);
                    const uint64_t d = *reinterpret_cast<const uint64_u *>(src + i + 16), e = *reinterpret_cast<const uint64_u *>(src + i + 24);
                    *reinterpret_cast<uint64_u *>(dst + i) = a; *reinterpret_cast<uint64_u *>(dst + i + 8) = c;
                    *reinterpret_cast<uint64_u *>(dst + i + 16) = d; *reinterpret_cast<uint64_u *>(dst + i + 24) = e;
                }
                for (; i + 8 <= len; i += 8) *reinterpret_cast<uint64_u *>(dst + i) = *reinterpret_cast<const uint64_u *>(src + i);
                if (i < len) {          // 1..7 bytes left: one more load, byte stores (never past o + len)
                    uint64_t t = *reinterpret_cast<const uint64_u *>(src + i);
                    for (; i < len; ++i) { dst[i] = (uint8_t)t; t >>= 8; }
                }
            }
            for (; i < len; ++i) dst[i] = src[i];            // byte order matters when dist < len (run replication)
            o += len;
        }
    }
}

// members: cdata_off/cdata_len (compressed payload inside cbuf), uoff (offset in the inflated stream), isize.
__global__ __launch_bounds__(INFL_BLOCK) void k_bgzf_inflate(const uint8_t *__restrict__ cbuf,
                                                            const int64_t *__restrict__ cdata_off,
                                                            const int32_t *__restrict__ cdata_len,
                                                            const int64_t *__restrict__ uoff,
                                                            const int32_t *__restrict__ isize, int64_t nmem,
                                                            uint8_t *__restrict__ data, int32_t *__restrict__ status) {
    __shared__ uint16_t t_lcnt[(MAXBITS + 1) * INFL_BLOCK], t_lsym[FIXLCODES * INFL_BLOCK];
    __shared__ uint16_t t_dcnt[(MAXBITS + 1) * INFL_BLOCK], t_dsym[32 * INFL_BLOCK];
    const int lane = threadIdx.x;
    const int64_t m = (int64_t)blockIdx.x * INFL_BLOCK + lane;
    if (m >= nmem) return;
    Huff lc{t_lcnt + lane, t_lsym + lane}, dc{t_dcnt + lane, t_dsym + lane};
    BitIn b;
    b.p = cbuf + cdata_off[m]; b.end = b.p + cdata_len[m]; b.buf = 0; b.cnt = 0; b.err = 0;
    uint8_t *out = data + uoff[m];
    const int64_t cap = isize[m];
    int64_t o = 0;
    int st = INFL_OK, last;
    uint8_t lengths[MAXLCODES + MAXDCODES + 4];
    do {
        last = (int)getbits(b, 1);
        const int type = (int)getbits(b, 2);
        if (b.err) { st = INFL_EINPUT; break; }
        if (type == 0) {                                     // stored
            b.buf >>= (b.cnt & 7); b.cnt -= (b.cnt & 7);     // to the byte boundary
            const uint32_t len = getbits(b, 16), nlen = getbits(b, 16);
            if (b.err || (len ^ 0xFFFFu) != nlen) { st = INFL_EBLOCK; break; }
            if (o + (int64_t)len > cap) { st = INFL_EOUTPUT; break; }
            for (uint32_t i = 0; i < len; ++i) { out[o++] = (uint8_t)getbits(b, 8); }
            if (b.err) { st = INFL_EINPUT; break; }
        } else if (type == 1) {                              // fixed codes
            int s = 0;
            for (; s < 144; ++s) lengths[s] = 8;
            for (; s < 256; ++s) lengths[s] = 9;
            for (; s < 280; ++s) lengths[s] = 7;
            for (; s < FIXLCODES; ++s) lengths[s] = 8;
            construct(lc, lengths, FIXLCODES);
            for (s = 0; s < MAXDCODES; ++s) lengths[s] = 5;
            construct(dc, lengths, MAXDCODES);
            st = inflate_codes(b, lc, dc, out, o, cap);
            if (st) break;
        } else if (type == 2) {                              // dynamic codes
            const int nlen = (int)getbits(b, 5) + 257, ndist = (int)getbits(b, 5) + 1, ncode = (int)getbits(b, 4) + 4;
            if (b.err || nlen > MAXLCODES || ndist > MAXDCODES) { st = INFL_ECODES; break; }
            int idx = 0;
            for (; idx < ncode; ++idx) lengths[CLORDER[idx]] = (uint8_t)getbits(b, 3);
            for (; idx < 19; ++idx) lengths[CLORDER[idx]] = 0;
            if (construct(lc, lengths, 19) != 0) { st = INFL_ECODES; break; }     // code-length code must be complete
            idx = 0;
            while (idx < nlen + ndist) {
                int sym = decode(b, lc);
                if (sym < 0) { st = INFL_EINPUT; break; }
                if (sym < 16) lengths[idx++] = (uint8_t)sym;
                else {
                    int len = 0, rep;
                    if (sym == 16) {
                        if (idx == 0) { st = INFL_ECODES; break; }
                        len = lengths[idx - 1]; rep = 3 + (int)getbits(b, 2);
                    } else if (sym == 17) rep = 3 + (int)getbits(b, 3);
                    else rep = 11 + (int)getbits(b, 7);
                    if (idx + rep > nlen + ndist) { st = INFL_ECODES; break; }
                    while (rep--) lengths[idx++] = (uint8_t)len;
                }
            }
            if (st) break;
            if (b.err) { st = INFL_EINPUT; break; }
            if (lengths[256] == 0) { st = INFL_ECODES; break; }
            int err = construct(lc, lengths, nlen);
            if (err < 0 || (err > 0 && nlen - lc.cnt[0] != 1)) { st = INFL_ECODES; break; }
            err = construct(dc, lengths + nlen, ndist);
            if (err < 0 || (err > 0 && ndist - dc.cnt[0] != 1)) { st = INFL_ECODES; break; }
            st = inflate_codes(b, lc, dc, out, o, cap);
            if (st) break;
        } else { st = INFL_EBLOCK; break; }
    } while (!last);
    if (st == INFL_OK && o != cap) st = INFL_ESIZE;          // ISIZE of the member trailer must match
    status[m] = st;
}

}  // namespace fx
