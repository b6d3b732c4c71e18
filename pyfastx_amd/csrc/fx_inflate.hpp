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
// Two kernels.  DEFLATE decoding is bit-serial, but only the DECODING is: what a match copies does not
// influence how the following bits are parsed.  So
//   k_bgzf_decode   one work-item per member parses the bit stream: literals are stored straight to
//                   their final position, matches become tokens (dst, len, dist) in a side buffer.  No
//                   load of previously written output, so the only memory latency on the critical path
//                   is the input, and that is read one 8-byte word ahead of use.
//   k_bgzf_copy     one WAVE per member resolves the tokens, 64 at a time: a token whose source lies
//                   before the batch's first output byte is independent of the rest of the batch (the
//                   common case: distances are long against 64 tokens' worth of output) and is copied
//                   by its own lane; the others follow in order, each as one wave-wide gather --
//                   out[dst + j] = out[src + j % dist] has no dependency inside a token, run
//                   replication (dist < len) included.
// This replaces the serial gzread() inflate that feeds the reference's scan (kseq.c:70) and the
// zran_seek/zran_read random access (index.c:685-686).  The decoder is the canonical-code
// "count/symbol" scheme (no 2 KiB fast tables): per work-item state is 16+288+16+32 16-bit words,
// kept in LDS (44 KiB per 64-lane workgroup), code lengths for dynamic blocks in private memory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fx {

#ifndef FX_INFL_BLOCK
#define FX_INFL_BLOCK 64
#endif
constexpr int INFL_BLOCK = FX_INFL_BLOCK; // decode: members per workgroup (one wave, possibly partly filled); 0.7 KiB of LDS tables each
constexpr int MAXBITS = 15, MAXLCODES = 286, MAXDCODES = 30, FIXLCODES = 288;

enum InflStatus { INFL_OK = 0, INFL_EINPUT = 1, INFL_EOUTPUT = 2, INFL_EBLOCK = 3, INFL_ECODES = 4, INFL_EDIST = 5,
                  INFL_ESIZE = 6 };

typedef uint64_t __attribute__((aligned(1))) uint64_u;    // 8 bytes at any address (gfx9 unaligned access mode)

// Bit reader.  The compressed bytes are consumed through a window (w0, w1, w2) of aligned 8-byte words; when
// the read position crosses into w1, w2 moves up and the word after it is requested -- two words (~6 symbols of
// work) before it is needed, so the decoder does not wait for input.  (The buffer is readable 32 bytes past
// the last member.)
struct BitIn {
    const uint8_t *p, *end;      // next unconsumed byte, end of the member's payload
    const uint64_t *q;           // aligned word that holds *p
    uint64_t w0, w1, w2;         // q[0], q[1], q[2]
    uint64_t buf;
    int cnt;
    int err;
};
__device__ __forceinline__ void bit_init(BitIn &b, const uint8_t *p, const uint8_t *end) {
    b.p = p; b.end = end; b.buf = 0; b.cnt = 0; b.err = 0;
    b.q = reinterpret_cast<const uint64_t *>((uintptr_t)p & ~(uintptr_t)7);
    b.w0 = b.q[0]; b.w1 = b.q[1]; b.w2 = b.q[2];
}
// tops the bit buffer up to >= 56 bits; the bits of a partially consumed byte are OR-ed in again by the
// next refill at the same position: same data, harmless
__device__ __forceinline__ void refill(BitIn &b) {
    const int s = (int)((uintptr_t)b.p & 7) * 8;
    const uint64_t w = s ? (b.w0 >> s) | (b.w1 << (64 - s)) : b.w0;      // the 8 bytes at p
    int64_t avail = b.end - b.p;                                          // bytes of this member left
    uint64_t v = w;
    if (avail < 8) v = avail <= 0 ? 0 : (w & (~0ull >> (64 - 8 * avail)));   // never feed bytes of the next member
    b.buf |= v << b.cnt;
    int take = (63 - b.cnt) >> 3;
    if (take > avail) take = avail < 0 ? 0 : (int)avail;
    b.p += take;
    b.cnt += take * 8;
    const uint64_t *nq = reinterpret_cast<const uint64_t *>((uintptr_t)b.p & ~(uintptr_t)7);
    if (nq != b.q) { b.q = nq; b.w0 = b.w1; b.w1 = b.w2; b.w2 = nq[2]; }   // at most one word forward: take <= 7
}
__device__ __forceinline__ uint32_t getbits(BitIn &b, int n) {
    if (b.cnt < n) { refill(b); if (b.cnt < n) { b.err = 1; return 0; } }
    const uint32_t v = (uint32_t)(b.buf & ((1ull << n) - 1ull));
    b.buf >>= n; b.cnt -= n;
    return v;
}

// LDS-resident canonical Huffman table of one work-item: column `lane` of cnt[][64] / sym[][64].
struct Huff { uint16_t *cnt; uint16_t *sym; };     // element i of this lane is ptr[i * INFL_BLOCK]

// The per-length code counts of a table live in REGISTERS while it is in use (16 x 16 bits in 8 VGPRs): the
// canonical decode walks code lengths 1..15 without touching LDS, and only the final symbol is one LDS read.
// (With one wave per SIMD at most -- a genome is ~47 k members -- nothing hides a chain of dependent LDS reads.)
struct HuffCnt { uint32_t c[8]; };
__device__ __forceinline__ void load_counts(const Huff &h, HuffCnt &r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) r.c[i] = (uint32_t)h.cnt[(2 * i) * INFL_BLOCK] | ((uint32_t)h.cnt[(2 * i + 1) * INFL_BLOCK] << 16);
}

__device__ __forceinline__ int decode(BitIn &b, const Huff &h, const HuffCnt &r) {
    if (b.cnt < MAXBITS) refill(b);
    uint32_t bits = (uint32_t)b.buf;
    int code = 0, first = 0, index = 0;
#pragma unroll
    for (int len = 1; len <= MAXBITS; ++len) {
        code |= (int)(bits & 1u);
        bits >>= 1;
        const int count = (int)((r.c[len >> 1] >> (16 * (len & 1))) & 0xFFFFu);
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

// length / distance bases and extra-bit counts of RFC 1951 3.2.5 in closed form (no table loads on the decode path)
__device__ __forceinline__ void len_code(int sym, int &base, int &ext) {       // sym = literal/length code - 257, 0..28
    if (sym < 8) { base = 3 + sym; ext = 0; }
    else if (sym == 28) { base = 258; ext = 0; }
    else { ext = (sym - 4) >> 2; base = ((4 + (sym & 3)) << ext) + 3; }
}
__device__ __forceinline__ void dist_code(int ds, int &base, int &ext) {       // 0..29
    if (ds < 4) { base = 1 + ds; ext = 0; }
    else { ext = (ds >> 1) - 1; base = ((2 + (ds & 1)) << ext) + 1; }
}
__device__ const uint8_t CLORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// match token: dst (17 bits) | len (9 bits) << 17 | dist (16 bits) << 26
__device__ __forceinline__ uint64_t tok_pack(int64_t dst, int len, int64_t dist) { return (uint64_t)dst | ((uint64_t)len << 17) | ((uint64_t)dist << 26); }

// Literal/length + distance codes of one block: literals -> out[o...], matches -> tok[nt...].
__device__ inline int inflate_codes(BitIn &b, const Huff &lc, const Huff &dc, uint8_t *out, int64_t &o, int64_t cap,
                                    uint64_t *tok, int &nt) {
    HuffCnt lr, dr;
    load_counts(lc, lr); load_counts(dc, dr);
    for (;;) {
        int sym = decode(b, lc, lr);
        if (sym < 0) return INFL_EINPUT;
        if (sym < 256) {
            if (o >= cap) return INFL_EOUTPUT;
            out[o++] = (uint8_t)sym;
        } else if (sym == 256) {
            return INFL_OK;
        } else {
            sym -= 257;
            if (sym >= 29) return INFL_ECODES;
            int lb, le, db, de;
            len_code(sym, lb, le);
            const int len = lb + (int)getbits(b, le);
            const int ds = decode(b, dc, dr);
            if (ds < 0 || ds >= 30) return INFL_EINPUT;
            dist_code(ds, db, de);
            const int64_t dist = db + (int64_t)getbits(b, de);
            if (b.err) return INFL_EINPUT;
            if (dist > o) return INFL_EDIST;                 // BGZF members never reference outside themselves
            if (o + len > cap) return INFL_EOUTPUT;
            tok[nt++] = tok_pack(o, len, dist);              // at most cap / 3 of them: a match is >= 3 bytes
            o += len;
        }
    }
}

// members: cdata_off/cdata_len (compressed payload inside cbuf), uoff (offset in the inflated stream), isize.
// tok_off[m]: first token slot of member m (isize / 3 + 1 slots each); ntok[m]: tokens written.
__global__ __launch_bounds__(INFL_BLOCK) void k_bgzf_decode(const uint8_t *__restrict__ cbuf,
                                                           const int64_t *__restrict__ cdata_off,
                                                           const int32_t *__restrict__ cdata_len,
                                                           const int64_t *__restrict__ uoff,
                                                           const int32_t *__restrict__ isize, int64_t nmem,
                                                           uint8_t *__restrict__ data, int32_t *__restrict__ status,
                                                           uint64_t *__restrict__ tokens, const int64_t *__restrict__ tok_off,
                                                           int32_t *__restrict__ ntok) {
    __shared__ uint16_t t_lcnt[(MAXBITS + 1) * INFL_BLOCK], t_lsym[FIXLCODES * INFL_BLOCK];
    __shared__ uint16_t t_dcnt[(MAXBITS + 1) * INFL_BLOCK], t_dsym[32 * INFL_BLOCK];
    const int lane = threadIdx.x;
    const int64_t m = (int64_t)blockIdx.x * INFL_BLOCK + lane;
    if (m >= nmem) return;
    Huff lc{t_lcnt + lane, t_lsym + lane}, dc{t_dcnt + lane, t_dsym + lane};
    BitIn b;
    bit_init(b, cbuf + cdata_off[m], cbuf + cdata_off[m] + cdata_len[m]);
    uint64_t *tok = tokens + tok_off[m];
    int nt = 0;
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
            st = inflate_codes(b, lc, dc, out, o, cap, tok, nt);
            if (st) break;
        } else if (type == 2) {                              // dynamic codes
            const int nlen = (int)getbits(b, 5) + 257, ndist = (int)getbits(b, 5) + 1, ncode = (int)getbits(b, 4) + 4;
            if (b.err || nlen > MAXLCODES || ndist > MAXDCODES) { st = INFL_ECODES; break; }
            int idx = 0;
            for (; idx < ncode; ++idx) lengths[CLORDER[idx]] = (uint8_t)getbits(b, 3);
            for (; idx < 19; ++idx) lengths[CLORDER[idx]] = 0;
            if (construct(lc, lengths, 19) != 0) { st = INFL_ECODES; break; }     // code-length code must be complete
            idx = 0;
            HuffCnt cr;
            load_counts(lc, cr);
            while (idx < nlen + ndist) {
                int sym = decode(b, lc, cr);
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
            st = inflate_codes(b, lc, dc, out, o, cap, tok, nt);
            if (st) break;
        } else { st = INFL_EBLOCK; break; }
    } while (!last);
    if (st == INFL_OK && o != cap) st = INFL_ESIZE;          // ISIZE of the member trailer must match
    status[m] = st;
    ntok[m] = st == INFL_OK ? nt : 0;
}

// ---- phase B: resolve the match tokens, one wave per member -------------------------------------------
// Source bytes are read with device-scope loads of aligned words (they bypass the CU's vector L1, which may
// hold a line from before another lane of this wave wrote into it).
__device__ __forceinline__ uint32_t ld_word(const uint8_t *base, int64_t byte_off) {       // aligned word that holds the byte
    return __hip_atomic_load(reinterpret_cast<const uint32_t *>(base) + (byte_off >> 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint8_t ld_byte(const uint8_t *base, int64_t byte_off) {
    return (uint8_t)(ld_word(base, byte_off) >> ((byte_off & 3) * 8));
}
__device__ __forceinline__ uint64_t ld_u64(const uint8_t *base, int64_t byte_off) {          // 8 bytes at any offset: three words
    const uint32_t a = ld_word(base, byte_off), b = ld_word(base, byte_off + 4), c = ld_word(base, byte_off + 8);
    const int s = (int)(byte_off & 3) * 8;
    const uint64_t lo = ((uint64_t)b << 32) | a;
    return s ? (lo >> s) | ((uint64_t)c << (64 - s)) : lo;
}
// len bytes from src to dst (offsets into base), non-overlapping in the sense src + len <= dst: 8 at a time
__device__ __forceinline__ void copy_plain(uint8_t *base, int64_t dst, int64_t src, int len) {
    int j = 0;
    for (; j + 8 <= len; j += 8) *reinterpret_cast<uint64_u *>(base + dst + j) = ld_u64(base, src + j);
    if (j < len) {
        uint64_t t = ld_u64(base, src + j);
        for (; j < len; ++j) { base[dst + j] = (uint8_t)t; t >>= 8; }
    }
}

constexpr int COPY_BLOCK = 256;
__global__ __launch_bounds__(COPY_BLOCK) void k_bgzf_copy(const int64_t *__restrict__ uoff, int64_t nmem,
                                                         uint8_t *__restrict__ data,          // 4-byte aligned, readable 12 bytes past the end
                                                         const uint64_t *__restrict__ tokens, const int64_t *__restrict__ tok_off,
                                                         const int32_t *__restrict__ ntok) {
    const int lane = threadIdx.x & 63;
    const int64_t m = ((int64_t)blockIdx.x * COPY_BLOCK + threadIdx.x) >> 6;
    if (m >= nmem) return;
    const int nt = ntok[m];
    const uint64_t *T = tokens + tok_off[m];
    const int64_t ub = uoff[m];                            // member's offset in the stream
    for (int b0 = 0; b0 < nt; b0 += 64) {
        const int i = b0 + lane;
        const bool have = i < nt;
        const uint64_t t = have ? T[i] : 0ull;
        const int64_t dst = (int64_t)(t & 0x1FFFFu), dist = (int64_t)(t >> 26);
        const int len = (int)((t >> 17) & 0x1FFu);
        const int64_t src = dst - dist;
        const int64_t need = dist < len ? dist : len;      // source bytes that must be final: [src, src + need)
        const int64_t D0 = (int64_t)(T[b0] & 0x1FFFFu);    // first output byte of the batch (same address for all lanes)
        const bool indep = have && src + need <= D0;
        if (indep) {                                       // copied by its own lane
            if (dist >= len) copy_plain(data, ub + dst, ub + src, len);
            else for (int j = 0; j < len; ++j) data[ub + dst + j] = ld_byte(data, ub + src + j % dist);   // run replication
        }
        unsigned long long dep = __ballot(have && !indep);
        while (dep) {                                      // in order, each one as a wave-wide gather
            __threadfence_block();                         // this wave's earlier stores are at the L2 before these loads
            const int l = __ffsll(dep) - 1;
            dep &= dep - 1;
            const int64_t d_l = __shfl((int)dst, l, 64), k_l = __shfl((int)dist, l, 64);
            const int n_l = __shfl(len, l, 64);
            for (int j = lane; j < n_l; j += 64) data[ub + d_l + j] = ld_byte(data, ub + d_l - k_l + (k_l < n_l ? j % k_l : j));
        }
        __threadfence_block();                             // the next batch may read what this one wrote
    }
}

}  // namespace fx
