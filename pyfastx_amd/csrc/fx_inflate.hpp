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
// Three kernels.  DEFLATE decoding is bit-serial, but only the DECODING is: what a match copies does not
// influence how the following bits are parsed.  So
//   k_bgzf_decode   one work-item per member parses the bit stream: literals are stored straight to
//                   their final position; a match leaves a 3-byte token (length, distance) in the first
//                   bytes of the place it will fill and one bit in the member's match map (one bit per
//                   output byte).  No load of previously written output, so the only memory latency on
//                   the critical path is the input, and that is read one 8-byte word ahead of use.
//   k_bgzf_copy     one WAVE per member walks the map, 64 matches at a time, in rounds: everything below
//                   the first match still to do is final, so a match whose source ends there is ready;
//                   the ready matches of a round are independent of each other -- short ones are made by
//                   their own lanes, long ones by the whole wave (out[dst + j] = out[src + j % dist] has
//                   no dependency inside a match, run replication (dist < len) included).  Distances are
//                   long against 64 matches' worth of output: one or two rounds per batch.
//   k_bgzf_crc      the CRC-32 of every member against its trailer.
// This replaces the serial gzread() inflate that feeds the reference's scan (kseq.c:70) and the
// zran_seek/zran_read random access (index.c:685-686).  The decoder reads a symbol with one or two LDS look-ups (8-bit
// literal/length and 6-bit distance root tables + sub-tables per work-item, 52 KiB per 64-lane workgroup); codes in no
// table fall back to the canonical "count/symbol" walk of zlib's contrib/puff (a public algorithm, not part of the
// reference tree).  Code lengths for dynamic blocks live in private memory.  (See the note at struct Huff.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fx {

#ifndef FX_INFL_BLOCK
#define FX_INFL_BLOCK 64
#endif
constexpr int INFL_BLOCK = FX_INFL_BLOCK; // decode: members per workgroup = one wave (32 and 16 measured: 17.9 and 23.5 ms against 17.4); 832 B of LDS tables each
constexpr int MAXBITS = 15, MAXLCODES = 286, MAXDCODES = 30, FIXLCODES = 288;

enum InflStatus { INFL_OK = 0, INFL_EINPUT = 1, INFL_EOUTPUT = 2, INFL_EBLOCK = 3, INFL_ECODES = 4, INFL_EDIST = 5,
                  INFL_ESIZE = 6, INFL_ECRC = 7 };

typedef uint64_t __attribute__((aligned(1))) uint64_u;    // 8 bytes at any address (gfx9 unaligned access mode)

// Bit reader.  `next` holds the 8 bytes at the read position p, requested (one unaligned 8-byte load off the kernel
// argument) by the refill BEFORE: a refill ORs them into the bit buffer, advances p by the whole bytes that fitted and
// at once requests the 8 bytes at the new p -- which is exactly what the following refill will want, a whole symbol
// later.  No window of aligned words to rotate: the rotation's register copies made the compiler wait for the word it
// had just requested (1-2 us per 8 bytes of input), and a pointer rebuilt from an integer is a FLAT pointer whose loads
// count as LDS traffic, so every table look-up waited for the prefetch as well.  (The buffer is readable 48 bytes past
// the last member.)
struct BitIn {
    const uint8_t *base;         // the member's payload in the compressed buffer (global memory, off the kernel argument)
    uint32_t p, end;             // byte offsets in it (a member is < 64 KiB): next unconsumed byte, end of the payload
    uint64_t next;               // the 8 bytes at p
    uint64_t buf;
    int cnt;                     // bits in buf; the symbol loop lets it go negative (= input exhausted) and checks once per symbol
    int err;
};
__device__ __forceinline__ void bit_init(BitIn &b, const uint8_t *cbuf, int64_t p, int64_t end) {
    b.base = cbuf + p; b.p = 0; b.end = (uint32_t)(end - p); b.buf = 0; b.cnt = 0; b.err = 0;
    b.next = *reinterpret_cast<const uint64_u *>(b.base);
}
// tops the bit buffer up to >= 56 bits (branch-free; a refill of a full buffer moves nothing); the bits of a partially
// consumed byte are OR-ed in again by the next refill at the same position: same data, harmless
__device__ __forceinline__ void refill(BitIn &b) {
    const uint32_t avail = b.end - b.p;                                   // bytes of this member left
    uint64_t v = b.next;
    if (avail < 8) v = avail == 0 ? 0 : (v & (~0ull >> (64 - 8 * avail)));   // never feed bytes of the next member
    b.buf |= v << b.cnt;
    uint32_t take = (uint32_t)(63 - b.cnt) >> 3;
    take = take > avail ? avail : take;
    b.p += take;
    b.cnt += (int)take * 8;
    b.next = *reinterpret_cast<const uint64_u *>(b.base + b.p);
}
__device__ __forceinline__ uint32_t getbits(BitIn &b, int n) {
    if (b.cnt < n) { refill(b); if (b.cnt < n) { b.err = 1; return 0; } }
    const uint32_t v = (uint32_t)(b.buf & ((1ull << n) - 1ull));
    b.buf >>= n; b.cnt -= n;
    return v;
}

// Huffman tables of one work-item (column `lane` of the LDS arrays: element i of this lane is ptr[i * INFL_BLOCK]).
//   lut    root table: index = the next LBITS (literal/length) or DBITS (distance) bits of the stream;
//          entry = symbol << 4 | code length for codes that fit the index, LINK | offset << 4 | k for the codes that share
//          these bits and are longer (a sub-table of 2^k entries in `pool`, indexed by the following k bits, entry =
//          symbol << 4 | remaining length), 0 for unused codes and when the pool is full (-> slow path).
//   pool   POOL entries per work-item shared by the sub-tables of both codes of a block
//   sym, cnt (GLOBAL scratch, 320 + 32 words per member): symbols in canonical order and codes per length, read by the
//          slow path only -- the canonical walk of zlib's contrib/puff (a public algorithm, not part of the reference
//          tree) -- which no stream of a well-behaved compressor ever reaches.
// What the stream of a genome looks like decides the sizes: ~13 k symbols per 64 KiB member, three quarters of them
// matches (zlib takes any match it finds in four-letter text; mean length 6.5), length codes of 2-10 bits, distance
// codes of 3-12 bits + up to 13 extra bits; one dynamic block per member.  With 832 B per member every member of the
// file is resident at once (46 723 members = 731 waves on 256 CUs x 3 workgroups), which matters more than anything
// else here: a member is ONE serial chain of symbols, so the kernel's time is that chain's latency, not anybody's
// throughput -- and because 64 members share a wave, a path that ONE lane takes is paid for by all of them: a rare slow
// path (3 % of the symbols) ran in 9 of 10 iterations and made the first table-driven version slower than the bit-serial
// walk it replaced.  Hence the sub-tables: every code of a normal stream is one or two LDS reads.
// Measured on the 3.05 GB C4 file (profiles/r02_*): 32.1 ms for the round-1 kernel, of which most was the bit reader's
// state living in scratch memory behind a non-inlined call; 17 ms now.  PMC (profiles/r02_pmc_bgzf.txt): 160 VALU + 64
// SALU instructions and 3 LDS reads per symbol, 43 % of the wave cycles issuing, 53 % parked in s_waitcnt; the time moves
// with neither the instruction count (a 30 % leaner loop: the same) nor the waves per SIMD, and leaving both stores out
// takes 4 ms off -- what is left is the chain  load 8 bytes -> table -> table  of one symbol after the other.
constexpr int LBITS = 8, DBITS = 6, POOL = 96, GSYM = 320 + 32;
constexpr uint32_t LINK = 0x8000u;
struct Huff { uint16_t *lut; uint16_t *pool; uint16_t *sym; uint16_t *cnt; int bits; };     // sym, cnt: global, contiguous

// canonical walk, one code length per step: only for codes that are in no table
__device__ __forceinline__ int decode_slow(BitIn &b, const Huff &h) {
    uint32_t bits = (uint32_t)b.buf;
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= MAXBITS; ++len) {
        code |= (int)(bits & 1u);
        bits >>= 1;
        const int count = (int)h.cnt[len];
        if (code - count < first) {
            if (b.cnt < len) { b.err = 1; return -1; }
            b.buf >>= len; b.cnt -= len;
            return h.sym[index + (code - first)];
        }
        index += count; first += count;
        first <<= 1; code <<= 1;
    }
    b.err = 1;
    return -1;
}
// next symbol; the caller has made sure of >= 15 bits in the buffer (or the input is at its end)
__device__ __forceinline__ int decode(BitIn &b, const Huff &h) {
    uint32_t e = h.lut[((uint32_t)b.buf & ((1u << h.bits) - 1u)) * INFL_BLOCK];
    int len = (int)(e & 15u);
    if (e & LINK) {                                          // a longer code: the sub-table of these leading bits
        const uint32_t sub = ((uint32_t)(b.buf >> h.bits)) & ((1u << len) - 1u);
        e = h.pool[(((e >> 4) & 0x7FFu) + sub) * INFL_BLOCK];
        len = (e & 15u) ? (int)(e & 15u) + h.bits : 0;
    }
    if (__builtin_expect(len != 0, 1)) {
        if (b.cnt < len) { b.err = 1; return -1; }
        b.buf >>= len; b.cnt -= len;
        return (int)(e >> 4);
    }
    return decode_slow(b, h);
}

// Build the tables from code lengths (canonical codes).  Returns <0 for an over-subscribed set, >0 incomplete.
// pool_used: entries of the work-item's pool taken so far (the two codes of a block share it).
__device__ __noinline__ int construct(const Huff &h, const uint8_t *length, int n, int &pool_used) {
    uint16_t count[MAXBITS + 1], offs[MAXBITS + 1], next[MAXBITS + 1];
    for (int len = 0; len <= MAXBITS; ++len) count[len] = 0;
    for (int s = 0; s < n; ++s) count[length[s]]++;
    const int size = 1 << h.bits;
    for (int i = 0; i < size; ++i) h.lut[i * INFL_BLOCK] = 0;
    for (int len = 1; len <= MAXBITS; ++len) h.cnt[len] = count[len];
    if (count[0] == n) return 0;
    int left = 1;
    for (int len = 1; len <= MAXBITS; ++len) {
        left <<= 1;
        left -= count[len];
        if (left < 0) return left;
    }
    offs[1] = 0; next[1] = 0;
    for (int len = 1; len < MAXBITS; ++len) {
        offs[len + 1] = offs[len] + count[len];
        next[len + 1] = (uint16_t)((next[len] + count[len]) << 1);
    }
    bool any_long = false;
    for (int s = 0; s < n; ++s) {
        const int L = length[s];
        if (!L) continue;
        h.sym[offs[L]++] = (uint16_t)s;
        const uint32_t code = next[L]++;
        const uint32_t rev = __brev(code) >> (32 - L);       // the stream carries codes most significant bit first, bits least first
        if (L <= h.bits) {
            const uint16_t e = (uint16_t)((s << 4) | L);
            for (uint32_t i = rev; i < (uint32_t)size; i += 1u << L) h.lut[i * INFL_BLOCK] = e;
        } else {                                             // pass A: how many further bits do the codes under this root entry need?
            const uint32_t p = rev & (uint32_t)(size - 1);
            const uint32_t need = (uint32_t)(L - h.bits), have = h.lut[p * INFL_BLOCK] & 15u;
            if (need > have) h.lut[p * INFL_BLOCK] = (uint16_t)(LINK | need);
            any_long = true;
        }
    }
    if (!any_long) return left;
    for (int p = 0; p < size; ++p) {                         // pass B: a sub-table per such root entry, while the pool lasts
        const uint32_t e = h.lut[p * INFL_BLOCK];
        if (!(e & LINK)) continue;
        const int k = (int)(e & 15u);
        if (pool_used + (1 << k) > POOL) { h.lut[p * INFL_BLOCK] = 0; continue; }     // (those codes take the slow path)
        h.lut[p * INFL_BLOCK] = (uint16_t)(LINK | ((uint32_t)pool_used << 4) | (uint32_t)k);
        for (int i = 0; i < (1 << k); ++i) h.pool[(pool_used + i) * INFL_BLOCK] = 0;
        pool_used += 1 << k;
    }
    for (int len = 1; len <= MAXBITS; ++len) next[len] = 0;   // pass C: the long codes again, in the same canonical order
    next[1] = 0;
    for (int len = 1; len < MAXBITS; ++len) next[len + 1] = (uint16_t)((next[len] + count[len]) << 1);
    for (int s = 0; s < n; ++s) {
        const int L = length[s];
        if (!L) continue;
        const uint32_t code = next[L]++;
        if (L <= h.bits) continue;
        const uint32_t rev = __brev(code) >> (32 - L);
        const uint32_t e = h.lut[(rev & (uint32_t)(size - 1)) * INFL_BLOCK];
        if (!(e & LINK)) continue;
        const int k = (int)(e & 15u), rest = L - h.bits;
        const uint32_t off = (e >> 4) & 0x7FFu;
        const uint16_t v = (uint16_t)((s << 4) | rest);
        for (uint32_t i = rev >> h.bits; i < (1u << k); i += 1u << rest) h.pool[(off + i) * INFL_BLOCK] = v;
    }
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
typedef uint32_t __attribute__((aligned(1))) uint32_u;    // 4 bytes at any address
constexpr int BM_WORDS = 1024;                            // words of a member's match map: one bit per output byte of at most 64 KiB

// Literal/length + distance codes of one block: literals -> out[o...], matches -> tok[nt...].
// One refill per symbol: a literal/length code (<= 15 bits) with its extra bits (<= 5) and a distance code (<= 15)
// with its extra bits (<= 13) are 48 bits, the refill leaves >= 56.
// (Everything that touches the bit reader is force-inlined into the kernel: one call that takes `BitIn &` puts the
// reader's state into scratch memory, and then every symbol costs a dozen round trips to it -- that, not the Huffman
// walk, was the 32 ms of the first version.)
__device__ __forceinline__ int inflate_codes(BitIn &b, const Huff &lc, const Huff &dc, uint8_t *out, int64_t &o, int64_t cap,
                                             uint64_t *bm, uint64_t &bmw_io, uint32_t &bwin_io) {
    // The loop's memory traffic per symbol is ONE request for the next 8 input bytes and TWO stores (the symbol -- a literal
    // byte, or the 3-byte token of a match put where the match begins -- and the current word of the member's match map).  vmcnt counts both and the compiler, asked to wait for the input bytes, waits for "everything"
    // -- i.e. for the store of the symbol before to be acknowledged by the L2, ~2 800 cycles per symbol, which WAS the
    // kernel's time.  So the request is issued by hand (the compiler does not see it) and waited for with vmcnt(2) right
    // after the symbol's two stores: memory operations of a wave complete in issue order on gfx9, so "at most two
    // outstanding" means the request is back while the stores are still on their way.  Paths that leave the loop
    // without the stores wait for everything.
    int ret = -1;
    uint64_t cur = b.next;
    asm volatile("" : "+v"(cur));                            // (the compiler's own wait for the load behind b.next happens HERE, not at the top of every iteration)
    // the loop works on copies in registers of 32-bit quantities: positions inside the member and its output (both < 2^17)
    uint64_t buf = b.buf;
    int cnt = b.cnt;
    uint32_t pos = b.p, oo = (uint32_t)o;
    const uint32_t end = b.end, ocap = (uint32_t)cap;
    const uint8_t *const base = b.base;
    uint64_t bmw = bmw_io;                                   // bits of the 64 output bytes of window bwin that begin a match
    uint32_t bwin = bwin_io;
    uint16_t *const llut = lc.lut, *const dlut = dc.lut, *const pool = lc.pool;
    BitIn sb;                                                // what the slow path works on
    sb.err = 0;
    for (;;) {
        // refill from the 8 bytes at pos (cur: valid here), then ask for the 8 bytes at the new position
        const uint32_t avail = end - pos;
        uint64_t v = cur;
        if (__builtin_expect(avail < 8, 0)) v = avail == 0 ? 0 : (v & (~0ull >> (64 - 8 * avail)));
        buf |= v << cnt;
        uint32_t take = (uint32_t)(63 - cnt) >> 3;
        take = take > avail ? avail : take;
        pos += take;
        cnt += (int)take * 8;
        uint64_t nxt;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(nxt) : "v"(base + pos) : "memory");
        // literal / length symbol
        uint32_t e = llut[((uint32_t)buf & ((1u << LBITS) - 1u)) * INFL_BLOCK];
        int len = (int)(e & 15u);
        if (e & LINK) {
            const uint32_t sub = ((uint32_t)(buf >> LBITS)) & ((1u << len) - 1u);
            e = pool[(((e >> 4) & 0x7FFu) + sub) * INFL_BLOCK];
            len = (e & 15u) ? (int)(e & 15u) + LBITS : 0;
        }
        int sym = (int)(e >> 4);
        if (__builtin_expect(len == 0, 0)) {                 // in no table
            sb.buf = buf; sb.cnt = cnt;
            sym = decode_slow(sb, lc);
            buf = sb.buf; cnt = sb.cnt;
            if (sym < 0) { ret = INFL_EINPUT; break; }
        } else { buf >>= len; cnt -= len; }
        if (sym == 256) { ret = cnt < 0 ? INFL_EINPUT : INFL_OK; break; }
        const bool lit = sym < 256;
        uint32_t adv = 1;
        if (!lit) {
            sym -= 257;
            if (sym >= 29) { ret = INFL_ECODES; break; }
            int lb, le, db, de;
            len_code(sym, lb, le);
            const uint32_t mlen = (uint32_t)lb + ((uint32_t)buf & ((1u << le) - 1u));
            buf >>= le; cnt -= le;
            uint32_t d = dlut[((uint32_t)buf & ((1u << DBITS) - 1u)) * INFL_BLOCK];
            int dl = (int)(d & 15u);
            if (d & LINK) {
                const uint32_t sub = ((uint32_t)(buf >> DBITS)) & ((1u << dl) - 1u);
                d = pool[(((d >> 4) & 0x7FFu) + sub) * INFL_BLOCK];
                dl = (d & 15u) ? (int)(d & 15u) + DBITS : 0;
            }
            int ds = (int)(d >> 4);
            if (__builtin_expect(dl == 0, 0)) {
                sb.buf = buf; sb.cnt = cnt;
                ds = decode_slow(sb, dc);
                buf = sb.buf; cnt = sb.cnt;
                if (ds < 0) { ret = INFL_EINPUT; break; }
            } else { buf >>= dl; cnt -= dl; }
            if (ds >= 30) { ret = INFL_EINPUT; break; }
            dist_code(ds, db, de);
            const uint32_t dist = (uint32_t)db + ((uint32_t)buf & ((1u << de) - 1u));
            buf >>= de; cnt -= de;
            if (dist > oo) { ret = INFL_EDIST; break; }      // BGZF members never reference outside themselves
            sym = (int)((mlen - 3u) | ((dist - 1u) << 8));                            // the token: 8 + 15 bits (tok_len / tok_dist)
            adv = mlen;
        }
        if (cnt < 0) { ret = INFL_EINPUT; break; }           // the symbol took bits the member does not have
        if (oo + adv > ocap) { ret = INFL_EOUTPUT; break; }
        // exactly TWO stores per symbol, whatever it is (lanes of one wave decode different kinds at the same moment, and
        // a store under a branch would be one instruction per kind).  (1) Four bytes at the symbol's place: the literal,
        // or the token of the match in the first three of the bytes the match will fill (k_bgzf_copy reads it there and
        // then overwrites it); the bytes behind belong to symbols that are decoded -- and stored -- later, or to the same
        // match.  Only within four bytes of the member's end the store is done byte by byte (the next member's bytes are
        // another lane's).  (2) The word of the match map that holds this position, as far as it is known: the last store
        // to a word carries all its bits; words no symbol begins in stay zero (the map is cleared before the launch).
        const uint32_t w = oo >> 6;
        bmw = w == bwin ? bmw : 0ull;
        bwin = w;
        bmw |= lit ? 0ull : 1ull << (oo & 63u);
        if (__builtin_expect(oo + 4u <= ocap, 1)) *reinterpret_cast<uint32_u *>(out + oo) = (uint32_t)sym;
        else {
            out[oo] = (uint8_t)sym;
            if (!lit) { out[oo + 1] = (uint8_t)(sym >> 8); out[oo + 2] = (uint8_t)(sym >> 16); }
        }
        bm[w] = bmw;
        oo += adv;
        asm volatile("s_waitcnt vmcnt(2)" : "+v"(nxt) : : "memory");       // the request is back; this symbol's stores need not be
        cur = nxt;
    }
    uint64_t keep = 0;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(keep) : : "memory");           // left without the stores: everything, then read p again
    b.buf = buf; b.cnt = cnt < 0 ? 0 : cnt; b.p = pos; o = oo;
    bmw_io = bmw; bwin_io = bwin;
    b.next = *reinterpret_cast<const uint64_u *>(b.base + b.p);
    return ret;
}

// members: cdata_off/cdata_len (compressed payload inside cbuf), uoff (offset in the inflated stream), isize.
// match_map: BM_WORDS 64-bit words per member, zero on entry: bit k of word w <-> a match begins at output byte 64 w + k
// of the member (its 3-byte token sits there).  Round 1 kept the matches as a list of 8-byte tokens per member, sized
// for the worst case (isize / 3 + 1 of them): 8.2 GB of scratch for a 3 GB file, and a hipMalloc of that size that
// took anything between 1 and 750 ms.  The map is 1/8 of the output: 0.38 GB.
// gsym: GSYM words of scratch per member (the canonical symbol order of its current tables: slow path only).
__global__ __launch_bounds__(INFL_BLOCK) void k_bgzf_decode(const uint8_t *__restrict__ cbuf,
                                                           const int64_t *__restrict__ cdata_off,
                                                           const int32_t *__restrict__ cdata_len,
                                                           const int64_t *__restrict__ uoff,
                                                           const int32_t *__restrict__ isize, int64_t nmem,
                                                           uint8_t *__restrict__ data, int32_t *__restrict__ status,
                                                           uint64_t *__restrict__ match_map, uint16_t *__restrict__ gsym, int only_status) {
    __shared__ uint16_t t_llut[(1 << LBITS) * INFL_BLOCK], t_dlut[(1 << DBITS) * INFL_BLOCK], t_pool[POOL * INFL_BLOCK];
    const int lane = threadIdx.x;
    const int64_t m = (int64_t)blockIdx.x * INFL_BLOCK + lane;
    if (m >= nmem) return;
    if (only_status >= 0 && status[m] < only_status) return;          // behind k_bgzf_decode_par: only the members it handed over
    uint16_t *gs = gsym + m * GSYM;                          // 288 + 32 symbols, 16 + 16 counts
    Huff lc{t_llut + lane, t_pool + lane, gs, gs + 320, LBITS}, dc{t_dlut + lane, t_pool + lane, gs + FIXLCODES, gs + 336, DBITS};
    int pool_used = 0;
    BitIn b;
    bit_init(b, cbuf, cdata_off[m], cdata_off[m] + cdata_len[m]);
    uint64_t *bm = match_map + m * BM_WORDS;
    uint64_t bmw = 0;
    uint32_t bwin = 0;
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
            pool_used = 0;
            construct(lc, lengths, FIXLCODES, pool_used);
            for (s = 0; s < MAXDCODES; ++s) lengths[s] = 5;
            construct(dc, lengths, MAXDCODES, pool_used);
        } else if (type == 2) {                              // dynamic codes
            const int nlen = (int)getbits(b, 5) + 257, ndist = (int)getbits(b, 5) + 1, ncode = (int)getbits(b, 4) + 4;
            if (b.err || nlen > MAXLCODES || ndist > MAXDCODES) { st = INFL_ECODES; break; }
            int idx = 0;
            for (; idx < ncode; ++idx) lengths[CLORDER[idx]] = (uint8_t)getbits(b, 3);
            for (; idx < 19; ++idx) lengths[CLORDER[idx]] = 0;
            pool_used = 0;
            if (construct(lc, lengths, 19, pool_used) != 0) { st = INFL_ECODES; break; }     // code-length code must be complete
            idx = 0;
            while (idx < nlen + ndist) {
                if (b.cnt < 24) refill(b);
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
            int nz = 0;
            for (int s = 0; s < nlen; ++s) nz += lengths[s] != 0;
            pool_used = 0;
            int err = construct(lc, lengths, nlen, pool_used);
            if (err < 0 || (err > 0 && nz != 1)) { st = INFL_ECODES; break; }
            nz = 0;
            for (int s = 0; s < ndist; ++s) nz += lengths[nlen + s] != 0;
            err = construct(dc, lengths + nlen, ndist, pool_used);
            if (err < 0 || (err > 0 && nz != 1)) { st = INFL_ECODES; break; }
        } else { st = INFL_EBLOCK; break; }
        if (type != 0) {                                     // the one place the codes are decoded (see inflate_codes)
            st = inflate_codes(b, lc, dc, out, o, cap, bm, bmw, bwin);
            if (st) break;
        }
    } while (!last);
    if (st == INFL_OK && o != cap) st = INFL_ESIZE;          // ISIZE of the member trailer must match
    status[m] = st;
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

// A wave takes its member 4 KiB of output at a time: the 64 words of the match map of that stretch (one per lane) become
// the list of the positions where matches begin (in LDS, in order), and the list is worked through 64 matches at a
// time -- token from the first three bytes of the match's place (requested one batch ahead), then the copies.
// 32 bytes at any address, read at the L2 (sc0 sc1: not from the CU's vector L1, which may hold a line from before a lane
// of this wave wrote into it) with two instructions; ld_u64 above costs three loads and a funnel shift per 8 bytes, and
// this kernel is bound by the instructions it issues.  The bytes past the ones a match needs are not used (the blob is
// readable a tile past its end).  The wait is inside: the compiler does not see these loads.
__device__ __forceinline__ void ld32_l2(const uint8_t *p, uint64_t (&w)[4]) {
    uint4 a, b;
    asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
    w[0] = a.x | ((uint64_t)a.y << 32); w[1] = a.z | ((uint64_t)a.w << 32);
    w[2] = b.x | ((uint64_t)b.y << 32); w[3] = b.z | ((uint64_t)b.w << 32);
}

// the token of a match: the word at its place, any alignment, ONE request to the L2 (ld_u64 above is three)
__device__ __forceinline__ uint32_t ld_tok(const uint8_t *base, int64_t byte_off) {
    return __hip_atomic_load(reinterpret_cast<const uint32_t *>(base + byte_off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ld16_l2(const uint8_t *p, uint64_t (&w)[4]) {
    uint4 a;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(p) : "memory");
    w[0] = a.x | ((uint64_t)a.y << 32); w[1] = a.z | ((uint64_t)a.w << 32);
    w[2] = 0; w[3] = 0;
}

#ifndef FX_COPY_SW
#define FX_COPY_SW 32
#endif
constexpr int COPY_SW = FX_COPY_SW;                        // map words per stretch: 32 = 2 KiB of output (64: 4 KiB -- 27 KB of LDS per workgroup, five waves per SIMD instead of eight)
constexpr int COPY_SB = COPY_SW * 64;                      // bytes of a stretch
constexpr int COPY_BLOCK = 256, COPY_POS = COPY_SB / 3 + 2;
__global__ __launch_bounds__(COPY_BLOCK) void k_bgzf_copy(const int64_t *__restrict__ uoff, const int32_t *__restrict__ isize, int64_t nmem,
                                                         uint8_t *__restrict__ data,          // 4-byte aligned, readable a tile past the end
                                                         const uint64_t *__restrict__ match_map) {
    __shared__ uint16_t s_pos[COPY_BLOCK / 64][COPY_POS];
    // The stretch as the decode kernel left it -- literals, tokens, zeros -- is read ONCE, coalesced, before any of its matches is
    // made: the tokens then come out of LDS.  (One load per token, at 64 places per instruction, was a quarter of the requests
    // this kernel sends to the L2, and those are what bound it.)
    __shared__ __attribute__((aligned(16))) uint32_t s_tok[COPY_BLOCK / 64][COPY_SB / 4 + 4];
    const int lane = threadIdx.x & 63;
    const int64_t m = ((int64_t)blockIdx.x * COPY_BLOCK + threadIdx.x) >> 6;
    if (m >= nmem) return;                                 // waves are independent: no workgroup barrier below
    uint16_t *sp = s_pos[threadIdx.x >> 6];
    uint32_t *stok = s_tok[threadIdx.x >> 6];
    const uint64_t *B = match_map + m * BM_WORDS;
    const int64_t ub = uoff[m];                            // member's offset in the stream
    const int nwords = (isize[m] + 63) >> 6;
    uint64_t wnext = lane < COPY_SW && lane < nwords ? B[lane] : 0ull;       // the map of a stretch is requested while the stretch before is resolved
    for (int c0 = 0; c0 < nwords; c0 += COPY_SW) {
        uint64_t wd = wnext;
        wnext = lane < COPY_SW && c0 + COPY_SW + lane < nwords ? B[c0 + COPY_SW + lane] : 0ull;
        const uint32_t cnt = (uint32_t)__popcll(wd), incl = wave_incl_scan(cnt);
        const int total = __shfl((int)incl, 63, 64);
        if (total == 0) continue;
        {
            const uint8_t *sb = data + ub + ((int64_t)c0 << 6);
            // COPY_SB / 1024 x 1 KiB coalesced + the 16 bytes behind the stretch (every lane the same: one request)
            uint4 t[COPY_SB / 1024], t4;
            const uint8_t *pl = sb + lane * 16, *pt = sb + COPY_SB;
            if (COPY_SB == 4096)
                asm volatile("global_load_dwordx4 %0, %5, off sc0 sc1\n\tglobal_load_dwordx4 %1, %5, off offset:1024 sc0 sc1\n\t"
                             "global_load_dwordx4 %2, %5, off offset:2048 sc0 sc1\n\tglobal_load_dwordx4 %3, %5, off offset:3072 sc0 sc1\n\t"
                             "global_load_dwordx4 %4, %6, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[COPY_SB / 1024 - 2]), "=&v"(t[COPY_SB / 1024 - 1]), "=&v"(t4) : "v"(pl), "v"(pt) : "memory");
            else
                asm volatile("global_load_dwordx4 %0, %3, off sc0 sc1\n\tglobal_load_dwordx4 %1, %3, off offset:1024 sc0 sc1\n\t"
                             "global_load_dwordx4 %2, %4, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t4) : "v"(pl), "v"(pt) : "memory");
            uint4 *st4 = reinterpret_cast<uint4 *>(stok);
#pragma unroll
            for (int k = 0; k < COPY_SB / 1024; ++k) st4[k * 64 + lane] = t[k];
            if (lane == 0) st4[COPY_SB / 16] = t4;
        }
        uint32_t r = incl - cnt;
        while (wd) {
            const int k = __ffsll((long long)wd) - 1;
            wd &= wd - 1;
            sp[r++] = (uint16_t)(((c0 + lane) << 6) + k);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        // 64 matches at a time.  Everything below `frontier` is final: first the place of the batch's first match, then, round
        // by round, the place of its first match that is still to do (the bytes before it are literals or matches already
        // made).  A match whose source ends below the frontier is ready; the ready matches of a round do not depend on each
        // other: the short ones are made by their own lanes, all at once (source words first, then the stores), the long
        // ones one after the other by the whole wave.  The first pending match is always ready, so a batch takes as many
        // rounds as its longest chain of matches feeding matches -- one or two -- instead of one step per dependent match.
        for (int b0 = 0; b0 < total; b0 += 64) {
            const int i = b0 + lane;
            const bool have = i < total;
            const int64_t dst = have ? sp[i] : 0;
            uint32_t tk = 0;                                   // 8 + 15 bits in the first three bytes of the match's place
            if (have) {
                const uint32_t rel = (uint32_t)dst - ((uint32_t)c0 << 6);
                const uint32_t wq = rel >> 2, sh = (rel & 3u) * 8u;
                tk = sh ? __funnelshift_r(stok[wq], stok[wq + 1], sh) : stok[wq];
            }
            const int len = (int)(tk & 0xFFu) + 3;
            const int64_t dist = (int64_t)((tk >> 8) & 0x7FFFu) + 1;
            const int64_t src = dst - dist;
            const int64_t need = dist < len ? dist : len;      // source bytes that must be final: [src, src + need)
            int64_t frontier = sp[b0];
            bool pending = have;
            for (;;) {
                const bool ready = pending && src + need <= frontier;
                const bool wide = ready && len > 32;
                if (ready && !wide) {
                    uint8_t *o = data + ub + dst;
                    if (dist >= len || dist == 1) {
                        // the bytes at the source -- 16 in one load when that is all the match takes (most matches of genome text are
                        // short), else 32 in two; a run of ONE byte (dist 1: N runs, poly-A) is that byte sixteen times
                        uint64_t w[4];
                        if (dist == 1) { const uint64_t v = (uint64_t)ld_byte(data, ub + src) * 0x0101010101010101ull; w[0] = w[1] = w[2] = w[3] = v; }
                        else if (len <= 16) ld16_l2(data + ub + src, w);
                        else ld32_l2(data + ub + src, w);
                        // as few stores as the length allows: the head of the match, then its LAST 16 / 8 / 4 bytes once more,
                        // overlapping the head where they must
                        if (len >= 16) {
                            *reinterpret_cast<uint4_u *>(o) = make_uint4((uint32_t)w[0], (uint32_t)(w[0] >> 32), (uint32_t)w[1], (uint32_t)(w[1] >> 32));
                            if (len > 16) {
                                const int sh = len - 16;                       // 1 .. 16: the 16 bytes from byte sh on
                                const int q = sh >> 3, rr = (sh & 7) * 8;
                                const uint64_t a0 = q == 0 ? w[0] : q == 1 ? w[1] : w[2], a1 = q == 0 ? w[1] : q == 1 ? w[2] : w[3],
                                               a2 = q == 0 ? w[2] : q == 1 ? w[3] : 0ull;           // (selects: an indexed array would live in scratch)
                                const uint64_t t0 = rr ? (a0 >> rr) | (a1 << (64 - rr)) : a0, t1 = rr ? (a1 >> rr) | (a2 << (64 - rr)) : a1;
                                *reinterpret_cast<uint4_u *>(o + sh) = make_uint4((uint32_t)t0, (uint32_t)(t0 >> 32), (uint32_t)t1, (uint32_t)(t1 >> 32));
                            }
                        } else if (len >= 8) {
                            *reinterpret_cast<uint64_u *>(o) = w[0];
                            if (len > 8) { const int rr = (len - 8) * 8; *reinterpret_cast<uint64_u *>(o + len - 8) = (w[0] >> rr) | (w[1] << (64 - rr)); }
                        } else if (len >= 4) {
                            *reinterpret_cast<uint32_u *>(o) = (uint32_t)w[0];
                            if (len > 4) *reinterpret_cast<uint32_u *>(o + len - 4) = (uint32_t)(w[0] >> ((len - 4) * 8));
                        } else {                                               // 3 bytes: a match is never shorter
                            *reinterpret_cast<uint16_u *>(o) = (uint16_t)w[0];
                            o[2] = (uint8_t)(w[0] >> 16);
                        }
                    } else for (int j = 0; j < len; ++j) o[j] = ld_byte(data, ub + src + j % dist);   // a short period: byte by byte
                }
                unsigned long long wb = __ballot(wide);
                while (wb) {                                   // each long one by the whole wave, 16 bytes per lane where the source allows
                    const int l = __ffsll(wb) - 1;
                    wb &= wb - 1;
                    const int64_t d_l = __shfl((int)dst, l, 64), k_l = __shfl((int)dist, l, 64);
                    const int n_l = __shfl(len, l, 64);
                    uint8_t *o = data + ub + d_l;
                    if (k_l == 1 || k_l >= n_l) {              // one byte repeated, or a source that ends in front of the place
                        int j = lane * 16;
                        if (j < n_l) {
                            if (j + 16 > n_l) j = n_l - 16;    // the last piece once more, overlapping (n_l > 32)
                            uint4 v;
                            if (k_l == 1) { const uint32_t c = (uint32_t)ld_byte(data, ub + d_l - 1) * 0x01010101u; v = make_uint4(c, c, c, c); }
                            else { uint64_t w[4]; ld16_l2(o - k_l + j, w); v = make_uint4((uint32_t)w[0], (uint32_t)(w[0] >> 32), (uint32_t)w[1], (uint32_t)(w[1] >> 32)); }
                            *reinterpret_cast<uint4_u *>(o + j) = v;
                        }
                    } else for (int j = lane; j < n_l; j += 64) o[j] = ld_byte(data, ub + d_l - k_l + j % k_l);
                }
                pending = pending && !ready;
                __threadfence_block();                         // this wave's stores are at the L2 before the next round's (or batch's) loads
                const unsigned long long pb = __ballot(pending);
                if (!pb) break;
                frontier = __shfl((int)dst, __ffsll(pb) - 1, 64);
            }
        }
    }
}

// ---- phase C: the CRC-32 of every member against its trailer (zlib checks it in the reference's gzread; a member that
// inflates to the right length with wrong bytes must not go unnoticed).  One wave per member, 1 KiB of it per step,
// read the way everything else reads the stream: 64 lanes x 16 contiguous bytes (the first form gave every lane its own
// 1 KiB piece, 4 bytes at a time -- 32 x the member in L2 traffic, 12 ms).  CRCs are linear:
//     crc_0(A || B) = shift_{|B|}(crc_0(A)) ^ crc_0(B)        crc(D) = ~(crc_0(D) ^ shift_{|D|}(~0))
// so a lane's 16-byte CRC (init 0) meets its neighbours' in a butterfly of six steps (shift by 16, 32, ... 512 bytes),
// rows are chained with shift by 1 KiB, the member is right-aligned to the rows (zeros in front of init-0 data change
// nothing) and the all-ones start value is shifted by the member's length once, at the end.  "Shift by 2^k bytes" is a
// 32 x 32 matrix over GF(2) (computed on the host by repeated squaring, as zlib's crc32_combine does); the seven the hot
// loop uses are expanded into 4 x 256-entry tables (one LDS look-up per byte of the operand).
// Round 4: 64 bytes per lane and row (4 KiB rows) instead of 16: the butterfly of six steps -- 24 of the 40 table look-ups a
// lane made per 16 bytes -- is paid once per 64, and a lane's own bytes go through four tables at a time ("slicing by 4":
// four independent look-ups per word instead of a chain of four): 1.26 -> see DESIGN.md for C4.
constexpr int CRC_LB = 64, CRC_ROW = 64 * CRC_LB, CRC_NSH = 7, CRC_SH0 = 6;   // shifts by CRC_LB << k bytes, k = 0 .. 6 (6 = one row); 2^CRC_SH0 = CRC_LB
struct CrcTables {
    uint32_t crc[4][256];                                    // crc[0]: the byte table of the reflected polynomial 0xEDB88320; crc[j]: the same j bytes further on
    uint32_t sh[CRC_NSH][4][256];                            // sh[k][b][v] = shift_{CRC_LB << k}(v << 8 b)
    uint32_t pow2[17][32];                                   // matrix of "shift by 2^j bytes", j = 0 .. 16 (column i = image of bit i)
};
__device__ __forceinline__ uint32_t crc_shift(const uint32_t (*t)[256], uint32_t v) {
    return t[0][v & 0xFFu] ^ t[1][(v >> 8) & 0xFFu] ^ t[2][(v >> 16) & 0xFFu] ^ t[3][v >> 24];
}
__global__ __launch_bounds__(256) void k_bgzf_crc(const uint8_t *__restrict__ data, const int64_t *__restrict__ uoff,
                                                 const int32_t *__restrict__ isize, const uint8_t *__restrict__ cbuf,
                                                 const int64_t *__restrict__ cdata_off, const int32_t *__restrict__ cdata_len,
                                                 int64_t nmem, const CrcTables *__restrict__ T, int32_t *__restrict__ status) {
    __shared__ uint32_t tab[4][256], sh[CRC_NSH][4][256];
    for (int i = threadIdx.x; i < 4 * 256; i += 256) (&tab[0][0])[i] = (&T->crc[0][0])[i];
    for (int i = threadIdx.x; i < CRC_NSH * 4 * 256; i += 256) (&sh[0][0][0])[i] = (&T->sh[0][0][0])[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t m = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (m >= nmem) return;
    const int64_t n = isize[m];
    const uint8_t *p = data + uoff[m];
    const int64_t nrows = (n + CRC_ROW - 1) / CRC_ROW;
    uint32_t acc = 0;                                        // crc_0 of the rows so far (wave-uniform)
    for (int64_t r = 0; r < nrows; ++r) {
        const int64_t a = n - (nrows - r) * CRC_ROW + lane * CRC_LB;   // member-relative offset of this lane's 64 bytes (< 0 in the first row: nothing there)
        uint32_t c = 0;
        if (a >= 0) {
            uint4 v[CRC_LB / 16];
#pragma unroll
            for (int q = 0; q < CRC_LB / 16; ++q) v[q] = *reinterpret_cast<const uint4_u *>(p + a + 16 * q);
#pragma unroll
            for (int q = 0; q < CRC_LB / 16; ++q) {
                const uint32_t w[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    c ^= w[k];
                    c = tab[3][c & 0xFFu] ^ tab[2][(c >> 8) & 0xFFu] ^ tab[1][(c >> 16) & 0xFFu] ^ tab[0][c >> 24];
                }
            }
        } else if (a > -CRC_LB) {                            // the member begins inside this lane's piece: its bytes one by one
            for (int k = (int)-a; k < CRC_LB; ++k) c = tab[0][(c ^ p[a + k]) & 0xFFu] ^ (c >> 8);
        }
#pragma unroll
        for (int l = 0; l < 6; ++l) {                        // butterfly: after step l every lane holds the crc_0 of its block of 2^(l+1) lanes
            const uint32_t o = (uint32_t)__shfl_xor((int)c, 1 << l, 64);
            const bool left = ((lane >> l) & 1) == 0;
            c = crc_shift(sh[l], left ? c : o) ^ (left ? o : c);
        }
        acc = crc_shift(sh[6], acc) ^ c;
    }
    if (lane == 0 && status[m] == INFL_OK) {
        uint32_t init = 0xFFFFFFFFu;                         // shift_n(~0): n in binary, one stored matrix per set bit
        for (int j = 0; j < 17; ++j)
            if ((n >> j) & 1) { uint32_t v = 0; for (int b = 0; b < 32; ++b) v ^= ((init >> b) & 1u) ? T->pow2[j][b] : 0u; init = v; }
        const uint32_t got = n ? ~(acc ^ init) : 0u;
        const uint8_t *t = cbuf + cdata_off[m] + cdata_len[m];
        const uint32_t want = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        if (want != got) status[m] = INFL_ECRC;
    }
}

}  // namespace fx
