// fx_inflate_par.hpp -- BGZF decode with the parallelism INSIDE a member: one WAVE per member (round 3).
//
// k_bgzf_decode (fx_inflate.hpp) gives every member to one lane: a member is then ONE serial chain of ~12 k symbols,
// 47 k members are under one wave per SIMD, and the kernel's time is that chain's latency (18 ms for the 3 Gbp file).
// Here the 64 lanes of a wave share one member and its Huffman tables (in LDS, built once per block by the wave) and start
// decoding at 64 evenly spaced BIT positions of the block.  Only lane 0 starts at a symbol boundary; the others start
// anywhere -- and fall into step with the true sequence of symbols after a handful of them (Huffman codes
// self-synchronise; measured on zlib streams of genome text: half of the lanes within 6 symbols, all within 133 of the
// ~190 a lane decodes; a wrong path that meets an end-of-block code just goes on).  Three phases per block:
//   A   every lane decodes from its start S_k to the next lane's start, counting output bytes (T_k) and remembering the
//       end-of-block codes it met; E_k = where it landed beyond S_k+1.
//   A2  hand-over: lane k walks on from E_k (a TRUE symbol boundary if lane k's path was true at its end) side by side
//       with lane k+1's path from S_k+1 -- always advancing the one that is behind -- until the two meet: Y_k+1, the first
//       boundary they share.  From there on lane k+1's path is the true one.  By induction from lane 0 every Y is true,
//       and lane k OWNS the symbols in [Y_k, Y_k+1): T_k minus what lane k counted before Y_k plus what it walked beyond E_k.
//       The first lane whose own stretch holds an end-of-block code ends the block; lanes behind it own nothing.
//   B   exclusive prefix sum of the owned byte counts -> where every lane's output begins; the lanes decode their own
//       stretch again and this time store: literals at their final place, a match as its 3-byte token where it begins
//       plus its bit in the member's match map -- the format k_bgzf_copy resolves, unchanged.
// Blocks follow each other (header parse by lane 0, table construction by the wave); a non-final block is handled the
// same way (the lanes behind its end decode bits of the next header: never owned); stored blocks are copied by the wave.
// Anything out of the ordinary -- no meeting point inside a lane's stretch, more end-of-block codes on a path than are
// remembered, a sub-table pool that overflows -- hands the member to the serial kernel (status INFL_RETRY), which also
// produces the definitive error codes for damaged members.
#pragma once
#include "fx_inflate.hpp"

#if !defined(FX_BGZF_GLOBAL_MAP) && !defined(FX_BGZF_LDS_MAP)
#define FX_BGZF_REG_MAP 1                                               // the word of the match map a lane is in waits in a register (see P_MAP_SET)
#endif

namespace fx {

// root bits; sub-table entries (an overflow hands the member over).  Round 6: the pools sized for what zlib's trees need (genome
// text: none at all; FASTQ quality strings, protein: under 200 entries) instead of the worst case -- 384 + 256 -- and the arrays
// only the table construction uses share their bytes with the arrays only the hand-over uses: 12 784 -> 9 760 B per wave,
// sixteen waves per CU instead of twelve (FX_BGZF_P_LB / _LPOOL / _DPOOL: experiment builds)
#ifndef FX_BGZF_P_LB
#define FX_BGZF_P_LB 10
#endif
#ifndef FX_BGZF_P_LPOOL
#define FX_BGZF_P_LPOOL 384
#endif
#ifndef FX_BGZF_P_DPOOL
#define FX_BGZF_P_DPOOL 256
#endif
constexpr int P_LB = FX_BGZF_P_LB, P_DB = 8, P_LPOOL = FX_BGZF_P_LPOOL, P_DPOOL = FX_BGZF_P_DPOOL;
constexpr int INFL_RETRY = 100;
constexpr uint32_t P_MINCH = 2048;                                     // bits per lane at least, ~150 symbols: a wrong start needs up to ~130 to fall into step (short blocks use fewer lanes)
constexpr int P_MAXSKIP = 3;                                           // stretches a walk may cross without meeting their lane's path

// Table entries are 32-bit and carry everything the symbol loop needs, so that it does no arithmetic on symbol numbers:
//   bits 0-3   L: length of the code (sub-table entries: of the WHOLE code)        bits 4-7   number of extra bits
//   bits 8-9   literal/length table: 0 literal, 1 length, 2 end of block, 3 no such code;  distance table: bit 8 = no such code
//   bit 15     LINK (root entries): codes longer than the root index -> sub-table of 2^(bits 0-3) entries at pool[bits 16-31]
//   bits 16-31 the literal byte / the base of the length (3..258) or distance (1..24577)
constexpr uint32_t P_LINK = 1u << 15, P_INV_L = 1u | (3u << 8), P_INV_D = 1u | (1u << 8);
struct PTab {
    uint32_t llut[1 << P_LB], dlut[1 << P_DB], lpool[P_LPOOL], dpool[P_DPOOL];
    int hdr[8];                                                         // lane 0 -> wave: status, nlen, ndist, position behind the header, type, last
    union {
        struct {                                                        // live from the block header to the end of the table construction
            uint16_t code[MAXLCODES + MAXDCODES + 4];                   // canonical code of every symbol, bit-reversed
            uint16_t longs[MAXLCODES + 4];                              // symbols whose code is longer than the root index
            uint16_t cl[128];                                           // the code-length code: 7-bit root, complete
            uint8_t lengths[MAXLCODES + MAXDCODES + 8];
            int cnt[16];
            int tmp[32];                                                // p_header's small arrays (private arrays would be selected out of ~40 registers)
        };
        struct {                                                        // live from the end of phase A2 to the start of phase B
            uint32_t hY[64], hc[64], oY[64], oc[64], hn[64], on[64];  // hand-over: what lane k found for its target / what lane t was given (position, bytes, symbols)
            int htgt[64], own[64];
        };
    };
#ifdef FX_BGZF_LDS_MAP
    uint32_t map[2048];                                                // the member's match map (one bit per output byte), flushed once
#endif
};
#ifdef FX_BGZF_WPE
#define P_WPE __attribute__((amdgpu_waves_per_eu(FX_BGZF_WPE, FX_BGZF_WPE)))
#else
#define P_WPE
#endif

// >= 57 bits of the payload from bit position bitpos on.  STAGE: the payload sits in LDS (three aligned words and two
// funnel shifts: an unaligned 8-byte LDS read is split into byte reads by the compiler); else in place, one unaligned load.
template <bool STAGE> __device__ __forceinline__ uint64_t p_peek(const uint8_t *base, uint32_t bitpos) {
    if (STAGE) {
        const uint32_t *W = reinterpret_cast<const uint32_t *>(base) + (bitpos >> 5);
        const uint32_t a = W[0], b = W[1], c = W[2], sh = bitpos & 31u;
        return ((uint64_t)__builtin_amdgcn_alignbit(c, b, sh) << 32) | __builtin_amdgcn_alignbit(b, a, sh);
    }
    return *reinterpret_cast<const uint64_u *>(base + (bitpos >> 3)) >> (bitpos & 7u);
}

template <bool DIST> __device__ __forceinline__ uint32_t p_entry(int s, int L) {
    int base, ext;
    if (DIST) {
        if (s >= 30) return P_INV_D;
        dist_code(s, base, ext);
        return (uint32_t)L | ((uint32_t)ext << 4) | ((uint32_t)base << 16);
    }
    if (s < 256) return (uint32_t)L | ((uint32_t)s << 16);
    if (s == 256) return (uint32_t)L | (2u << 8);
    if (s >= 257 + 29) return P_INV_L;
    len_code(s - 257, base, ext);
    return (uint32_t)L | ((uint32_t)ext << 4) | (1u << 8) | ((uint32_t)base << 16);
}

// Sequential reader for phase B, where the symbol loop also STORES: loads and stores of a wave complete in issue order and
// share one counter, so a load per symbol would wait for the store of the symbol before, every time.  Here a lane fetches
// its stretch 16 bytes at a time -- one load per ~8 symbols -- and the chunk behind is requested when the current one
// starts being used: `buf` holds `cnt` valid bits, q0..q3 the 32-bit words of the current chunk not yet in buf (qn of
// them), `nx` the next chunk.  pr_fill32 moves one word into buf whenever there is room (cnt <= 32): a symbol is read in
// two parts, its literal/length code with the extra bits (<= 20 bits of >= 33) and its distance code with the extra bits
// (<= 28 bits of >= 45 after a second fill).
typedef uint32_t p_v4u __attribute__((ext_vector_type(4)));
typedef p_v4u __attribute__((aligned(1))) p_v4u_u;
struct PRd { const uint8_t *base; uint64_t buf; p_v4u nx; uint32_t q0, q1, q2, q3, p; int cnt, qn; };
__device__ __forceinline__ void pr_init(PRd &r, const uint8_t *base, uint32_t bitpos) {
    r.base = base;
    r.p = (bitpos >> 3) & ~3u;                               // chunks start at a word of the payload
    const p_v4u c = *reinterpret_cast<const p_v4u_u *>(base + r.p);
    r.p += 16u;
    r.nx = *reinterpret_cast<const p_v4u_u *>(base + r.p);
    const uint32_t skip = bitpos - ((bitpos >> 3) & ~3u) * 8u;   // 0..31
    r.buf = (uint64_t)(c.x >> skip);
    r.cnt = 32 - (int)skip;
    r.q0 = c.y; r.q1 = c.z; r.q2 = c.w; r.q3 = 0; r.qn = 3;
}
__device__ __forceinline__ void pr_fill32(PRd &r) {
    if (r.cnt <= 32) {
        if (r.qn == 0) {                                     // the next chunk becomes the current one; the one behind it is requested
            r.q0 = r.nx.x; r.q1 = r.nx.y; r.q2 = r.nx.z; r.q3 = r.nx.w; r.qn = 4;
            r.p += 16u;
            r.nx = *reinterpret_cast<const p_v4u_u *>(r.base + r.p);
        }
        r.buf |= (uint64_t)r.q0 << r.cnt;
        r.cnt += 32;
        r.q0 = r.q1; r.q1 = r.q2; r.q2 = r.q3; --r.qn;
    }
}
__device__ __forceinline__ void pr_skip(PRd &r, uint32_t nbits) { r.buf >>= nbits; r.cnt -= (int)nbits; }

// ---- table construction by the wave.  lengths[0, n) in LDS; lut: 1 << bits entries; returns < 0 over-subscribed,
// > 0 incomplete, -1000 when the pool is too small.
template <bool DIST> __device__ __forceinline__ int p_construct(PTab &T, uint32_t *lut, uint32_t *pool, int pool_cap, int bits, const uint8_t *length, int n, int lane) {
    const int size = 1 << bits;
    const uint32_t inv = DIST ? P_INV_D : P_INV_L;
    if (lane < 16) T.cnt[lane] = 0;
    __syncthreads();
    for (int s = lane; s < n; s += 64) { const int L = length[s]; if (L) atomicAdd(&T.cnt[L], 1); }
    for (int i = lane; i < size; i += 64) lut[i] = inv;
    __syncthreads();
    int left = 1, used = 0;
    for (int len = 1; len <= MAXBITS; ++len) { left = (left << 1) - T.cnt[len]; used += T.cnt[len]; if (left < 0) return left; }
    if (used == 0) return 0;
    // canonical codes, length by length: a symbol's code = first code of its length + its rank among the symbols of that length
    uint32_t first = 0;
    int nlong = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int len = 1; len <= MAXBITS; ++len) {
        const int c = T.cnt[len];
        if (c) {
            uint32_t run = 0;
            for (int s0 = 0; s0 < n; s0 += 64) {
                const int s = s0 + lane;
                const bool mine = s < n && length[s] == len;
                const unsigned long long b = __ballot(mine);
                if (mine) {
                    const uint32_t code = first + run + (uint32_t)__popcll(b & below);
                    const uint32_t rev = __brev(code) >> (32 - len);     // the stream carries codes most significant bit first
                    T.code[s] = (uint16_t)rev;
                    if (len <= bits) {
                        const uint32_t e = p_entry<DIST>(s, len);
                        for (uint32_t i = rev; i < (uint32_t)size; i += 1u << len) lut[i] = e;
                    } else T.longs[nlong + __popcll(b & below)] = (uint16_t)s;
                }
                run += (uint32_t)__popcll(b);
                if (len > bits) nlong += (int)__popcll(b);
            }
        }
        first = (first + (uint32_t)c) << 1;
    }
    if (nlong == 0) return left;
    __syncthreads();
    if (lane == 0)                                             // how many further bits do the codes under a root entry need? (few symbols: serial)
        for (int i = 0; i < nlong; ++i) {
            const int s = T.longs[i], L = length[s];
            const uint32_t p = T.code[s] & (uint32_t)(size - 1);
            const uint32_t need = (uint32_t)(L - bits), e = lut[p];
            if (!(e & P_LINK) || need > (e & 15u)) lut[p] = P_LINK | need;
        }
    __syncthreads();
    int pool_used = 0;                                         // a sub-table per such root entry: sizes scanned over the wave
    for (int p0 = 0; p0 < size; p0 += 64) {
        const uint32_t e = lut[p0 + lane];
        const uint32_t sz = (e & P_LINK) ? 1u << (e & 15u) : 0u;
        const uint32_t incl = wave_incl_scan(sz);
        const int total = __shfl((int)incl, 63, 64);
        if (pool_used + total > pool_cap) return -1000;
        if (sz) {
            const uint32_t off = (uint32_t)pool_used + incl - sz;
            lut[p0 + lane] = P_LINK | (off << 16) | (e & 15u);
            for (uint32_t i = 0; i < sz; ++i) pool[off + i] = inv;
        }
        pool_used += total;
    }
    __syncthreads();
    for (int i = lane; i < nlong; i += 64) {                   // the long codes into their sub-tables (distinct symbols, distinct entries)
        const int s = T.longs[i], L = length[s];
        const uint32_t rev = T.code[s];
        const uint32_t e = lut[rev & (uint32_t)(size - 1)];
        const int k = (int)(e & 15u), rest = L - bits;
        const uint32_t off = e >> 16;
        const uint32_t v = p_entry<DIST>(s, L);
        for (uint32_t j = rev >> bits; j < (1u << k); j += 1u << rest) pool[off + j] = v;
    }
    __syncthreads();
    return left;
}

// ---- one symbol of the literal/length + distance codes.  kind: 0 literal, 1 match, 2 end of block, 3 no such code (a
// wrong path: skip a bit; the true path: a damaged member).  Straight-line: the distance look-up runs for literals too (its
// result is dropped) -- the lanes of a wave hold both kinds at almost every step, and a branch would cost more than the
// look-up.  Only the sub-tables of long codes (rare symbols) sit behind a wave-wide test.
struct PSym { uint32_t nbits, out, kind, val; };
struct PLit { uint32_t used, mlen, kind, base; };
__device__ __forceinline__ PLit p_litlen(const PTab &T, uint32_t w32) {       // >= 20 valid bits
    uint32_t e = T.llut[w32 & ((1u << P_LB) - 1u)];
    if (__ballot((e & P_LINK) != 0u)) { if (e & P_LINK) e = T.lpool[(e >> 16) + __builtin_amdgcn_ubfe(w32, P_LB, e & 15u)]; }
    const uint32_t L = e & 15u, le = (e >> 4) & 15u;
    PLit r;
    r.kind = (e >> 8) & 3u; r.base = e >> 16;
    r.mlen = r.base + __builtin_amdgcn_ubfe(w32, L, le);
    r.used = L + le;
    return r;
}
template <bool FULL> __device__ __forceinline__ PSym p_finish(const PTab &T, const PLit &l, uint32_t v) {       // v: >= 28 valid bits behind the length
    uint32_t d = T.dlut[v & ((1u << P_DB) - 1u)];
    if (__ballot((d & P_LINK) != 0u)) { if (d & P_LINK) d = T.dpool[(d >> 16) + __builtin_amdgcn_ubfe(v, P_DB, d & 15u)]; }
    const uint32_t dl = d & 15u, de = (d >> 4) & 15u;
    const bool match = l.kind == 1u, dbad = match && (d & 0x100u) != 0u;
    PSym r;
    r.nbits = match && !dbad ? l.used + dl + de : l.used;       // (no such distance code: the length code alone, nothing stored)
    r.out = dbad ? 0u : (match ? l.mlen : (l.kind == 0u ? 1u : 0u));
    r.kind = dbad ? 3u : l.kind;
    r.val = 0;
    if (FULL) {
        const uint32_t dist = (d >> 16) + __builtin_amdgcn_ubfe(v, dl, de);
        r.val = match ? (l.mlen - 3u) | ((dist - 1u) << 8) : l.base;
    }
    return r;
}
template <bool FULL> __device__ __forceinline__ PSym p_symbol(const PTab &T, uint64_t w) {     // from >= 57 bits at once (p_peek)
    const PLit l = p_litlen(T, (uint32_t)w);
    return p_finish<FULL>(T, l, (uint32_t)(w >> l.used));
}
// the next symbol of a sequential reader, consumed
template <bool FULL> __device__ __forceinline__ PSym p_next(const PTab &T, PRd &r) {
    pr_fill32(r);
    const PLit l = p_litlen(T, (uint32_t)r.buf);
    pr_skip(r, l.used);
    pr_fill32(r);
    const PSym s = p_finish<FULL>(T, l, (uint32_t)r.buf);
    pr_skip(r, s.nbits - l.used);
    return s;
}

// the block header at bit position hp, by lane 0: type, last, and for a dynamic block the code lengths into T.lengths.
// T.hdr = {status, nlen, ndist, position behind the header, type, last}
// (Round 6, tried: the header's bytes staged in LDS by the wave before lane 0 walks them -- 10.38 -> 10.55 ms for C4: the walk's loads
// hit the L1, they follow each other through a few lines; the staging's own load and barrier per block cost more than they save.)
template <bool STAGE> __device__ __forceinline__ void p_header(PTab &T, const uint8_t *base, uint32_t hp, uint32_t pend) {
    int st = INFL_OK, nlen = 0, ndist = 0;
    uint64_t w = p_peek<STAGE>(base, hp);
    const int last = (int)(w & 1u), type = (int)((w >> 1) & 3u);
    hp += 3;
    if (hp > pend) st = INFL_EINPUT;
    else if (type == 3) st = INFL_EBLOCK;
    else if (type == 2) {
        w = p_peek<STAGE>(base, hp);
        nlen = (int)(w & 31u) + 257; ndist = (int)((w >> 5) & 31u) + 1;
        const int ncode = (int)((w >> 10) & 15u) + 4;
        hp += 14;
        if (nlen > MAXLCODES || ndist > MAXDCODES) st = INFL_ECODES;
        else {
            int *cll = T.tmp;                                  // [19]; count and next: T.cnt[0..7], T.cnt[8..15]
            for (int i = 0; i < 19; ++i) cll[i] = 0;
            w = p_peek<STAGE>(base, hp);                              // 19 x 3 = 57 bits
            for (int i = 0; i < ncode; ++i) cll[CLORDER[i]] = (int)((w >> (3 * i)) & 7u);
            hp += 3 * (uint32_t)ncode;
            // the code-length code: at most 7 bits, must be complete
            int *count = T.cnt, left = 1;
            for (int i = 0; i < 8; ++i) count[i] = 0;
            for (int i = 0; i < 19; ++i) count[cll[i]]++;
            for (int len = 1; len <= 7; ++len) { left = (left << 1) - count[len]; }
            if (left != 0) st = INFL_ECODES;
            else {
                uint32_t *next = reinterpret_cast<uint32_t *>(T.cnt + 8);
                next[1] = 0;
                for (int len = 1; len < 7; ++len) next[len + 1] = (next[len] + (uint32_t)count[len]) << 1;
                for (int s = 0; s < 19; ++s) {
                    const int L = cll[s];
                    if (!L) continue;
                    const uint32_t rev = __brev(next[L]++) >> (32 - L);
                    for (uint32_t i = rev; i < 128u; i += 1u << L) T.cl[i] = (uint16_t)((s << 4) | L);
                }
                int idx = 0;
                while (idx < nlen + ndist && st == INFL_OK) {
                    w = p_peek<STAGE>(base, hp);
                    const uint32_t e = T.cl[(uint32_t)w & 127u];
                    const int L = (int)(e & 15u), sym = (int)(e >> 4);
                    w >>= L; hp += (uint32_t)L;
                    if (sym < 16) T.lengths[idx++] = (uint8_t)sym;
                    else {
                        int len = 0, rep;
                        if (sym == 16) {
                            if (idx == 0) { st = INFL_ECODES; break; }
                            len = T.lengths[idx - 1]; rep = 3 + (int)(w & 3u); hp += 2;
                        } else if (sym == 17) { rep = 3 + (int)(w & 7u); hp += 3; }
                        else { rep = 11 + (int)(w & 127u); hp += 7; }
                        if (idx + rep > nlen + ndist) { st = INFL_ECODES; break; }
                        while (rep--) T.lengths[idx++] = (uint8_t)len;
                    }
                    if (hp > pend) st = INFL_EINPUT;
                }
                if (st == INFL_OK && T.lengths[256] == 0) st = INFL_ECODES;
            }
        }
    }
    T.hdr[0] = st; T.hdr[1] = nlen; T.hdr[2] = ndist; T.hdr[3] = (int)hp; T.hdr[4] = type; T.hdr[5] = last;
}

// REPLAY: phases A and A2 leave every symbol they decode in a scratch row (row i of a wave = the i-th symbol of each of its
// 64 lanes: one coalesced 256-byte store per step), and phase B does not decode again -- it replays its own rows.  The
// scratch belongs to the WAVE, not to the member (the grid is as many waves as the device holds at once, every wave takes
// members m, m + grid, ...): sym_rows rows of 64 words each, a member with a lane that needs more is handed over.
// the bit of output byte o in the member's match map.  Round 3-4: in LDS and flushed once per member (FX_BGZF_LDS_MAP: 8 KiB of the
// wave's 21, seven waves per CU, 12.45 ms for C4); straight in memory (FX_BGZF_GLOBAL_MAP: 13 KiB, twelve waves per CU, but 460 M
// atomics on memory: 20.3 ms); now in a register of the lane that writes that stretch of output (13 KiB, twelve waves per CU,
// one store per map word: 11.3-12.1 ms; capped at 6 / 8 / 10 waves per CU, FX_BGZF_WAVES_PER_CU: 15.2 / 12.6 / 11.6 -- the
// kernel stops gaining at ten: from there it is bound by the instructions it issues)
#if defined(FX_BGZF_REG_MAP)
// FX_BGZF_REG_MAP: a lane's matches come in rising order of their place, so the 64-bit word of the map it is in can wait in a
// register and go out when the lane moves on to the next one -- a plain store for a word that lies inside the lane's own
// stretch of output (nobody else has bits in it; the host cleared the map), an atomic for the word at either end of it
__device__ __forceinline__ void p_map_flush(unsigned long long *bm, uint32_t wi, unsigned long long mw, uint32_t o_start, uint32_t o_end) {
    if (wi == ~0u || !mw) return;
#ifndef FX_BGZF_REG_MAP_ATOMIC                                           // (an atomic for every word: 11.4-12.7 ms for C4 against 11.3-12.1)
    if (wi * 64u >= o_start && wi * 64u + 64u <= o_end) bm[wi] = mw;
    else
#endif
    atomicOr(&bm[wi], mw);
}
#define P_MAP_SET(o) do { const uint32_t w_ = (o) >> 6; if (w_ != mw_i) { p_map_flush(bm, mw_i, mw, o_first, o_end); mw = 0; mw_i = w_; } mw |= 1ull << ((o) & 63u); } while (0)
#elif defined(FX_BGZF_LDS_MAP)
#define P_MAP_SET(o) atomicOr(&T.map[(o) >> 5], 1u << ((o) & 31u))
#else
#define P_MAP_SET(o) atomicOr(reinterpret_cast<unsigned int *>(bm) + ((o) >> 5), 1u << ((o) & 31u))
#endif
template <bool STAGE, bool REPLAY>
__global__ __launch_bounds__(64) P_WPE void k_bgzf_decode_par(const uint8_t *__restrict__ cbuf, const int64_t *__restrict__ cdata_off,
                                                         const int32_t *__restrict__ cdata_len, const int64_t *__restrict__ uoff,
                                                         const int32_t *__restrict__ isize, int64_t nmem, uint8_t *__restrict__ data,
                                                         int32_t *__restrict__ status, uint64_t *__restrict__ match_map, int dbg, int lds_payload,
                                                         int32_t *__restrict__ par_status, uint32_t *__restrict__ symbuf, int sym_rows, int *next_member) {
    extern __shared__ __attribute__((aligned(16))) uint8_t p_smem[];
    PTab &T = *reinterpret_cast<PTab *>(p_smem);
    const int lane = threadIdx.x;
    uint32_t *const sb = REPLAY ? symbuf + (size_t)blockIdx.x * (size_t)sym_rows * 64u + (uint32_t)lane : nullptr;   // this lane's column
  // The members are handed out by a counter (next_member, zeroed by the host), not m, m + grid, ...: 46 723 members over 2 560 waves are
  // 18.25 each -- with fixed shares a quarter of the waves decodes a 19th member while the others are through, and a wave whose
  // members happen to be slow holds the kernel alone at the end.  (nullptr: the fixed shares.)
  for (int64_t mi = blockIdx.x;; mi += gridDim.x) {
    __syncthreads();                                         // (the tables of the member before are done with)
    int64_t m = mi;
    if (next_member) {
        if (lane == 0) T.hdr[6] = atomicAdd(next_member, 1);
        __syncthreads();
        m = T.hdr[6];
    }
    if (m >= nmem) break;
#ifdef FX_BGZF_LDS_MAP
    for (int i = lane; i < 2048; i += 64) T.map[i] = 0;
#endif
    const uint8_t *gbase = cbuf + cdata_off[m];
    const uint32_t pend = (uint32_t)cdata_len[m] * 8u;       // the payload in bits (< 2^19)
    const uint8_t *base = gbase;
    bool too_big = false;
    if (STAGE) {
        // the payload through the texture path ONCE, coalesced, into LDS; the symbol loops then peek from there
        uint8_t *const sbuf = p_smem + ((sizeof(PTab) + 15) & ~(size_t)15);
        too_big = (int)(pend >> 3) + 16 > lds_payload;       // larger than the launch provided for
        if (!too_big)
            for (uint32_t i = (uint32_t)lane * 16u; i < (pend >> 3) + 16u; i += 1024u)
                *reinterpret_cast<uint4 *>(sbuf + i) = *reinterpret_cast<const uint4_u *>(gbase + i);
        base = sbuf;
        __syncthreads();
    }
    uint8_t *out = data + uoff[m];
    unsigned long long *bm = reinterpret_cast<unsigned long long *>(match_map + m * BM_WORDS);
    const uint32_t cap = (uint32_t)isize[m];
    uint32_t obase = 0, hp = 0;                              // output bytes so far, bit position of the next block header
    int st = too_big ? INFL_RETRY + 9 : INFL_OK;
    for (; !too_big;) {
        if (lane == 0) p_header<STAGE>(T, base, hp, pend);
        __syncthreads();
        st = T.hdr[0];
        if (st) break;
        const int type = T.hdr[4], last = T.hdr[5];
        uint32_t p0 = (uint32_t)T.hdr[3];
        if (type == 0) {                                     // stored: to the byte boundary, LEN, NLEN, the bytes
            const uint32_t bp = (p0 + 7u) >> 3;
            if (bp * 8u + 32u > pend) { st = INFL_EINPUT; break; }
            const uint32_t len = base[bp] | ((uint32_t)base[bp + 1] << 8), nl = base[bp + 2] | ((uint32_t)base[bp + 3] << 8);
            if ((len ^ 0xFFFFu) != nl) { st = INFL_EBLOCK; break; }
            if ((bp + 4u + len) * 8u > pend) { st = INFL_EINPUT; break; }
            if (obase + len > cap) { st = INFL_EOUTPUT; break; }
            for (uint32_t i = lane; i < len; i += 64) out[obase + i] = base[bp + 4 + i];
            obase += len;
            hp = (bp + 4u + len) * 8u;
            __syncthreads();
            if (last) break;
            continue;
        }
        if (type == 1) {                                     // fixed codes
            for (int s = lane; s < FIXLCODES; s += 64) T.lengths[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            if (lane < MAXDCODES) T.lengths[FIXLCODES + lane] = 5;
            __syncthreads();
            p_construct<false>(T, T.llut, T.lpool, P_LPOOL, P_LB, T.lengths, FIXLCODES, lane);
            p_construct<true>(T, T.dlut, T.dpool, P_DPOOL, P_DB, T.lengths + FIXLCODES, MAXDCODES, lane);
        } else {
            const int nlen = T.hdr[1], ndist = T.hdr[2];
            int nz = 0;
            for (int s = lane; s < nlen; s += 64) nz += T.lengths[s] != 0;
            nz = (int)wave_sum((uint32_t)nz);
            int err = p_construct<false>(T, T.llut, T.lpool, P_LPOOL, P_LB, T.lengths, nlen, lane);
            if (err == -1000) { st = INFL_RETRY + 1; break; }
            if (err < 0 || (err > 0 && nz != 1)) { st = INFL_ECODES; break; }
            nz = 0;
            for (int s = lane; s < ndist; s += 64) nz += T.lengths[nlen + s] != 0;
            nz = (int)wave_sum((uint32_t)nz);
            err = p_construct<true>(T, T.dlut, T.dpool, P_DPOOL, P_DB, T.lengths + nlen, ndist, lane);
            if (err == -1000) { st = INFL_RETRY + 2; break; }
            if (err < 0 || (err > 0 && nz != 1)) { st = INFL_ECODES; break; }
        }
        // ---- the symbols of this block: [p0, end of block), somewhere in [p0, pend)
        if (p0 >= pend) { st = INFL_EINPUT; break; }
        const uint32_t rem = pend - p0;
        uint32_t ch = (rem + 63u) / 64u;
        ch = ch < P_MINCH ? P_MINCH : ch;
        const uint32_t S = p0 + (uint32_t)lane * ch, Sn = S + ch;          // this lane's stretch [S, Sn)
        const bool active = S < pend;
        if (dbg == 8) break;
        // phase A
        uint32_t pos = S, T_k = 0, ns = 0;                                    // ns: symbols this lane has decoded (= rows it has written)
        uint32_t e1p = ~0u, e1a = 0, e1b = 0, e1i = 0, e2p = ~0u, e2a = 0, e2b = 0, e2i = 0;   // end-of-block codes met: position, position behind, bytes before, row
        int neob = 0;
        if (active) {
            const uint32_t lim = Sn < pend ? Sn : pend;
#ifndef FX_BGZF_NO_A_CHUNK
            PRd ra;
            if (!STAGE) pr_init(ra, base, S);
#endif
            while (pos < lim) {
#ifndef FX_BGZF_NO_A_CHUNK
                const PSym s = STAGE ? p_symbol<REPLAY>(T, p_peek<STAGE>(base, pos)) : p_next<REPLAY>(T, ra);
#else
                const PSym s = p_symbol<REPLAY>(T, p_peek<STAGE>(base, pos));
#endif
                if (REPLAY) { if (ns < (uint32_t)sym_rows) sb[(size_t)ns * 64u] = (s.kind << 30) | s.val; }
                if (s.kind == 2) {
                    if (neob == 0) { e1p = pos; e1a = pos + s.nbits; e1b = T_k; e1i = ns; }
                    else if (neob == 1) { e2p = pos; e2a = pos + s.nbits; e2b = T_k; e2i = ns; }
                    ++neob;
                }
                pos += s.nbits; T_k += s.out; ++ns;
            }
        }
        const uint32_t E = pos;
        if (dbg == 1) { if (E + T_k + neob == 0xFFFFFFFFu) T.hdr[7] = 1; break; }      // timing probes (FX_BGZF_DBG): wrong answers
        // phase A2: walk on from E side by side with the next lane's path from its start until they meet.  A walk that crosses
        // the whole next stretch without meeting its lane's path leaves that lane out (it owns nothing) and tries the one behind.
        uint32_t p = E, ovb = 0, cq = 0, oep = ~0u, oea = 0, oeb = 0, oei = 0;   // oe*: the first end-of-block code on the walk from E
        uint32_t pn = 0, qn = 0;                                                 // symbols of the walk from E / of the other lane's path up to the meeting point
        int tgt = 64;                                                            // the lane this one hands over to
        if (active) {
            int t = lane + 1;
            uint32_t St = Sn;
            for (int skipped = 0; t < 64 && St < pend && skipped <= P_MAXSKIP && oep == ~0u; ++t, St += ch, ++skipped) {
                const uint32_t lim = (St + ch) < pend ? St + ch : pend;          // the meeting point must lie inside lane t's stretch
                uint32_t q = St;
                cq = 0; qn = 0;
                while (p != q && p < lim && oep == ~0u) {
                    const bool adv_p = p < q || q >= lim;                        // (the other path has left the stretch: only an end-of-block code on this one still matters)
                    const uint32_t at = adv_p ? p : q;
                    const PSym s = p_symbol<REPLAY>(T, p_peek<STAGE>(base, at));
                    if (adv_p) {
                        if (REPLAY) { if (ns + pn < (uint32_t)sym_rows) sb[(size_t)(ns + pn) * 64u] = (s.kind << 30) | s.val; }
                        if (s.kind == 2) { oep = p; oea = p + s.nbits; oeb = ovb; oei = ns + pn; }
                        p += s.nbits; ovb += s.out; ++pn;
                    } else { q += s.nbits; cq += s.out; ++qn; }
                }
                if (p == q && oep == ~0u) { tgt = t; break; }
            }
        }
        if (dbg == 2) { if (p + ovb + cq == 0xFFFFFFFFu) T.hdr[7] = 1; break; }
        // hand-over: the chain of owners from lane 0 on (lane 0 follows it through LDS: a few dozen steps), every owner learns
        // where its own stretch begins (Y) and what it counted before that (c)
        T.htgt[lane] = tgt; T.hY[lane] = p; T.hc[lane] = cq; T.hn[lane] = qn; T.own[lane] = 0;
        __syncthreads();
        if (lane == 0) {
            int k = 0;
            T.oY[0] = p0; T.oc[0] = 0; T.on[0] = 0;
            for (int guard = 0; guard < 64; ++guard) {
                T.own[k] = 1;
                const int t = T.htgt[k];
                if (t >= 64) break;
                T.oY[t] = T.hY[k]; T.oc[t] = T.hc[k]; T.on[t] = T.hn[k];
                k = t;
            }
        }
        __syncthreads();
        const bool owner = active && T.own[lane] != 0;
        const uint32_t Y = T.oY[lane], c = T.oc[lane], Yn = p, skipn = T.on[lane];      // skipn: rows of this lane that lie in front of Y
        // the first end-of-block code at or behind Y on this lane's own path, else on its walk beyond E
        uint32_t eob_after = 0, eob_bytes = 0, eob_row = 0;
        bool has_eob = false, ambiguous = false;
        if (owner) {
            if (e1p != ~0u && e1p >= Y) { has_eob = true; eob_after = e1a; eob_bytes = e1b - c; eob_row = e1i; }
            else if (e2p != ~0u && e2p >= Y) { has_eob = true; eob_after = e2a; eob_bytes = e2b - c; eob_row = e2i; }
            else if (neob > 2) ambiguous = true;
            else if (oep != ~0u) { has_eob = true; eob_after = oea; eob_bytes = (T_k - c) + oeb; eob_row = oei; }
        }
        if (__ballot(ambiguous)) {                           // more end-of-block codes on the way than were remembered: the own stretch once more, from Y
            if (ambiguous) {
                uint32_t bp = Y, nb = 0, nr = skipn;
                while (bp < E) {
                    const PSym s = p_symbol<false>(T, p_peek<STAGE>(base, bp));
                    if (s.kind == 2) { has_eob = true; eob_after = bp + s.nbits; eob_bytes = nb; eob_row = nr; break; }
                    bp += s.nbits; nb += s.out; ++nr;
                }
                if (!has_eob && oep != ~0u) { has_eob = true; eob_after = oea; eob_bytes = (T_k - c) + oeb; eob_row = oei; }
            }
        }
        const unsigned long long eb = __ballot(has_eob);
        const int j = eb ? __ffsll((long long)eb) - 1 : 64;                      // the owner that ends the block
        if (j == 64) { st = INFL_RETRY + (__ballot(owner && tgt == 64) ? 4 : 5); break; }   // a walk gave up / no end of block inside the payload
        const uint32_t n_k = !owner ? 0u : lane < j ? (T_k - c) + ovb : (lane == j ? eob_bytes : 0u);
        const uint32_t incl = wave_incl_scan(n_k);
        const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
        if (obase + total > cap) { st = INFL_EOUTPUT; break; }
        if (REPLAY && __ballot(owner && lane <= j && ns + pn > (uint32_t)sym_rows)) { st = INFL_RETRY + 6; break; }   // a lane with more symbols than rows
        // phase B: the own stretch again, stored this time.  Every byte of the stretch's output is written, 8 at a time from a
        // register (the bytes of a match behind its token are k_bgzf_copy's to fill: zeros here) -- whole words instead of one
        // partial store per symbol, and no line of the output is left half-written for the copy kernel to merge.
        int bad = 0;
        if (owner && lane <= j) {
            uint32_t o = obase + incl - n_k;
            const uint32_t o_end = o + n_k;
#ifdef FX_BGZF_REG_MAP
            const uint32_t o_first = o;
            unsigned long long mw = 0;
            uint32_t mw_i = ~0u;
#endif
            const uint32_t stop = lane < j ? Yn : eob_after;                     // (lane j stops AT its end-of-block code, see below)
            uint32_t bp = Y;
            uint64_t acc = 0;                                                    // the bytes [o - fill, o)
            uint32_t fill = 0;
            if (REPLAY) {
                // the rows [skipn, r1) of this lane's column, four at a time (their loads together, then the byte work and the stores)
                const uint32_t r1 = lane < j ? ns + pn : eob_row;
                for (uint32_t i = skipn; i < r1 && !bad; i += 4) {
                    uint32_t e[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) e[u] = i + u < r1 ? sb[(size_t)(i + u) * 64u] : (2u << 30);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t kind = e[u] >> 30, val = e[u] & 0x3FFFFFFFu;
                        if (kind == 2u) continue;                                // (past the end of the column; an end-of-block row is never inside [skipn, r1))
                        if (kind == 3u) { bad = INFL_EINPUT; break; }
                        const uint32_t outn = kind ? (val & 0xFFu) + 3u : 1u;
                        if (o + outn > o_end) { bad = INFL_ESIZE; break; }
                        if (kind) {
                            if ((val >> 8) + 1u > o) { bad = INFL_EDIST; break; }
                            P_MAP_SET(o);
                        }
                        const uint64_t v = kind ? (uint64_t)(val & 0xFFFFFFu) : (uint64_t)(val & 0xFFu);
                        acc |= v << (8u * fill);
                        const uint64_t spill = fill > 5u ? v >> (8u * (8u - fill)) : 0ull;
                        fill += outn;
                        o += outn;
                        if (fill >= 8u) {
                            *reinterpret_cast<uint64_u *>(out + (o - fill)) = acc;
                            acc = spill; fill -= 8u;
                            while (fill >= 8u) { *reinterpret_cast<uint64_u *>(out + (o - fill)) = acc; acc = 0; fill -= 8u; }
                        }
                    }
                }
                bp = stop;
            }
            PRd r;
            if (!STAGE && !REPLAY) pr_init(r, base, Y);
            for (; !REPLAY;) {
                if (lane < j && bp >= stop) break;
                const PSym s = STAGE ? p_symbol<true>(T, p_peek<STAGE>(base, bp)) : p_next<true>(T, r);
                if (s.kind == 2) { bp += s.nbits; break; }
                if (s.kind == 3) { bad = INFL_EINPUT; break; }
                if (bp + s.nbits > pend) { bad = INFL_EINPUT; break; }
                if (o + s.out > o_end) { bad = INFL_ESIZE; break; }
                if (s.kind == 1) {
                    if ((s.val >> 8) + 1u > o) { bad = INFL_EDIST; break; }      // BGZF members never reference outside themselves
                    P_MAP_SET(o);
                }
                if (dbg == 4) { acc += s.val; o += s.out; bp += s.nbits; continue; }
                // literal: one byte; match: the 3-byte token, then out - 3 bytes of nothing
                const uint64_t v = s.kind == 0 ? (uint64_t)(s.val & 0xFFu) : (uint64_t)(s.val & 0xFFFFFFu);
                acc |= v << (8u * fill);
                const uint64_t spill = fill > 5u ? v >> (8u * (8u - fill)) : 0ull;   // what of a token did not fit the word
                fill += s.out;
                o += s.out; bp += s.nbits;
                if (fill >= 8u) {
                    if (dbg == 5) {                                              // timing probe: the same stores into 4 KiB per member (they stay in the L2)
                        *reinterpret_cast<uint64_u *>(out + ((o - fill) & 0xFFFu)) = acc;
                        acc = spill; fill -= 8u;
                        while (fill >= 8u) { *reinterpret_cast<uint64_u *>(out + ((o - fill) & 0xFFFu)) = acc; acc = 0; fill -= 8u; }
                        continue;
                    }
                    *reinterpret_cast<uint64_u *>(out + (o - fill)) = acc;
                    acc = spill; fill -= 8u;
                    while (fill >= 8u) { *reinterpret_cast<uint64_u *>(out + (o - fill)) = acc; acc = 0; fill -= 8u; }   // a long match
                }
            }
            for (uint32_t i = 0; i < fill; ++i) out[o - fill + i] = (uint8_t)(acc >> (8u * i));      // the last few bytes
#ifdef FX_BGZF_REG_MAP
            p_map_flush(bm, mw_i, mw, o_first, o_end);
#endif
            if (!bad && (o != o_end || bp != stop)) bad = INFL_ESIZE;            // the second walk must land where the first one did
        }
        const unsigned long long bb = __ballot(bad != 0);
        if (bb) { st = __shfl(bad, __ffsll((long long)bb) - 1, 64); break; }
        obase += total;
        hp = (uint32_t)__shfl((int)eob_after, j, 64);
        __syncthreads();
        if (last) break;
    }
    if (st == INFL_OK && obase != cap) st = INFL_ESIZE;
    __syncthreads();
#ifdef FX_BGZF_LDS_MAP
    if (st == INFL_OK) {                                     // the map out of LDS: whole words, coalesced (the buffer was cleared by the host)
        const uint32_t nw = (cap + 63u) >> 6;
        for (uint32_t i = lane; i < nw; i += 64) bm[i] = (unsigned long long)T.map[2 * i] | ((unsigned long long)T.map[2 * i + 1] << 32);
    }
#endif
    if (st != INFL_OK) {
        // whatever went wrong, the serial kernel decodes the member again (and names the error if there is one): it wants a
        // clean match map.  The status keeps the reason: INFL_RETRY + 1 / 2 sub-table pool, 3 end-of-block codes, 4 no meeting
        // point, 5 no end of block, 9 payload larger than the staging area; INFL_RETRY + 16 + code: a decode error
        for (int i = lane; i < BM_WORDS; i += 64) bm[i] = 0ull;
        if (st < INFL_RETRY) st = INFL_RETRY + 16 + st;
    }
    if (lane == 0) { status[m] = st; par_status[m] = st; }              // par_status: what THIS kernel made of the member (the serial one overwrites status)
  }
}

}  // namespace fx
