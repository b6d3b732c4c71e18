// fx_comp.hpp -- per-record letter composition of a FASTA stream (fasta.c:901-950) on gfx950.
//
// The reference adds one to seq_comp[byte] for every byte of every sequence line (fasta.c:922-926) and writes
// the non-zero bins of each record to the `comp` table.  What bounds that on a GPU is not HBM but the VALU: a
// histogram needs a decision per BYTE, and the straightforward forms (one SWAR compare + popcount per letter, or
// one LDS atomic per byte) cost 15-20 instructions per byte -- 2.2 ms for 3 GB, a quarter of what the memory
// system can deliver.  This kernel spends ~4.5 instructions per byte:
//
//   classify   four bytes at a time, no compares: the 3-bit code (b >> 1) & 7 is distinct for A C G T N \n \r
//              (and for their lower-case forms, which differ only in bit 5), so ONE v_perm_b32 with the code as
//              selector looks up a one-hot class byte per input byte (A=1 C=2 G=4 T=8 N=16 \r=32, \n=0), a second
//              v_perm looks up the byte the code stands for; x ^ case-bit ^ expected != 0 flags every byte that
//              is none of the 13 expected ones (IUPAC codes, protein letters, '*', ...), exactly.
//   count      the one-hot words are never popcounted one by one: they are summed bit-plane-wise with carry-save
//              adders (Harley-Seal; a full adder is two v_bitop3_b32), 16 words per 4 KiB granule and lane, into
//              9 planes that hold, for every bit position, the number of words that had it set (up to 256).  A
//              second set of planes counts the lower-case letters (one-hot & lower-case byte mask).
//   flush      only when the wave moves on to another record or is done (normally once per 64 KiB): per class and
//              plane one v_and + one v_dot4_u32_u8 (weights 2^plane), a wave reduction, 11 global atomics.
//
// One wave walks up to 16 consecutive granules.  hdr_prefix (headers before each granule, from the index build)
// says without reading anything whether a granule lies inside one record's sequence block -- then the whole
// granule is counted unmasked.  Granules that hold header lines, the record's first granule and the tail of the
// stream are cut into (record, byte range) segments from the record table and counted with the bytes outside the
// range replaced by '\n'.  Bytes that are none of the expected ones do not slow the main path down: the planes count
// them as whatever class their 3-bit code aliases to, and -- only when a lane has seen one -- a compact second
// pass over that granule (re-read from L2) adds them to a per-wave LDS histogram and takes the alias back out.
// Semantics as before: '\n' not counted, '\r' counted, header lines and bytes before the first header skipped,
// bytes >= 128 ignored (the reference indexes a 128-entry table with them: undefined behaviour).
#pragma once
#include "fx_kernels.hpp"

#ifndef FX_GRAN
#define FX_GRAN 4096
#endif

namespace fx {

constexpr int COMP_GPW = 32;                    // granules per wave at most: 16 words per lane and granule -> 512 per flush at most
constexpr int COMP_NPL = 10;                    // planes 2^0 .. 2^9
constexpr int64_t COMP_SMALL = 2048;            // records of at most this many bytes are counted by k_fasta_comp_small
#ifndef FX_COMP_DEPTH
#define FX_COMP_DEPTH 2
#endif
constexpr int COMP_DEPTH = FX_COMP_DEPTH;       // granules in flight per wave in the PURE kernel (gpw is a multiple of it)

struct CompPlanes { uint32_t p[COMP_NPL]; };

// full adder on 32 independent bit positions: a + b + c = 2 * carry + sum
__device__ __forceinline__ void csa(uint32_t &carry, uint32_t &sum, uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t cy = __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8);     // majority
    sum = __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);                   // parity
    carry = cy;
}

// Harley-Seal over the 16 words of a granule, four at a time: after words 4j .. 4j+3 the partial carries wait in
// f[] (fours) and e[] (eights); planes_finish16 folds them into the planes
struct CompCarry { uint32_t f[2], e[2]; };
__device__ __forceinline__ void planes_add4(CompPlanes &P, CompCarry &c, int j, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    uint32_t tA, tB;
    csa(tA, P.p[0], P.p[0], w0, w1);
    csa(tB, P.p[0], P.p[0], w2, w3);
    csa(c.f[j & 1], P.p[1], P.p[1], tA, tB);
    if (j & 1) csa(c.e[j >> 1], P.p[2], P.p[2], c.f[0], c.f[1]);
}
__device__ __forceinline__ void planes_finish16(CompPlanes &P, const CompCarry &c) {
    uint32_t s;
    csa(s, P.p[3], P.p[3], c.e[0], c.e[1]);     // s: positions where sixteen more have been seen
#pragma unroll
    for (int k = 4; k < COMP_NPL; ++k) { const uint32_t t = P.p[k] & s; P.p[k] ^= s; s = t; }
}

// number of bytes whose class bit `cls` was set, over everything added to P by this lane
__device__ __forceinline__ uint32_t planes_count(const CompPlanes &P, int cls) {
    const uint32_t m = 0x01010101u << cls;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_udot4(P.p[k] & m, 0x01010101u << k, acc, false);
#pragma unroll
    for (int k = 8; k < COMP_NPL; ++k) acc += __builtin_amdgcn_udot4(P.p[k] & m, 0x01010101u, 0u, false) << k;
    return acc >> cls;
}

// v_perm_b32 looks a byte up in the 8-byte table {hi, lo} (selector 0..3 -> bytes of lo, 4..7 -> bytes of hi)
//   code = (b >> 1) & 7:   A a -> 0   C c -> 1   T t -> 2   G g -> 3   (4 unused)   \n -> 5   \r -> 6   N n -> 7
constexpr uint32_t OH_LO = 0x04080201u, OH_HI = 0x10200000u;     // one-hot class: A 1, C 2, G 4, T 8, N 16, \r 32, \n 0
constexpr uint32_t EX_LO = 0x47544341u, EX_HI = 0x4E0D0A80u;     // the upper-case byte the code stands for (0x80: none)
constexpr int CLS_A = 0, CLS_C = 1, CLS_G = 2, CLS_T = 3, CLS_N = 4, CLS_CR = 5;

// h: one-hot class of the 4 bytes of x; hl: the same for lower-case letters only; d: != 0 in every byte that is
// none of A C G T N a c g t n \n \r (for those bytes h / hl hold the class the code aliases to)
__device__ __forceinline__ void comp_classify(uint32_t x, uint32_t &h, uint32_t &hl, uint32_t &d) {
    const uint32_t s1 = x >> 1;
    const uint32_t code = s1 & 0x07070707u;
    h = __builtin_amdgcn_perm(OH_HI, OH_LO, code);
    const uint32_t e = __builtin_amdgcn_perm(EX_HI, EX_LO, code);
    const uint32_t q = x & s1 & 0x20202020u;    // bit 5 of the bytes that have bits 5 and 6 set: lower-case letters
    const uint32_t low = (q << 3) - (q >> 5);   // 0xFF in those bytes
    hl = h & low;
    d = __builtin_amdgcn_bitop3_b32(x, q, e, 0x96);     // x ^ q ^ e: the byte with its case bit cleared against the expected one
}

// bytes [lo, hi) of a word (0 <= lo, hi may be anything): 0xFF in the bytes kept
__device__ __forceinline__ uint32_t word_range_mask(int lo, int hi) {
    lo = lo < 0 ? 0 : lo;
    hi = hi > 4 ? 4 : hi;
    if (hi <= lo) return 0u;
    const uint32_t a = 0xFFFFFFFFu << (8 * lo);
    const uint32_t b = 0xFFFFFFFFu >> (8 * (4 - hi));
    return a & b;
}

struct CompState {
    CompPlanes all, low;
    int64_t rec;                                // record the planes belong to (-1: none)
    bool rare;                                  // the wave's LDS histogram holds something
};

__device__ __forceinline__ void comp_reset(CompState &s) {
#pragma unroll
    for (int k = 0; k < COMP_NPL; ++k) { s.all.p[k] = 0; s.low.p[k] = 0; }
}

// flush the lane counters of record s.rec into comp[rec][*]
__device__ __forceinline__ uint32_t wave_total(uint32_t v) {      // sum over the 64 lanes, wave-uniform (DPP scan + readlane)
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v), 63);
}
__device__ __forceinline__ uint32_t comp_symbol(int slot) {    // slot c: upper-case letter of class c (5: \r), slot 8 + c: lower case
    const int c = slot & 7;
    return (c == 0 ? 'A' : c == 1 ? 'C' : c == 2 ? 'G' : c == 3 ? 'T' : c == 4 ? 'N' : 13u) | (slot >= 8 ? 0x20u : 0u);
}
// Record indices: 0 .. n_hdr-1, and -1 = the bytes of the shard that precede its first header line (they belong to a
// record of an earlier shard; counted into row n_hdr from offset lead_from on when lead_from >= 0, see fx_fasta_comp).
// COMP_NONE = no record.
constexpr int64_t COMP_NONE = -2;
// blk_cnt != null: counts of record blk_rec are gathered per workgroup in LDS (slots as above) and reach comp with
// one set of atomics per workgroup -- thousands of waves adding to the same few cache lines of one chromosome's row
// serialise in L2 otherwise
__device__ __forceinline__ void comp_flush(CompState &s, uint32_t *__restrict__ rare_hist, unsigned long long *__restrict__ comp,
                                           uint32_t *__restrict__ blk_cnt, int64_t blk_rec, int64_t n_hdr) {
    if (s.rec != COMP_NONE) {
        const int lane = lane_id();
        unsigned long long *row = comp + (s.rec >= 0 ? s.rec : n_hdr) * 128;
        uint32_t mine = 0;                      // lane c: upper-case count of class c, lane 8 + c: lower-case count
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const uint32_t tot = planes_count(s.all, c);
            const uint32_t lo = c < 5 ? planes_count(s.low, c) : 0u;
            const uint32_t nu = wave_total(tot - lo);
            if (lane == c) mine = nu;
            if (c < 5) {
                const uint32_t nl = wave_total(lo);
                if (lane == 8 + c) mine = nl;
            }
            __builtin_amdgcn_sched_barrier(0);  // class by class: computed side by side they only cost registers
        }
        if (mine) {
            if (blk_cnt && s.rec == blk_rec) atomicAdd(&blk_cnt[lane], mine);
            else atomicAdd(&row[comp_symbol(lane)], (unsigned long long)mine);
        }
        if (s.rare) {                           // wave-uniform; the bins are signed: aliases were taken out of them
            for (int b = lane; b < 128; b += 64) {
                const int32_t v = (int32_t)rare_hist[b];
                if (v) { atomicAdd(&row[b], (unsigned long long)(long long)v); rare_hist[b] = 0; }
            }
            s.rare = false;
        }
    }
    comp_reset(s);
}

// count the bytes [a, b) (granule-relative) of the granule held in v[4] (lane l, load j: bytes (j*64 + l)*16 ..) into s;
// FULL: the whole granule, no masking.  Returns != 0 in the lanes that met a byte outside the expected set.
template <bool FULL>
__device__ __forceinline__ uint32_t comp_add_granule(CompState &s, const uint4 (&v)[4], int a, int b, uint32_t *drow = nullptr) {
    const int lane = lane_id();
    uint32_t dacc = 0;                          // (drow, optional: the same per load j, i.e. per KiB row of the granule)
    CompCarry ca, cl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t xs[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
        uint32_t tA, tB, uA, uB;                // twos of the first / second pair of words: all letters, lower case
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            uint32_t h[2], hl[2], dd[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                uint32_t x = xs[2 * pr + i];
                if (!FULL) {
                    const int off = (j * 64 + lane) * 16 + 4 * (2 * pr + i);
                    const uint32_t m = word_range_mask(a - off, b - off);
                    x = (x & m) | (0x0A0A0A0Au & ~m);
                }
                comp_classify(x, h[i], hl[i], dd[i]);
            }
            dacc = __builtin_amdgcn_bitop3_b32(dacc, dd[0], dd[1], 0xFE);       // dacc | d0 | d1
            csa(pr ? tB : tA, s.all.p[0], s.all.p[0], h[0], h[1]);
            csa(pr ? uB : uA, s.low.p[0], s.low.p[0], hl[0], hl[1]);
            // two words at a time: anything wider only costs registers.  The empty asm pins dacc: otherwise the
            // compiler turns the OR-reduction into sixteen compares at the end of the granule and keeps the
            // operands of all sixteen alive until then
            asm volatile("" : "+v"(dacc));
            if (FULL) __builtin_amdgcn_sched_barrier(0);
        }
        if (drow) { drow[j] = dacc; dacc = 0; }
        csa(ca.f[j & 1], s.all.p[1], s.all.p[1], tA, tB);
        csa(cl.f[j & 1], s.low.p[1], s.low.p[1], uA, uB);
        if (j & 1) {
            csa(ca.e[j >> 1], s.all.p[2], s.all.p[2], ca.f[0], ca.f[1]);
            csa(cl.e[j >> 1], s.low.p[2], s.low.p[2], cl.f[0], cl.f[1]);
        }
        if (FULL) __builtin_amdgcn_sched_barrier(0);
    }
    planes_finish16(s.all, ca);
    planes_finish16(s.low, cl);
    if (drow) dacc = drow[0] | drow[1] | drow[2] | drow[3];
    return dacc;
}

// Second pass over a granule for the lanes that met unexpected bytes (IUPAC codes, protein letters, noise): each
// such byte goes into the wave's LDS histogram, and the class the planes counted it as is taken out again.
__device__ __forceinline__ void comp_rare_pass(const uint8_t *__restrict__ data, int64_t n, int64_t gs, int a, int b,
                                            uint32_t *__restrict__ rare_hist) {
    const int lane = lane_id();
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
        const int off = ((i >> 2) * 64 + lane) * 16 + 4 * (i & 3);
        const int64_t p = gs + off;
        uint32_t x = 0;
        if (p + 4 <= n) x = *reinterpret_cast<const uint32_t *>(data + p);
        else for (int k = 0; k < 4; ++k) if (p + k < n) x |= (uint32_t)data[p + k] << (8 * k);
        const uint32_t m = word_range_mask(a - off, b - off);
        x = (x & m) | (0x0A0A0A0Au & ~m);
        uint32_t h, hl, d;
        comp_classify(x, h, hl, d);
        if (!d) continue;
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
            if (!((d >> (8 * k)) & 0xFFu)) continue;
            const uint32_t byte = (x >> (8 * k)) & 0xFFu, hk = (h >> (8 * k)) & 0xFFu, lk = (hl >> (8 * k)) & 0xFFu;
            if (byte < 128) atomicAdd(&rare_hist[byte], 1u);
            if (hk) {                           // counted as class hk by the planes: A C G T N in the case of the byte's bit 5/6, or \r
                const uint32_t sym = hk == 1 ? 'A' : hk == 2 ? 'C' : hk == 4 ? 'G' : hk == 8 ? 'T' : hk == 16 ? 'N' : 13u;
                atomicSub(&rare_hist[hk == 32 ? 13u : (sym | (lk ? 0x20u : 0u))], 1u);
            }
        }
    }
}

// a value every lane holds identically, moved to scalar registers (so that what is computed from it stays scalar)
__device__ __forceinline__ int64_t uniform64(int64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ void comp_load_granule(uint4 (&v)[4], const uint8_t *__restrict__ data, int64_t gs, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {               // the four component loads fuse into one global_load_dwordx4 nt
        const uint4 *q = reinterpret_cast<const uint4 *>(data + gs + (j * 64 + lane) * 16);
        v[j].x = __builtin_nontemporal_load(&q->x); v[j].y = __builtin_nontemporal_load(&q->y);
        v[j].z = __builtin_nontemporal_load(&q->z); v[j].w = __builtin_nontemporal_load(&q->w);
    }
}

// One wave counts a run of `gpw` consecutive granules, 4 waves per workgroup.  (Interleaving the granules of the
// resident waves instead -- wave t of a team takes granules t, t + team, ... -- was measured slower: 0.83 vs 0.60 ms.)
// The kernel exists twice: PURE takes the runs that lie inside one record's sequence block (all but a handful for
// a genome) with a tight double-buffered loop and lists the others in edge_list; the second launch takes those --
// header lines, record boundaries, the tail of the stream -- segment by segment.
#ifndef FX_COMP_PROBE
#define FX_COMP_PROBE 0                         // tools/compbench.hip: 1 = loads only, 2 = counting only (timing probes)
#endif
constexpr int COMP_WPB = 8;                     // waves per workgroup
template <bool PURE>
__global__ __launch_bounds__(COMP_WPB * 64) void k_fasta_comp(const uint8_t *__restrict__ data, int64_t n, int64_t gbase,
                                                     const int64_t *__restrict__ hdr, const int64_t *__restrict__ boff,
                                                     int64_t n_hdr, const int64_t *__restrict__ hdr_prefix,
                                                     int64_t ngran, int gpw, int32_t *__restrict__ edge_list, int64_t lead_from,
                                                     unsigned long long *__restrict__ comp) {
    __shared__ uint32_t rare_all[COMP_WPB][128];
    __shared__ uint32_t blk_cnt[16];
    const int lane = lane_id(), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar: all control below is wave-uniform
    uint32_t *rare_hist = rare_all[wv];
    int64_t wave = (int64_t)blockIdx.x * COMP_WPB + wv;
    int64_t nreal = (n + FX_GRAN - 1) / FX_GRAN;            // granules that hold bytes (the one after the last byte holds nothing)
    if (nreal > ngran) nreal = ngran;
    CompState s;
    s.rec = COMP_NONE; s.rare = false;
    comp_reset(s);
    const int64_t rmin = lead_from >= 0 ? -1 : 0;           // lowest record index that is counted
    auto rec_boff = [&](int64_t r) { return r >= 0 ? uniform64(boff[r]) : lead_from; };   // where the record's sequence bytes begin
    if (PURE) {
        if (threadIdx.x < 16) blk_cnt[threadIdx.x] = 0;
        __syncthreads();
        const int64_t gblock = (int64_t)blockIdx.x * COMP_WPB * gpw;      // the workgroup's first granule and its record
        const int64_t blk_rec = gblock < nreal ? uniform64(hdr_prefix[gblock]) - 1 : COMP_NONE;
        const int64_t gfirst = wave * gpw;
        bool active = gfirst < nreal;
        if (active) {
            const int cnt = (int)(nreal - gfirst < gpw ? nreal - gfirst : gpw);
            // the first granules are requested before anything else is known about the run: the table lookups below
            // are two dependent round trips, and a wave that sits through them with nothing in flight wastes its slot
            uint4 buf[COMP_DEPTH][4];
            const bool candidate = cnt == gpw && gpw % COMP_DEPTH == 0 && (gfirst + cnt) * (int64_t)FX_GRAN <= n;
            if (candidate) {
#pragma unroll
                for (int k = 0; k < COMP_DEPTH; ++k) comp_load_granule(buf[k], data, (gfirst + k) * (int64_t)FX_GRAN, lane);
            }
            const int64_t r0 = uniform64(hdr_prefix[gfirst]) - 1;
            const bool pure = candidate && uniform64(hdr_prefix[gfirst + cnt]) == r0 + 1 && r0 >= rmin &&
                              gbase + gfirst * (int64_t)FX_GRAN >= rec_boff(r0);
            if (!pure) {                        // left to the second launch
                if (lane == 0) edge_list[1 + atomicAdd(&edge_list[0], 1)] = (int32_t)wave;
            } else {
                // COMP_DEPTH granules in flight per wave: while buffer k is counted the loads of the other buffers
                // are still travelling (s_waitcnt vmcnt(4 * (COMP_DEPTH - 1))).  Granules with unexpected bytes are
                // only noted here and revisited after the loop (comp_rare_pass re-reads them), which keeps the
                // hot loop straight.
                rare_hist[lane] = 0; rare_hist[lane + 64] = 0;
                s.rec = r0;
                uint32_t rare_mask = 0;
                int i = 0;
                for (; i < cnt - COMP_DEPTH; i += COMP_DEPTH) {
#pragma unroll
                    for (int k = 0; k < COMP_DEPTH; ++k) {
#if FX_COMP_PROBE != 1
                        if (__ballot(comp_add_granule<true>(s, buf[k], 0, FX_GRAN) != 0)) rare_mask |= 1u << (i + k);
#else
                        for (int j = 0; j < 4; ++j) s.all.p[j] ^= buf[k][j].x ^ buf[k][j].y ^ buf[k][j].z ^ buf[k][j].w;
#endif
#if FX_COMP_PROBE != 2
                        comp_load_granule(buf[k], data, (gfirst + i + k + COMP_DEPTH) * (int64_t)FX_GRAN, lane);
#endif
                        __builtin_amdgcn_sched_barrier(0);      // one granule at a time: interleaving them only costs registers
                    }
                }
#pragma unroll
                for (int k = 0; k < COMP_DEPTH; ++k) {
                    if (__ballot(comp_add_granule<true>(s, buf[k], 0, FX_GRAN) != 0)) rare_mask |= 1u << (i + k);
                    __builtin_amdgcn_sched_barrier(0);
                }
                while (rare_mask) {
                    const int g = __builtin_ctz(rare_mask);
                    rare_mask &= rare_mask - 1;
                    comp_rare_pass(data, n, (gfirst + g) * (int64_t)FX_GRAN, 0, FX_GRAN, rare_hist);
                    s.rare = true;
                }
                comp_flush(s, rare_hist, comp, blk_cnt, blk_rec, n_hdr);
            }
        }
        __syncthreads();
        if (threadIdx.x < 16 && blk_rec >= rmin) {
            const uint32_t v = blk_cnt[threadIdx.x];
            if (v) atomicAdd(&comp[(blk_rec >= 0 ? blk_rec : n_hdr) * 128 + comp_symbol((int)threadIdx.x)], (unsigned long long)v);
        }
        return;
    }
    // the runs the PURE launch left over: edge_list[0] of them, ids from [1]
    if (wave >= edge_list[0]) return;
    wave = edge_list[1 + wave];
    const int64_t gfirst = wave * gpw;
    if (gfirst >= nreal) return;
    const int cnt = (int)(nreal - gfirst < gpw ? nreal - gfirst : gpw);    // granules of this wave: gfirst + i, i < cnt
    rare_hist[lane] = 0; rare_hist[lane + 64] = 0;
    uint4 v[4];
    bool loaded = false, table = false;
    int i = 0;
    int64_t r = -1, hb = 0;
    unsigned long long segmask = 0;                          // table mode: lanes whose record has bytes to count in this granule
    int seg_a = 0, seg_b = 0;                               // ... and this lane's byte range (granule-relative)
    // every pass of the loop produces the next (record, byte range) segment of the wave's granules -- one per
    // granule inside a sequence block -- flushes the counters when the record changes, and counts the segment.
    // The segments of a granule that holds header lines come from a table built once per granule, one lane per
    // record (two vector loads instead of two scalar round trips per record); records of at most COMP_SMALL bytes
    // are left out: k_fasta_comp_small counts them.  All control values are wave-uniform.
    for (;;) {
        bool have = false, full = false;
        int64_t rr = COMP_NONE, gseg = 0;
        int a = 0, b = 0;
        while (!have && i < cnt) {
            const int64_t g = gfirst + i;
            const int64_t gs = g * (int64_t)FX_GRAN;
            const int64_t ge = (gs + FX_GRAN < n) ? gs + FX_GRAN : n;
            gseg = gs;
            if (!loaded) {
                const bool whole = ge - gs == FX_GRAN;
                if (whole) {
                    comp_load_granule(v, data, gs, lane);
                } else {
#pragma unroll 1
                    for (int j = 0; j < 4; ++j) {       // the partial last granule: bytes past the end read as 0 (and are masked out)
                        const uint4 t = load16(data, gs + (j * 64 + lane) * 16, n);
                        if (j == 0) v[0] = t; else if (j == 1) v[1] = t; else if (j == 2) v[2] = t; else v[3] = t;
                    }
                }
                loaded = true;
                hb = uniform64(hdr_prefix[g]);
                const int64_t he = uniform64(hdr_prefix[g + 1]);
                r = hb - 1;                     // record that owns the first byte of the granule
                if (whole && he == hb && r >= rmin && gbase + gs >= rec_boff(r)) {     // inside one record's sequence block
                    have = true; full = true; rr = r; a = 0; b = FX_GRAN;
                    ++i; loaded = false;
                    break;
                }
                table = he - hb + 1 <= 64;
                if (table) {                    // lane k <-> record hb - 1 + k
                    const int64_t rk = hb - 1 + lane;
                    const bool valid = lane <= he - hb && rk >= rmin && rk < n_hdr;
                    int64_t rb = 0, re = 0;
                    if (valid) {
                        rb = (rk >= 0 ? boff[rk] : lead_from) - gbase;
                        re = rk + 1 < n_hdr ? hdr[rk + 1] - gbase : n;
                    }
                    const int64_t aa = rb > gs ? rb : gs, bb = re < ge ? re : ge;
                    const bool small = rk >= 0 && rb >= 0 && re - rb <= COMP_SMALL;      // k_fasta_comp_small's records
                    segmask = __ballot(valid && aa < bb && !small);
                    seg_a = (int)(aa - gs); seg_b = (int)(bb - gs);
                }
            }
            if (table) {
                if (!segmask) { ++i; loaded = false; continue; }
                const int l = __ffsll(segmask) - 1;
                segmask &= segmask - 1;
                have = true; rr = hb - 1 + l;
                a = __builtin_amdgcn_readlane(seg_a, l); b = __builtin_amdgcn_readlane(seg_b, l);
                if (!segmask) { ++i; loaded = false; }
                break;
            }
            const int64_t nexth = (r + 1 < n_hdr) ? uniform64(hdr[r + 1]) - gbase : INT64_MAX;
            const int64_t bb = nexth < ge ? nexth : ge;
            if (r >= rmin) {
                int64_t aa = rec_boff(r) - gbase;
                if (aa < gs) aa = gs;
                const bool small = r >= 0 && rec_boff(r) - gbase >= 0 && (nexth == INT64_MAX ? n : nexth) - (rec_boff(r) - gbase) <= COMP_SMALL;
                if (aa < bb && !small) { have = true; rr = r; a = (int)(aa - gs); b = (int)(bb - gs); }
            }
            if (nexth >= ge) { ++i; loaded = false; } else ++r;
        }
        if (rr != s.rec) { comp_flush(s, rare_hist, comp, nullptr, COMP_NONE, n_hdr); s.rec = rr; }
        if (!have) break;
        const uint32_t dacc = full ? comp_add_granule<true>(s, v, 0, FX_GRAN) : comp_add_granule<false>(s, v, a, b);
        if (__ballot(dacc != 0)) {
            comp_rare_pass(data, n, gseg, a, b, rare_hist);
            s.rare = true;
        }
    }
}

// ------------------------------------------------------------------ short records
// A record of at most COMP_SMALL bytes (from its first sequence byte to the next header line) is counted whole by ONE
// 16-lane group -- four records per wave, 16 bytes per lane and step -- and its row is written once, instead of being met
// as a masked segment by every granule it touches and flushed there (a file of short records spent 7.5 ms in those
// flushes where this kernel needs about one).  Same classification; the four words of a piece go through three full
// adders and only the carry-out is popcounted, as in k_fastq_comp.  Unexpected bytes: per-group LDS histogram, and the
// aliased class is taken back out.
__global__ __launch_bounds__(BLOCK) void k_fasta_comp_small(const uint8_t *__restrict__ data, int64_t n, int64_t gbase,
                                                           const int64_t *__restrict__ hdr, const int64_t *__restrict__ boff,
                                                           int64_t n_hdr, unsigned long long *__restrict__ comp) {
    __shared__ int rare_all[BLOCK / 64][4][128];
    const int lane = lane_id(), sub = lane & 15, grp = lane >> 4, wv = threadIdx.x >> 6;
    int *rare = rare_all[wv][grp];
    for (int k = sub; k < 128; k += 16) rare[k] = 0;
    const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
    for (int64_t r0 = wave * 4; r0 < n_hdr; r0 += nwaves * 4) {
        const int64_t r = r0 + grp;
        int64_t s = 0, e = 0;
        if (r < n_hdr) {
            s = boff[r] - gbase;
            e = r + 1 < n_hdr ? hdr[r + 1] - gbase : n;
            if (e - s > COMP_SMALL || s < 0) e = s;          // not a short record (or not filled in): nothing to do here
        }
        uint32_t ones = 0, twos = 0, lones = 0, ltwos = 0, c4[6] = {0, 0, 0, 0, 0, 0}, l4[5] = {0, 0, 0, 0, 0};
        bool any_rare = false;
        for (int64_t p = s + sub * 16; p < e; p += 256) {
            uint4 v;
            if (p + 16 <= n) v = *reinterpret_cast<const uint4_u *>(data + p);
            else {
                uint32_t w[4] = {0, 0, 0, 0};
                for (int k = 0; k < 16; ++k) if (p + k < n) w[k >> 2] |= (uint32_t)data[p + k] << ((k & 3) * 8);
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            uint32_t x[4] = {v.x, v.y, v.z, v.w}, h[4], hl[4], dacc = 0;
            const int left = (int)(e - p);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (left < 16) { const uint32_t m = word_range_mask(0, left - 4 * k); x[k] = (x[k] & m) | (0x0A0A0A0Au & ~m); }
                uint32_t d;
                comp_classify(x[k], h[k], hl[k], d);
                dacc |= d;
            }
            if (dacc) {
                any_rare = true;
#pragma unroll 1
                for (int k = 0; k < 4; ++k) {
                    uint32_t hh, ll, d;
                    comp_classify(x[k], hh, ll, d);
                    if (!d) continue;
#pragma unroll 1
                    for (int j = 0; j < 4; ++j) {
                        if (!((d >> (8 * j)) & 0xFFu)) continue;
                        const uint32_t byte = (x[k] >> (8 * j)) & 0xFFu, hk = (hh >> (8 * j)) & 0xFFu, lk = (ll >> (8 * j)) & 0xFFu;
                        if (byte < 128) atomicAdd(&rare[byte], 1);
                        if (hk) {
                            const uint32_t sym = hk == 1 ? 'A' : hk == 2 ? 'C' : hk == 4 ? 'G' : hk == 8 ? 'T' : hk == 16 ? 'N' : 13u;
                            atomicSub(&rare[hk == 32 ? 13u : (sym | (lk ? 0x20u : 0u))], 1);
                        }
                    }
                }
            }
            uint32_t tA, tB, f;
            csa(tA, ones, ones, h[0], h[1]);   csa(tB, ones, ones, h[2], h[3]);   csa(f, twos, twos, tA, tB);
#pragma unroll
            for (int c = 0; c < 6; ++c) c4[c] += __popc(f & (0x01010101u << c));
            csa(tA, lones, lones, hl[0], hl[1]); csa(tB, lones, lones, hl[2], hl[3]); csa(f, ltwos, ltwos, tA, tB);
#pragma unroll
            for (int c = 0; c < 5; ++c) l4[c] += __popc(f & (0x01010101u << c));
        }
        // the group's totals (xor shuffles stay inside the 16 lanes), one lane per class writes the row
        uint32_t mine = 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const uint32_t m = 0x01010101u << c;
            uint32_t tot = __popc(ones & m) + 2 * __popc(twos & m) + 4 * c4[c];
            uint32_t lo = c < 5 ? __popc(lones & m) + 2 * __popc(ltwos & m) + 4 * l4[c] : 0u;
            tot -= lo;
#pragma unroll
            for (int d = 8; d > 0; d >>= 1) { tot += __shfl_xor(tot, d, 64); lo += __shfl_xor(lo, d, 64); }
            if (sub == c) mine = tot;
            if (c < 5 && sub == 8 + c) mine = lo;
        }
        if (mine && r < n_hdr) atomicAdd(&comp[r * 128 + comp_symbol(sub)], (unsigned long long)mine);
        const unsigned long long rb = __ballot(any_rare);
        if ((rb >> (grp * 16)) & 0xFFFFull) {               // this group met unexpected bytes: its histogram joins the row
            for (int k = sub; k < 128; k += 16) {
                const int vv = rare[k];
                if (vv) { atomicAdd(&comp[r * 128 + k], (unsigned long long)(long long)vv); rare[k] = 0; }
            }
        }
    }
}

// ------------------------------------------------------------------ sparse form of the composition
// The `comp` table of the .fxi holds only the non-zero bins of a record (fasta.c:904-914) -- about ten of 128 for DNA --
// and a dense matrix of a many-record file is large (5 M records: 5 GB; a protein database: more than the host has):
// k_comp_count / k_comp_emit turn the dense rows, which stay in HBM, into (record, letter, count) triples in record
// order, and the column totals.  One wave per record (grid-stride): lane l looks at bins l and l + 64.
__global__ __launch_bounds__(BLOCK) void k_comp_count(const unsigned long long *__restrict__ comp, int64_t n_rec,
                                                     int32_t *__restrict__ cnt, unsigned long long *__restrict__ total) {
    const int lane = lane_id();
    const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
    unsigned long long s0 = 0, s1 = 0;
    for (int64_t r = wave; r < n_rec; r += nwaves) {
        const unsigned long long a = comp[r * 128 + lane], b = comp[r * 128 + 64 + lane];
        s0 += a; s1 += b;
        const int c = __popcll(__ballot(a != 0)) + __popcll(__ballot(b != 0));
        if (lane == 0) cnt[r] = c;
    }
    if (s0) atomicAdd(&total[lane], s0);
    if (s1) atomicAdd(&total[64 + lane], s1);
}

// exclusive prefix sum of cnt[0..n) into off[0..n] (off[n] = total), three small kernels, 1024 elements per workgroup
constexpr int SCAN_CHUNK = 1024;
__global__ __launch_bounds__(BLOCK) void k_cnt_chunk_sums(const int32_t *__restrict__ cnt, int64_t n, int64_t *__restrict__ sums) {
    __shared__ uint32_t lds4[BLOCK / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
    uint32_t s = 0;
    for (int k = threadIdx.x; k < SCAN_CHUNK; k += BLOCK) if (base + k < n) s += (uint32_t)cnt[base + k];
    s = block_sum(s, lds4);
    if (threadIdx.x == 0) sums[blockIdx.x] = s;
}
__global__ __launch_bounds__(BLOCK) void k_cnt_chunk_bases(int64_t *__restrict__ sums, int64_t nchunks) {   // one workgroup: in place -> exclusive
    __shared__ int64_t wtot[BLOCK / 64];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t a = 0; a < nchunks; a += BLOCK) {
        const int64_t i = a + threadIdx.x;
        const int64_t v = i < nchunks ? sums[i] : 0;
        int64_t inc = v;                                    // inclusive scan over the wave (shuffles: 64-bit values)
        for (int d = 1; d < 64; d <<= 1) { const int64_t t = __shfl_up(inc, d, 64); if (lane_id() >= d) inc += t; }
        if (lane_id() == 63) wtot[threadIdx.x >> 6] = inc;
        __syncthreads();
        int64_t b = carry;
        for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) b += wtot[k];
        if (i < nchunks) sums[i] = b + inc - v;
        __syncthreads();
        if (threadIdx.x == BLOCK - 1) carry = b + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[nchunks] = carry;            // grand total
}
__global__ __launch_bounds__(BLOCK) void k_cnt_offsets(const int32_t *__restrict__ cnt, int64_t n, const int64_t *__restrict__ bases,
                                                      int64_t *__restrict__ off) {
    __shared__ uint32_t wtot[BLOCK / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
    const int lane = lane_id(), w = threadIdx.x >> 6;
    // thread t owns elements base + 4t .. 4t+3 (consecutive, so one pass of wave scans is enough)
    uint32_t v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int64_t i = base + 4 * threadIdx.x + k; v[k] = i < n ? (uint32_t)cnt[i] : 0u; s += v[k]; }
    const uint32_t inc = wave_incl_scan(s);
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    uint32_t b = inc - s;
    for (int k = 0; k < w; ++k) b += wtot[k];
    int64_t o = bases[blockIdx.x] + b;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int64_t i = base + 4 * threadIdx.x + k; if (i < n) off[i] = o; o += v[k]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) off[n] = bases[gridDim.x];
}

__global__ __launch_bounds__(BLOCK) void k_comp_emit(const unsigned long long *__restrict__ comp, int64_t n_rec,
                                                    const int64_t *__restrict__ off, int64_t *__restrict__ seqid,
                                                    int64_t *__restrict__ abc, int64_t *__restrict__ num) {
    const int lane = lane_id();
    const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
    const uint64_t lt = (1ull << lane) - 1;
    for (int64_t r = wave; r < n_rec; r += nwaves) {
        const unsigned long long a = comp[r * 128 + lane], b = comp[r * 128 + 64 + lane];
        const uint64_t ma = __ballot(a != 0), mb = __ballot(b != 0);
        const int64_t o = off[r];
        if (a) { const int64_t p = o + __popcll(ma & lt); seqid[p] = r + 1; abc[p] = lane; num[p] = (int64_t)a; }
        if (b) { const int64_t p = o + __popcll(ma) + __popcll(mb & lt); seqid[p] = r + 1; abc[p] = 64 + lane; num[p] = (int64_t)b; }
    }
}

}  // namespace fx
