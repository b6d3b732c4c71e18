// fx_spanscan.hpp -- single-read FASTA scan for gfx950 (MI355X, wave64).
//
// The reference walks lines one at a time and, per record, remembers the first
// sequence line's length and counts later lines that differ (index.c:230-339).
// Here the stream is read ONCE and nothing proportional to the file is written
// back: every 4 KiB granule (one wave's share) is reduced to a fixed-size summary
//
//   n_nl, n_hdr            newlines / header lines ('>' after '\n') that START in the granule
//   first_nl, last_nl      granule-local offsets of its first / last newline
//   (v1,c1) (v2,c2) ovf    the multiset of distances between consecutive newlines
//                          that both lie in the granule, as "first two distinct values
//                          and their counts" (+ overflow flag when a third shows up)
//
// which is all the record table needs: a record's bad_line count (index.c:325-327)
// only asks "how many line lengths differ from llen", so a stretch whose lines have
// <= 2 distinct lengths answers it exactly for any llen, without a line table.
// Stretches that contain a header line or overflow are re-read by k_span_exact
// (rare for genomes: one per record).  Header positions go to an append list with
// their granule-local rank and the number of granule newlines before them.
//
// Work decomposition: waves are independent (no LDS, no barrier).  A wave owns a
// contiguous 4 KiB granule as 4 rows of 1 KiB (one global_load_dwordx4 per lane per
// row: 64 lanes x 16 B contiguous) and walks granules with a grid stride.  Newline
// state is wave-uniform (SGPRs).  Fast path per row: every lane holds at most one
// newline and the newline positions continue the arithmetic progression of the
// current line length L (a hint carried from row to row): rank of a lane among the
// newline lanes by v_mbcnt, expected position carry + L*(rank+1), one compare, one
// ballot.  Anything else (first row, a record boundary, short or ragged lines)
// takes the general path: previous newline of a lane = highest set ballot bit
// below it (one ds_bpermute), distances fed to the two-value set.
#pragma once
#include "fx_kernels.hpp"

namespace fx {

#ifndef FX_GRAN
#define FX_GRAN 4096
#endif
constexpr int GRAN = FX_GRAN;                  // bytes per wave per step
constexpr int GR_ROWS = GRAN / (64 * CHUNK);   // 4 rows of 1 KiB
constexpr int SPAN_GRANS = 65536 / GRAN;
constexpr int SPAN = GRAN * SPAN_GRANS;     // 64 KiB: granularity of the post-scan passes
constexpr uint32_t SP_NONE = 0xFFFFFFFFu;

// per-granule summary: in registers (GranOut) and as stored by phase 1 (GranPk, one 16-byte word:
// every field is < 2^13 because it counts or addresses bytes of one granule)
struct GranOut { uint32_t n, h, first, last, v1, c1, v2, c2, ovf; };   // first/last: SP_NONE when n == 0
struct __attribute__((aligned(16))) GranPk { uint32_t nh, fl, d1, d2; };
__device__ __forceinline__ GranPk gran_pack(const GranOut &o) {
    GranPk p;
    p.nh = o.n | (o.h << 16); p.fl = (o.first & 0xFFFFu) | (o.last << 16);
    p.d1 = o.v1 | (o.c1 << 16); p.d2 = o.v2 | (o.c2 << 16) | (o.ovf << 31);
    return p;
}
__device__ __forceinline__ GranOut gran_unpack(const GranPk &p) {
    GranOut o;
    o.n = p.nh & 0xFFFFu; o.h = p.nh >> 16;
    o.first = o.n ? (p.fl & 0xFFFFu) : SP_NONE; o.last = o.n ? (p.fl >> 16) : SP_NONE;
    o.v1 = p.d1 & 0xFFFFu; o.c1 = p.d1 >> 16; o.v2 = p.d2 & 0xFFFFu; o.c2 = (p.d2 >> 16) & 0x7FFFu; o.ovf = p.d2 >> 31;
    return o;
}

struct DiffSet {                // first two distinct values with counts (+ overflow)
    uint32_t v1, c1, v2, c2, ovf;
    __device__ __forceinline__ void clear() { v1 = c1 = v2 = c2 = ovf = 0; }
    __device__ __forceinline__ void add(uint32_t d, uint32_t c) {
        if (!c) return;
        if (v1 == 0 || v1 == d) { v1 = d; c1 += c; }
        else if (v2 == 0 || v2 == d) { v2 = d; c2 += c; }
        else ovf = 1;
    }
    // number of the summarised distances that differ from llen; exact unless ovf
    __device__ __forceinline__ uint32_t count_ne(uint32_t llen) const {
        return (c1 && v1 != llen ? c1 : 0u) + (c2 && v2 != llen ? c2 : 0u);
    }
};

__device__ __forceinline__ uint32_t rdlane(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(lane));
}

// wave-uniform DiffSet fed by per-lane distances
struct WaveDiff : DiffSet {
    __device__ __forceinline__ void feed(bool valid, uint32_t d) {
        const unsigned long long vb = __ballot(valid);
        if (!vb) return;
        if (v1 == 0) v1 = rdlane(d, __ffsll(vb) - 1);
        const unsigned long long e1 = __ballot(valid && d == v1);
        c1 += (uint32_t)__popcll(e1);
        const unsigned long long rem = vb & ~e1;
        if (rem) {
            if (v2 == 0) v2 = rdlane(d, __ffsll(rem) - 1);
            const unsigned long long e2 = __ballot(valid && d == v2);
            c2 += (uint32_t)__popcll(e2);
            if (rem & ~e2) ovf = 1;
        }
    }
};

// exact header mask of a 16-byte chunk whose first byte is data[gp]:
// bit k set iff byte k is '>' and the byte before it is '\n' (index.c:234: line.s[0] == '>')
__device__ __forceinline__ uint32_t header_mask16(const uint4 &v, uint32_t nlm, const uint8_t *__restrict__ data,
                                                  int64_t gp, int prev_byte) {
    const uint32_t g = eq_mask16(v, 0x3E3E3E3Eu);
    if (!g) return 0;
    uint32_t hm = g & (nlm << 1);
    if (g & 1u) {
        const int prev = gp ? (int)data[gp - 1] : prev_byte;
        if (prev == '\n') hm |= 1u;
    }
    return hm & 0xFFFFu;
}

// does any byte of the chunk lie in 0x20..0x3F (space, digits, punctuation -- '>' is 0x3E)?  Sequence
// letters (bit 6 set) and '\n' / '\r' (bit 5 clear) never do, so this cheap filter is false for
// every chunk of pure sequence and the exact '>' test only runs on header lines.
__device__ __forceinline__ bool maybe_gt16(const uint4 &v) {
    const uint32_t a = (((v.x ^ 0x20202020u) & 0x60606060u) + 0x7F7F7F7Fu) & (((v.y ^ 0x20202020u) & 0x60606060u) + 0x7F7F7F7Fu);
    const uint32_t b = (((v.z ^ 0x20202020u) & 0x60606060u) + 0x7F7F7F7Fu) & (((v.w ^ 0x20202020u) & 0x60606060u) + 0x7F7F7F7Fu);
    return (a & b & 0x80808080u) != 0x80808080u;
}

// wave total of a small per-lane count (< 32) without shuffles: one ballot per bit
__device__ __forceinline__ uint32_t wave_sum_small(uint32_t c) {
    uint32_t s = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) s += (uint32_t)__popcll(__ballot((c >> b) & 1u)) << b;
    return s;
}

// ============================================================== phase 1
// One granule.  FULL: it lies entirely inside the stream (all but the last one): plain 16-byte
// loads issued back to back, no end-of-stream handling.  L: line-length hint, carried across granules.
// Granules that hold a header line are appended to hgl (order irrelevant): k_hdr_collect re-reads them.
struct GranList { uint32_t *g; uint32_t *count; };

// MODE 0: the full FASTA summary.  MODE 1 (FASTQ, where records are "every four lines"): newline count and the
// first / last newline only -- no line-length set, no header lines.
// Returns (wave-uniform) the number of header lines that start in the granule.  The caller puts the granule on
// the list hgl when that is non-zero (k_span_scan: once per workgroup, see there).
template <bool FULL>
__device__ __forceinline__ void granule_load(uint4 (&v)[GR_ROWS], const uint8_t *__restrict__ data, int64_t n, int is_last, int64_t g) {
    const int lane = lane_id();
    const int64_t sbase = g * (int64_t)GRAN;
    if (FULL) {
        const uint4 *q = reinterpret_cast<const uint4 *>(data + sbase + lane * CHUNK);
#pragma unroll
        for (int j = 0; j < GR_ROWS; ++j) {
            v[j].x = __builtin_nontemporal_load(&q[j * 64].x); v[j].y = __builtin_nontemporal_load(&q[j * 64].y);
            v[j].z = __builtin_nontemporal_load(&q[j * 64].z); v[j].w = __builtin_nontemporal_load(&q[j * 64].w);
        }
    } else {
#pragma unroll
        for (int j = 0; j < GR_ROWS; ++j) v[j] = load16(data, sbase + j * 1024 + lane * CHUNK, n);
        // virtual newline at end-of-stream when the last line is unterminated (`position += line.l + 1`
        // for that line too, index.c:231): the chunk that holds offset n gets a '\n' at n
        if (is_last) {
#pragma unroll
            for (int j = 0; j < GR_ROWS; ++j) {
                const int64_t p = sbase + j * 1024 + lane * CHUNK;
                if (n >= p && n < p + CHUNK && n > 0 && data[n - 1] != '\n') {
                    const int k = (int)(n - p);
                    const uint32_t b = 0x0Au << ((k & 3) * 8);
                    if ((k >> 2) == 0) v[j].x |= b; else if ((k >> 2) == 1) v[j].y |= b;
                    else if ((k >> 2) == 2) v[j].z |= b; else v[j].w |= b;
                }
            }
        }
    }
}

// odd (optional, MODE 0): per row, non-zero in the lanes whose chunk holds a byte that is no sequence letter and no line
// end -- a caller that has classified the bytes anyway (k_scan_comp) passes it in place of the '>' pre-filter below
template <int MODE = 0>
__device__ __forceinline__ uint32_t granule_body(const uint4 (&v)[GR_ROWS], const uint8_t *__restrict__ data, int prev_byte,
                                                 int64_t g, GranPk *__restrict__ out, uint32_t &L, const uint32_t *odd = nullptr) {
    const int lane = lane_id();
    const int64_t sbase = g * (int64_t)GRAN;

    WaveDiff wd;
    wd.clear();
    uint32_t n_w = 0, h_w = 0;                  // wave-uniform counts
    int first_w = -1, carry = -1;               // wave-uniform: first / latest newline (granule-local)

    uint32_t n_lane = 0;                         // MODE 1: per-lane newline count
#pragma unroll
    for (int j = 0; j < GR_ROWS; ++j) {
        const uint32_t cb = j * 1024 + lane * CHUNK;
        if (MODE == 1) {
            const uint32_t m = eq_mask16(v[j], 0x0A0A0A0Au);
            n_lane += __popc(m);
            const unsigned long long b = __ballot(m != 0);
            if (b) {
                if (first_w < 0) first_w = (int)rdlane(cb + (__ffs(m) - 1), __ffsll(b) - 1);
                carry = (int)rdlane(cb + (31 - __clz(m)), 63 - __clzll(b));
            }
            continue;
        }
        const uint32_t t0 = zero_bytes(v[j].x ^ 0x0A0A0A0Au), t1 = zero_bytes(v[j].y ^ 0x0A0A0A0Au);
        const uint32_t t2 = zero_bytes(v[j].z ^ 0x0A0A0A0Au), t3 = zero_bytes(v[j].w ^ 0x0A0A0A0Au);
        const uint32_t q = (t0 >> 7) | (t1 >> 6) | (t2 >> 5) | (t3 >> 4);      // bit 8*b+k <-> byte 4*k+b
        const bool has = q != 0;
        const unsigned long long bal = __ballot(has);
        uint32_t nlm = 0;                        // exact 16-bit mask, only built when needed
        if (bal) {
            const unsigned long long multi = __ballot((q & (q - 1)) != 0);
            uint32_t pf, pl;
            bool done = false;
            if (__builtin_expect(!multi, 1)) {
                const int f = __ffs(q) - 1;
                pf = pl = cb + (((f & 7) << 2) | (f >> 3));
                n_w += (uint32_t)__popcll(bal);
                if (L) {                         // do the newlines continue the progression of line length L?
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    const uint32_t p0 = carry >= 0 ? (uint32_t)carry + L : rdlane(pf, __ffsll(bal) - 1);
                    if (!__ballot(has && pf != p0 + __umul24(L, rank))) {
                        wd.add(L, (uint32_t)__popcll(bal) - (carry >= 0 ? 0u : 1u));
                        done = true;
                    }
                }
            } else {
                nlm = flags4(t0) | (flags4(t1) << 4) | (flags4(t2) << 8) | (flags4(t3) << 12);
                pf = cb + (__ffs(nlm) - 1);
                pl = cb + (31 - __clz(nlm));
                n_w += wave_sum_small(__popc(nlm));
            }
            if (__builtin_expect(!done, 0)) {
                const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
                const unsigned long long lower = bal & lt;
                const int prevlane = 63 - __clzll(lower | 1ull);
                const uint32_t prevP = (uint32_t)__builtin_amdgcn_ds_bpermute(prevlane << 2, (int)pl);
                const bool has_prev = lower != 0 || carry >= 0;
                const uint32_t pp = lower ? prevP : (uint32_t)carry;
                wd.feed(has && has_prev, pf - pp);
                if (multi) {                      // distances between newlines inside one 16-byte chunk
                    uint32_t m = nlm;
                    while (__ballot((m & (m - 1)) != 0)) {
                        const int lo = __ffs(m) - 1;
                        const bool two = (m & (m - 1)) != 0;
                        m &= m - 1;
                        const int nx = __ffs(m) - 1;
                        wd.feed(two, (uint32_t)(nx - lo));
                    }
                }
                if (wd.v1) L = wd.v1;
            }
            if (first_w < 0) first_w = (int)rdlane(pf, __ffsll(bal) - 1);
            carry = (int)rdlane(pl, 63 - __clzll(bal));
        }
        // header lines: the exact '>' test only runs where the cheap filter fires (header text)
        if (__builtin_expect(__ballot(odd ? odd[j] != 0 : maybe_gt16(v[j])) != 0, 0)) {
            if (!nlm) nlm = flags4(t0) | (flags4(t1) << 4) | (flags4(t2) << 8) | (flags4(t3) << 12);
            h_w += wave_sum_small(__popc(header_mask16(v[j], nlm, data, sbase + cb, prev_byte)));
        }
    }
    if (MODE == 1) n_w = wave_sum(n_lane);
    if (lane == 0) {
        GranOut o;
        o.n = n_w; o.h = h_w; o.first = (uint32_t)first_w; o.last = (uint32_t)carry;
        o.v1 = wd.v1; o.c1 = wd.c1; o.v2 = wd.v2; o.c2 = wd.c2; o.ovf = wd.ovf;
        out[g] = gran_pack(o);
    }
    return h_w;
}

template <bool FULL, int MODE = 0>
__device__ __forceinline__ uint32_t granule(const uint8_t *__restrict__ data, int64_t n, int prev_byte, int is_last,
                                            int64_t g, GranPk *__restrict__ out, uint32_t &L) {
    uint4 v[GR_ROWS];
    granule_load<FULL>(v, data, n, is_last, g);
    return granule_body<MODE>(v, data, prev_byte, g, out, L);
}

// One wave per granule.  Only the granules that lie entirely inside the stream (n / GRAN of them) are
// launched here; the last, partial one (which also holds the virtual end-of-stream newline) is done by
// k_gran_reduce with the FULL = false instantiation, so its bounds-checked loads cost this kernel
// neither registers nor branches.
// The granules that hold header lines go on a list (k_hdr_rec visits them).  Appending one by one -- an atomic with
// return on ONE counter per granule -- is fine for chromosomes and was the whole run time for a file of short records
// (5 M records: 405 k granules, all with headers, 11 ns per atomic = 4.6 ms): the waves of a workgroup collect their
// granules in LDS and the last one to finish reserves the block's slots with a single atomic.  No barrier at the end,
// so no wave waits for another; the one at the start costs nothing (the waves of a workgroup start together).
#ifndef FX_SCAN_GPW
#define FX_SCAN_GPW 1
#endif
// consecutive granules per wave, all requested before the first is processed.  1 is best: 3.2 GB in 0.468 ms against
// 0.482 ms with 2 and 0.528 ms with 4 (tools/scanbench2.hip) -- more loads in flight per wave do not make up for
// fewer, longer-lived waves here
constexpr int SCAN_GPW = FX_SCAN_GPW;
template <int MODE>
__global__ __launch_bounds__(1024) void k_span_scan(const uint8_t *__restrict__ data, int64_t n, int prev_byte,
                                                   int is_last, int64_t g_end, GranPk *__restrict__ out, GranList hgl) {
    __shared__ uint32_t hl_n, hl_done, hl_g[16 * SCAN_GPW];
    if (MODE == 0) {
        if (threadIdx.x == 0) { hl_n = 0; hl_done = 0; }
        __syncthreads();
    }
    const int64_t g0 = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * SCAN_GPW;
    uint4 v[SCAN_GPW][GR_ROWS];
#pragma unroll
    for (int k = 0; k < SCAN_GPW; ++k) if (g0 + k < g_end) granule_load<true>(v[k], data, n, is_last, g0 + k);
    uint32_t L = 0, hmask = 0;
#pragma unroll
    for (int k = 0; k < SCAN_GPW; ++k)
        if (g0 + k < g_end && granule_body<MODE>(v[k], data, prev_byte, g0 + k, out, L)) hmask |= 1u << k;
    if (MODE == 0 && lane_id() == 0) {
#pragma unroll
        for (int k = 0; k < SCAN_GPW; ++k) if ((hmask >> k) & 1u) hl_g[atomicAdd(&hl_n, 1u)] = (uint32_t)(g0 + k);
        __threadfence_block();
        if (atomicAdd(&hl_done, 1u) == (blockDim.x >> 6) - 1) {      // the last wave of the workgroup
            const uint32_t cnt = hl_n;
            if (cnt) {
                const uint32_t base = atomicAdd(hgl.count, cnt);
                for (uint32_t k = 0; k < cnt; ++k) hgl.g[base + k] = hl_g[k];
            }
        }
    }
}

// ============================================================== granule prefixes
// Exclusive prefixes over the granule summaries in two small kernels (no atomics, deterministic):
//   k_gran_reduce   one workgroup per CHUNK of 1024 granules (4 MiB of stream): totals.  The workgroup
//                   that owns the last granule first computes it (the partial tail of the stream).
//   k_gran_prefix   one workgroup per chunk: base = join of the totals of the chunks before it (one
//                   load per thread up to 4 GiB of stream), then a local 32-bit scan; the last one writes Totals.
// nl_prefix[g] / hdr_prefix[g] = newlines / header lines before granule g (entry [ngran] = totals),
// prevnl[g] = global offset of the last newline before granule g (-1: none in this shard).
constexpr int CHUNK_GRANS = 1024;        // largest chunk (thread-block size bound of the prefix kernels)
struct ChunkTot { long long n, h, last, pad; };
struct Totals {                       // device-side scalars of one build, copied to the host once at the end
    long long n_nl, n_hdr, last_nl, seq_len, pad0, pad1, pad2, pad3;
};

__device__ __forceinline__ long long shfl_up64(long long v, int d) {
    const int lo = __shfl_up((int)(v & 0xFFFFFFFFll), d, 64), hi = __shfl_up((int)(v >> 32), d, 64);
    return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ long long shfl64(long long v, int l) {
    const int lo = __shfl((int)(v & 0xFFFFFFFFll), l, 64), hi = __shfl((int)(v >> 32), l, 64);
    return ((long long)hi << 32) | (unsigned int)lo;
}
struct Tri { long long n, h, last; };
__device__ __forceinline__ Tri tri_join(const Tri &a, const Tri &b) { return Tri{a.n + b.n, a.h + b.h, a.last > b.last ? a.last : b.last}; }
__device__ __forceinline__ Tri tri_wave_incl(Tri v) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const Tri t{shfl_up64(v.n, d), shfl_up64(v.h, d), shfl_up64(v.last, d)};
        if (l >= d) v = tri_join(t, v);
    }
    return v;
}
// inclusive scan of v over the threads of the workgroup (<= 1024 threads); *total = join of all;
// lds (16 entries) keeps the per-wave totals until the next call
__device__ __forceinline__ Tri tri_block_incl(Tri v, Tri *lds, Tri *total) {
    const int w = threadIdx.x >> 6, l = lane_id(), nw = (blockDim.x + 63) >> 6;
    const Tri inc = tri_wave_incl(v);
    __syncthreads();
    if (l == 63) lds[w] = inc;
    __syncthreads();
    Tri base{0, 0, -1}, tot{0, 0, -1};
    for (int i = 0; i < nw; ++i) { const Tri s = lds[i]; if (i < w) base = tri_join(base, s); tot = tri_join(tot, s); }
    *total = tot;
    return tri_join(base, inc);
}
__device__ __forceinline__ Tri gran_tri(const GranPk *__restrict__ go, int64_t g, int64_t ngran, int64_t gbase) {
    if (g >= ngran) return Tri{0, 0, -1};
    const GranOut o = gran_unpack(go[g]);
    return Tri{(long long)o.n, (long long)o.h, o.n ? gbase + g * (long long)GRAN + o.last : -1};
}

// CG = granules per chunk = threads per workgroup: 256 up to 8 GB of stream (more, smaller workgroups hide the
// latency of these tiny kernels better), 1024 beyond (k_gran_prefix sums the totals of all earlier chunks in
// every workgroup: work ~ nchunks^2 / CG)
template <int MODE, int CG>
__global__ __launch_bounds__(CG) void k_gran_reduce(const uint8_t *__restrict__ data, int64_t n, int prev_byte,
                                                   int is_last, GranList hgl, GranPk *__restrict__ go,
                                                   int64_t ngran, int64_t gbase, ChunkTot *__restrict__ ct) {
    __shared__ Tri lds[CG / 64];
    if (blockIdx.x == gridDim.x - 1) {          // the tail granule (index ngran - 1) belongs to the last chunk
        if (threadIdx.x < 64) {
            uint32_t L = 0;
            const uint32_t h_w = granule<false, MODE>(data, n, prev_byte, is_last, ngran - 1, go, L);
            if (MODE == 0 && h_w && threadIdx.x == 0) hgl.g[atomicAdd(hgl.count, 1u)] = (uint32_t)(ngran - 1);
        }
        __threadfence_block();
        __syncthreads();
    }
    Tri tot;
    tri_block_incl(gran_tri(go, (int64_t)blockIdx.x * CG + threadIdx.x, ngran, gbase), lds, &tot);
    if (threadIdx.x == 0) ct[blockIdx.x] = ChunkTot{tot.n, tot.h, tot.last, 0};
}

// inclusive scan of three 32-bit values (sum, sum, max) over the workgroup; lds: 3 x 16 words
struct Tri32 { uint32_t n, h; int32_t last; };
__device__ __forceinline__ Tri32 tri32_block_incl(Tri32 v, uint32_t (*lds)[16], Tri32 *total) {
    const int w = threadIdx.x >> 6, l = lane_id(), nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t a = __shfl_up(v.n, d, 64), b = __shfl_up(v.h, d, 64);
        const int32_t c = __shfl_up(v.last, d, 64);
        if (l >= d) { v.n += a; v.h += b; v.last = c > v.last ? c : v.last; }
    }
    __syncthreads();
    if (l == 63) { lds[0][w] = v.n; lds[1][w] = v.h; lds[2][w] = (uint32_t)v.last; }
    __syncthreads();
    Tri32 base{0, 0, -1}, tot{0, 0, -1};
    for (int i = 0; i < nw; ++i) {
        const uint32_t a = lds[0][i], b = lds[1][i];
        const int32_t c = (int32_t)lds[2][i];
        if (i < w) { base.n += a; base.h += b; base.last = c > base.last ? c : base.last; }
        tot.n += a; tot.h += b; tot.last = c > tot.last ? c : tot.last;
    }
    *total = tot;
    return Tri32{base.n + v.n, base.h + v.h, base.last > v.last ? base.last : v.last};
}

template <int CG>
__global__ __launch_bounds__(CG) void k_gran_prefix(const GranPk *__restrict__ go, int64_t ngran, int64_t gbase,
                                                            const ChunkTot *__restrict__ ct, Totals *__restrict__ tot,
                                                            int64_t *__restrict__ nl_prefix, int64_t *__restrict__ hdr_prefix,
                                                            int64_t *__restrict__ prevnl) {
    __shared__ Tri lds[CG / 64];
    __shared__ uint32_t lds32[3][16];
    // base: join of the totals of the chunks before this one (a ticketed "last workgroup scans the totals"
    // variant inside k_gran_reduce was measured slower: 47 us for the pair instead of 31)
    Tri b{0, 0, -1};
    for (int64_t c = threadIdx.x; c < (int64_t)blockIdx.x; c += CG) { const ChunkTot t = ct[c]; b = tri_join(b, Tri{t.n, t.h, t.last}); }
    Tri base{0, 0, -1};
    if (blockIdx.x) tri_block_incl(b, lds, &base);         // workgroup-uniform branch
    // local scan in 32 bits: counts of one chunk are < 2^23, offsets relative to the chunk start < 2^22
    const int64_t g = (int64_t)blockIdx.x * CG + threadIdx.x;
    Tri32 v{0, 0, -1};
    if (g < ngran) {
        const GranOut o = gran_unpack(go[g]);
        v = Tri32{o.n, o.h, o.n ? (int32_t)(threadIdx.x * GRAN + o.last) : -1};
    }
    Tri32 total;
    const Tri32 inc = tri32_block_incl(v, lds32, &total);
    const int64_t cstart = gbase + (int64_t)blockIdx.x * CG * GRAN;
    if (g < ngran) {
        nl_prefix[g] = base.n + (inc.n - v.n);
        hdr_prefix[g] = base.h + (inc.h - v.h);
        // exclusive max: the inclusive value of the previous thread
        int32_t pm = __shfl_up(inc.last, 1, 64);
        if (lane_id() == 0) { pm = -1; for (int i = 0; i < (int)(threadIdx.x >> 6); ++i) { const int32_t c = (int32_t)lds32[2][i]; pm = c > pm ? c : pm; } }
        prevnl[g] = pm >= 0 ? cstart + pm : base.last;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const int64_t tn = base.n + total.n, th = base.h + total.h, tl = total.last >= 0 ? cstart + total.last : base.last;
        nl_prefix[ngran] = tn; hdr_prefix[ngran] = th; prevnl[ngran] = tl;
        tot->n_nl = tn; tot->n_hdr = th; tot->last_nl = tl;
    }
}

// ============================================================== navigation by granule
struct ScanCtx {
    const uint8_t *data; int64_t n, gbase, ngran;
    const GranPk *go;
    const int64_t *nl_prefix, *hdr_prefix, *prevnl;      // [ngran + 1]
};

// first newline at a global offset > x (x >= gbase - 1), or -1 when the shard holds none.  The virtual
// end-of-stream newline (offset gbase + n) counts: it is in the last granule's summary.
__device__ __forceinline__ int64_t next_nl(const ScanCtx &c, int64_t x) {
    const int64_t y = x + 1 - c.gbase;                    // local offset of the first candidate byte
    if (y > c.n) return -1;
    const int64_t g = y / GRAN;
    if (g < c.ngran) {
        const GranOut o = gran_unpack(c.go[g]);
        const int64_t gs = g * (int64_t)GRAN;
        if (o.n && gs + o.last >= y) {                    // the answer is in this granule
            if (gs + o.first >= y) return c.gbase + gs + o.first;
            for (int64_t p = y & ~(int64_t)(CHUNK - 1); p < gs + GRAN; p += CHUNK) {
                const uint4 v = load16(c.data, p, c.n);
                uint32_t m = eq_mask16(v, 0x0A0A0A0Au);
                if (p < y) m &= 0xFFFFu << (y - p);
                if (m) return c.gbase + p + (__ffs(m) - 1);
                if (p + CHUNK > c.n) break;
            }
            return c.gbase + gs + o.last;                 // only the virtual newline is left
        }
    }
    // a later granule: the first g' > g whose inclusive last-newline offset reaches past granule g
    const int64_t key = c.gbase + (g + 1) * (int64_t)GRAN;
    int64_t lo = g + 1, hi = c.ngran;
    if (lo >= hi || c.prevnl[c.ngran] < key) return -1;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (c.prevnl[mid + 1] >= key) hi = mid; else lo = mid + 1; }
    return c.gbase + lo * (int64_t)GRAN + (c.go[lo].fl & 0xFFFFu);
}

// ============================================================== record table
// first newline (and, if want_ws, first ' ' / '\t') at a local offset >= y, 64 bytes per step (four
// independent 16-byte loads).  Returns local offsets, -1 when the shard's bytes end first.
__device__ __forceinline__ void scan_line(const uint8_t *__restrict__ data, int64_t n, int64_t y, bool want_ws,
                                          int64_t *e_out, int64_t *ws_out, int *cr_before) {
    int64_t e = -1, ws = -1;
    int cr = 0;                                            // is the byte before the newline a '\r'
    unsigned long long prevC = 0;
    for (int64_t p = y & ~(int64_t)(CHUNK - 1); p < n; p += 4 * CHUNK) {
        unsigned long long M = 0, W = 0, C = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 v = load16(data, p + i * CHUNK, n);
            M |= (unsigned long long)eq_mask16(v, 0x0A0A0A0Au) << (16 * i);
            C |= (unsigned long long)eq_mask16(v, 0x0D0D0D0Du) << (16 * i);
            if (want_ws) W |= (unsigned long long)(eq_mask16(v, 0x20202020u) | eq_mask16(v, 0x09090909u)) << (16 * i);
        }
        if (p < y) { const unsigned long long keep = ~0ull << (y - p); M &= keep; W &= keep; }
        if (want_ws && ws < 0 && W) ws = p + (__ffsll(W) - 1);
        if (M) {
            const int k = __ffsll(M) - 1;
            e = p + k;
            cr = k ? (int)((C >> (k - 1)) & 1ull) : (int)(prevC >> 63);
            break;
        }
        prevC = C;
    }
    *e_out = e; *ws_out = ws; *cr_before = cr;
}

// columns of index.c:234-339 that depend only on the header line and the line after it, read from
// memory by the whole wave, 1 KiB per step (the fall-back of k_hdr_rec for records whose first two lines
// leave the granule; all arguments wave-uniform).
// first newline (and, if want_ws, first ' ' / '\t') at a local offset >= y; -1 when the shard's bytes end first
__device__ __forceinline__ void scan_line_wave(const uint8_t *__restrict__ data, int64_t n, int64_t y, bool want_ws,
                                               int64_t *e_out, int64_t *ws_out, int *cr_before) {
    const int lane = lane_id();
    int64_t e = -1, ws = -1;
    int cr = 0, prev_cr = 0;                               // prev_cr: is the last byte of the previous window a '\r'
    for (int64_t p0 = y & ~(int64_t)(CHUNK - 1); p0 < n; p0 += 64 * CHUNK) {
        const int64_t p = p0 + lane * CHUNK;
        const uint4 v = load16(data, p, n);
        uint32_t m = eq_mask16(v, 0x0A0A0A0Au), c = eq_mask16(v, 0x0D0D0D0Du);
        uint32_t w = want_ws ? (eq_mask16(v, 0x20202020u) | eq_mask16(v, 0x09090909u)) : 0u;
        if (p < y) { const uint32_t keep = (y - p >= CHUNK) ? 0u : (0xFFFFu << (y - p)); m &= keep; w &= keep; }
        const unsigned long long bw = __ballot(w != 0), bm = __ballot(m != 0);
        if (want_ws && ws < 0 && bw) { const int l = __ffsll(bw) - 1; ws = p0 + l * CHUNK + (__ffs(rdlane(w, l)) - 1); }
        if (bm) {
            const int l = __ffsll(bm) - 1;
            const int k = __ffs(rdlane(m, l)) - 1;
            e = p0 + l * CHUNK + k;
            if (k) cr = (int)((rdlane(c, l) >> (k - 1)) & 1u);
            else   cr = l ? (int)(rdlane(c, l - 1) >> 15) : prev_cr;
            break;
        }
        prev_cr = (int)(rdlane(c, 63) >> 15);
    }
    *e_out = e; *ws_out = ws; *cr_before = cr;
}

__device__ __forceinline__ void record_at(const ScanCtx &x, int is_last, int full_name, int64_t h, int64_t k,
                                          const FastaCols &c) {
    int64_t e, ws, e1, dummy;
    int cr, cr1;
    scan_line_wave(x.data, x.n, h + 1 - x.gbase, !full_name, &e, &ws, &cr);
    if (e < 0 && is_last) { e = x.n; cr = x.data[x.n - 1] == '\r'; }     // virtual end-of-stream newline (index.c:231)
    const bool lead = lane_id() == 0;
    if (e < 0) {
        // only possible for the LAST header of a non-final shard: its line ends in a later shard.
        // Leave a stub (dlen = -1) for the host-side stitch; name_len = local whitespace hit or -1.
        if (lead) {
            c.boff[k] = 0; c.llen[k] = 0; c.elen[k] = 0; c.dlen[k] = -1;
            c.name_len[k] = (!full_name && ws >= 0) ? (int32_t)(ws - (h + 1 - x.gbase)) : -1;
        }
        return;
    }
    const int elen = cr ? 2 : 1;                           // index.c:266-269
    const int dlen = (int)(e - (h - x.gbase)) - elen;      // index.c:271
    int name_len = dlen;                                   // index.c:289-293: cut at ' ' or '\t'
    if (!full_name && ws >= 0 && ws - (h + 1 - x.gbase) < dlen) name_len = (int)(ws - (h + 1 - x.gbase));
    // first sequence line (index.c:330-332): the line after the header line, unless that is a header too
    int64_t llen = 0;
    if (e + 1 < x.n && x.data[e + 1] != '>') {
        scan_line_wave(x.data, x.n, e + 1, false, &e1, &dummy, &cr1);
        if (e1 < 0 && is_last) e1 = x.n;
        if (e1 >= 0) llen = e1 - e;
    }
    if (lead) {
        c.boff[k] = x.gbase + e + 1;                       // index.c:258  start = position
        c.llen[k] = llen;
        c.elen[k] = elen; c.dlen[k] = dlen; c.name_len[k] = name_len;
    }
}

// first set bit of the 4096-bit mask M (16 bits per 16-byte chunk, in LDS) at a granule position > pos and < limit; -1 if none
__device__ __forceinline__ int lds_next(const uint16_t *M, int pos, int limit) {
    const int p = pos + 1;
    if (p >= limit) return -1;
    int q = p >> 4;
    uint32_t w = (uint32_t)M[q] & (0xFFFFu << (p & 15)) & 0xFFFFu;
    while (!w) {
        if (++q >= GRAN / CHUNK || q * CHUNK >= limit) return -1;
        w = M[q];
    }
    const int r = q * CHUNK + __ffs(w) - 1;
    return r < limit ? r : -1;
}
__device__ __forceinline__ int lds_bit(const uint16_t *M, int pos) { return (M[pos >> 4] >> (pos & 15)) & 1; }

// One wave per granule that holds a header line: re-read its 4 KiB once, build the exact masks of
// '\n', header starts, white space, '\r' and '>', write every header's offset and the number of stream
// newlines before it straight to their final, position-ordered slots (hdr_prefix[g] + rank), then fill the
// columns that depend only on the record's first two lines (blen / slen / norm: k_fasta_finalize2) -- ONE LANE PER
// HEADER LINE: the masks go to LDS (2 KiB per wave) and every lane that owns a header finds the end of its
// line, of the line after it and the first white space by walking those masks (a file of short records has a
// dozen headers per granule; doing them one after the other with wave-wide ballots was 3.3 ms for 5 M records).
// A header whose first two lines run past the granule falls back to reading memory (record_at, whole wave).
__global__ __launch_bounds__(BLOCK) void k_hdr_rec(ScanCtx x, int prev_byte, int is_last, int full_name, GranList hgl,
                                                  FastaCols c, int64_t cap) {
    __shared__ uint16_t lds_all[BLOCK / 64][4][GRAN / CHUNK];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    uint16_t *NL = lds_all[wv][0], *WS = lds_all[wv][1], *CR = lds_all[wv][2], *GT = lds_all[wv][3];
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t cnt = *hgl.count;
    // the granule of the NEXT pass is requested as soon as this one's bytes have been turned into masks, so its
    // load (and the list lookup in front of it) travels while the header lines are worked on
    int64_t i = wave, g = i < cnt ? hgl.g[i] : 0;
    uint4 v[GR_ROWS];
    if (i < cnt) {
#pragma unroll
        for (int j = 0; j < GR_ROWS; ++j) v[j] = load16(x.data, g * (int64_t)GRAN + j * 1024 + lane * CHUNK, x.n);
    }
    for (; i < cnt; i += nwaves) {
        const int64_t sbase = g * (int64_t)GRAN;
        if (is_last && sbase + GRAN > x.n) {
#pragma unroll
            for (int j = 0; j < GR_ROWS; ++j) {
                const int64_t p = sbase + j * 1024 + lane * CHUNK;
                if (x.n >= p && x.n < p + CHUNK && x.n > 0 && x.data[x.n - 1] != '\n') {
                    const int k = (int)(x.n - p);
                    const uint32_t b = 0x0Au << ((k & 3) * 8);
                    if ((k >> 2) == 0) v[j].x |= b; else if ((k >> 2) == 1) v[j].y |= b; else if ((k >> 2) == 2) v[j].z |= b; else v[j].w |= b;
                }
            }
        }
        uint32_t hmask[GR_ROWS];
        int rfirst[GR_ROWS];                                // record of the lane's first header of row j, relative to hrank0
        int64_t hrank = x.hdr_prefix[g], nlb = x.nl_prefix[g];
        const int64_t hrank0 = hrank;
#pragma unroll
        for (int j = 0; j < GR_ROWS; ++j) {
            const int64_t p = sbase + j * 1024 + lane * CHUNK;
            const uint32_t nl = eq_mask16(v[j], 0x0A0A0A0Au);
            const uint32_t hm = header_mask16(v[j], nl, x.data, p, prev_byte);
            hmask[j] = hm;
            NL[j * 64 + lane] = (uint16_t)nl;
            WS[j * 64 + lane] = (uint16_t)(full_name ? 0u : (eq_mask16(v[j], 0x20202020u) | eq_mask16(v[j], 0x09090909u)));
            CR[j * 64 + lane] = (uint16_t)eq_mask16(v[j], 0x0D0D0D0Du);
            GT[j * 64 + lane] = (uint16_t)eq_mask16(v[j], 0x3E3E3E3Eu);
            const uint32_t cn = __popc(nl), ch = __popc(hm);
            const uint32_t in = wave_incl_scan(cn), ih = wave_incl_scan(ch);
            int64_t r = hrank + ih - ch;
            rfirst[j] = (int)(r - hrank0);
            const int64_t nb0 = nlb + in - cn;
            uint32_t hh = hm;
            while (hh) {
                const int k = __ffs(hh) - 1;
                hh &= hh - 1;
                if (r < cap) { c.hoff[r] = x.gbase + p + k; c.hdr_line[r] = nb0 + __popc(nl & ((1u << k) - 1u)); c.bad[r] = 0; }
                ++r;
            }
            nlb += (uint32_t)__shfl((int)in, 63, 64);
            hrank += (uint32_t)__shfl((int)ih, 63, 64);
        }
        // (sbase keeps addressing this granule below; g moves on to the next one now)
        if (i + nwaves < cnt) {
            g = hgl.g[i + nwaves];
#pragma unroll
            for (int j = 0; j < GR_ROWS; ++j) v[j] = load16(x.data, g * (int64_t)GRAN + j * 1024 + lane * CHUNK, x.n);
        }
        // records: every lane takes its own header lines, lowest first; the loop runs as long as any lane has one left
#pragma unroll
        for (int j = 0; j < GR_ROWS; ++j) {
            uint32_t hh = hmask[j];
            int rr = rfirst[j];
            while (__ballot(hh != 0)) {
                const bool has = hh != 0 && hrank0 + rr < cap;
                const int hp = j * 1024 + lane * CHUNK + (hh ? __ffs(hh) - 1 : 0);
                bool fallback = false;
                if (has) {
                    const int e = lds_next(NL, hp, GRAN);                                     // end of the header line
                    const int e1 = (e >= 0 && e + 1 < GRAN) ? lds_next(NL, e, GRAN) : -1;     // end of the line after it
                    const bool inside = e >= 0 && e + 1 < GRAN && (e1 >= 0 || sbase + e + 1 >= x.n);
                    if (!inside) fallback = true;                                             // runs past the granule: read memory
                    else {
                        const int elen = lds_bit(CR, e - 1) ? 2 : 1;                          // index.c:266-269 (e - 1 >= hp)
                        const int dlen = e - hp - elen;                                       // index.c:271
                        int name_len = dlen;                                                  // index.c:289-293: cut at ' ' or '\t'
                        if (!full_name) { const int w = lds_next(WS, hp, e); if (w >= 0 && w - (hp + 1) < dlen) name_len = w - (hp + 1); }
                        // first sequence line (index.c:330-332): the line after the header line, unless that is a header too
                        int64_t llen = 0;
                        if (sbase + e + 1 < x.n && !lds_bit(GT, e + 1) && e1 >= 0) llen = e1 - e;
                        const int64_t r = hrank0 + rr;
                        c.boff[r] = x.gbase + sbase + e + 1;                                  // index.c:258  start = position
                        c.llen[r] = llen; c.elen[r] = elen; c.dlen[r] = dlen; c.name_len[r] = name_len;
                    }
                }
                unsigned long long fb = __ballot(fallback);
                while (fb) {                                                                  // wave-uniform from here
                    const int l = __ffsll(fb) - 1;
                    fb &= fb - 1;
                    const int hq = (int)rdlane((uint32_t)hp, l), rq = (int)rdlane((uint32_t)rr, l);
                    record_at(x, is_last, full_name, x.gbase + sbase + hq, hrank0 + rq, c);
                }
                if (hh) { hh &= hh - 1; ++rr; }
            }
        }
    }
}

// ============================================================== bad_line (index.c:325-327)
struct RecView { const int64_t *boff, *llen; const int32_t *dlen; uint32_t *bad; const int64_t *hdr; };

// Walk the newlines of one granule exactly, all 64 lanes.  For the newline at p with predecessor q (in
// the lane, in a lower lane, in an earlier row, or prevnl[g]) the record is the last header <= p
// (headers of this granule are hdr[hdr_prefix[g] .. hdr_prefix[g+1])); the header line and the first
// sequence line are skipped, any other line with p - q != llen counts.
// v: the granule's bytes (requested by the caller); on return v is free again -- the caller requests the next granule
// into it before the rows are walked (exact_rows), which only needs the masks.
struct ExactMasks { uint32_t nl[GR_ROWS], hm[GR_ROWS]; };
__device__ __forceinline__ void exact_masks(const ScanCtx &x, uint4 (&v)[GR_ROWS], int is_last, int prev_byte, int64_t g, bool has_hdr,
                                            ExactMasks &mk) {
    const int lane = lane_id();
    const int64_t sbase = g * (int64_t)GRAN;
    if (is_last && sbase + GRAN > x.n) {
#pragma unroll
        for (int j = 0; j < GR_ROWS; ++j) {
            const int64_t p0 = sbase + j * 1024 + lane * CHUNK;
            if (x.n >= p0 && x.n < p0 + CHUNK && x.n > 0 && x.data[x.n - 1] != '\n') {
                const int k = (int)(x.n - p0);
                const uint32_t b = 0x0Au << ((k & 3) * 8);
                if ((k >> 2) == 0) v[j].x |= b; else if ((k >> 2) == 1) v[j].y |= b; else if ((k >> 2) == 2) v[j].z |= b; else v[j].w |= b;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < GR_ROWS; ++j) {
        mk.nl[j] = eq_mask16(v[j], 0x0A0A0A0Au);
        mk.hm[j] = has_hdr ? header_mask16(v[j], mk.nl[j], x.data, sbase + j * 1024 + lane * CHUNK, prev_byte) : 0u;
    }
}
__device__ __forceinline__ void exact_rows(const ScanCtx &x, const RecView &rv, int64_t cap, int64_t g, int64_t hb, const ExactMasks &mk) {
    const int lane = lane_id();
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int64_t sbase = g * (int64_t)GRAN, gs = x.gbase + sbase;
    int64_t carry = x.prevnl[g];                           // wave-uniform: latest newline so far (global, -1 none)
    const int64_t r0 = hb - 1;                             // record that owns the first byte of the granule
    const bool ok0 = r0 >= 0 && r0 < cap;
    const int32_t d0 = ok0 ? rv.dlen[r0] : -1;
    const int64_t e0 = ok0 ? rv.boff[r0] - 1 : 0, ll0 = ok0 ? rv.llen[r0] : 0;
    // The record of a newline at p is the last header line <= p: hdr_prefix[g] - 1 + the header lines of this granule
    // before p, counted from the exact header masks (wave prefix per row) -- no search in the header array.
    int64_t hseen = 0;                                     // wave-uniform: header lines in the rows already done
#pragma unroll
    for (int j = 0; j < GR_ROWS; ++j) {
        uint32_t m = mk.nl[j];
        const uint32_t hm = mk.hm[j];
        const int cb = j * 1024 + lane * CHUNK;
        const uint32_t ch = __popc(hm), ih = wave_incl_scan(ch);
        const int64_t hbefore = hseen + ih - ch;           // header lines of the granule before this lane's chunk
        hseen += (uint32_t)__shfl((int)ih, 63, 64);
        const unsigned long long bal = __ballot(m != 0);
        if (!bal) continue;
        const int pl = cb + (31 - __clz(m | 1u));           // last newline of this lane (granule-local)
        const unsigned long long lower = bal & lt;
        const int pprev = __builtin_amdgcn_ds_bpermute((63 - __clzll(lower | 1ull)) << 2, pl);
        int64_t q = lower ? gs + pprev : carry;
        while (m) {
            const int k = __ffs(m) - 1;
            m &= m - 1;
            const int64_t p = gs + cb + k;
            const int64_t r = hb - 1 + hbefore + __popc(hm & ((1u << k) - 1u));
            if (r == r0) {                                  // still in the record that owns the granule start
                if (d0 >= 0 && p != e0 && p != e0 + ll0 && q >= 0 && p - q != ll0) atomicAdd(&rv.bad[r0], 1u);
            } else if (r >= 0 && r < cap && rv.dlen[r] >= 0) {
                const int64_t e = rv.boff[r] - 1, ll = rv.llen[r];
                if (p != e && p != e + ll && q >= 0 && p - q != ll) atomicAdd(&rv.bad[r], 1u);
            }
            q = p;
        }
        carry = gs + (int)rdlane((uint32_t)pl, 63 - __clzll(bal));
    }
}

// One thread per granule.  A granule that lies inside one record's body past its first sequence line,
// without a header line and with <= 2 distinct line lengths, is answered from its summary: lines that
// differ from the record's llen = those of the two summarised lengths that differ, plus the line that
// crosses into the granule.  Anything else is irregular and goes to a list; k_gran_exact walks those
// granules, one wave each (they cluster where records are short, so they are spread over the chip
// instead of being walked by the wave that found them).
__global__ __launch_bounds__(BLOCK) void k_gran_lines(ScanCtx x, RecView rv, int64_t cap, GranList irr) {
    const int64_t g = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= x.ngran) return;
    const GranOut o = gran_unpack(x.go[g]);
    if (!o.n) return;
    const int64_t r = x.hdr_prefix[g] - 1;                 // record that owns the first byte of the granule
    const int64_t pv = x.prevnl[g];
    if (!o.h) {
        if (r < 0 || r >= cap || rv.dlen[r] < 0) return;   // r < 0: before the first header -- the shard summary handles the lead
        const int64_t ll = rv.llen[r];                     // 0: no sequence line, the only newline after the header ends it
        const int64_t e1 = rv.boff[r] - 1 + ll;            // end of the first sequence line
        const int64_t gs = x.gbase + g * (int64_t)GRAN;
        if (!ll || e1 >= gs + GRAN) return;                // nothing but the header line can end here
        if (e1 < gs && !o.ovf) {
            DiffSet ds;
            ds.v1 = o.v1; ds.c1 = o.c1; ds.v2 = o.v2; ds.c2 = o.c2; ds.ovf = 0;
            const uint32_t mism = ds.count_ne((uint32_t)ll) + ((gs + o.first - pv) != ll ? 1u : 0u);
            if (mism) atomicAdd(&rv.bad[r], mism);
            return;
        }
    }
    irr.g[atomicAdd(irr.count, 1u)] = (uint32_t)g;
}

__global__ __launch_bounds__(BLOCK) void k_gran_exact(ScanCtx x, RecView rv, int64_t cap, GranList irr, int is_last, int prev_byte) {
    const int lane = lane_id();
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t cnt = *irr.count;
    int64_t i = wave, g = i < cnt ? irr.g[i] : 0;
    uint4 v[GR_ROWS];
    if (i < cnt) {
#pragma unroll
        for (int j = 0; j < GR_ROWS; ++j) v[j] = load16(x.data, g * (int64_t)GRAN + j * 1024 + lane * CHUNK, x.n);
    }
    for (; i < cnt; i += nwaves) {
        const int64_t gc = g, hb = x.hdr_prefix[gc];
        int64_t he = x.hdr_prefix[gc + 1];
        if (he > cap) he = cap;
        ExactMasks mk;
        exact_masks(x, v, is_last, prev_byte, gc, he > hb, mk);
        if (i + nwaves < cnt) {                            // the next granule travels while this one's rows are walked
            g = irr.g[i + nwaves];
#pragma unroll
            for (int j = 0; j < GR_ROWS; ++j) v[j] = load16(x.data, g * (int64_t)GRAN + j * 1024 + lane * CHUNK, x.n);
        }
        exact_rows(x, rv, cap, gc, hb, mk);
    }
}

// blen, slen (index.c:243,335-338,348), norm (index.c:237,342), stat.seqlen (index.c:253-254, 360-369)
__global__ __launch_bounds__(BLOCK) void k_fasta_finalize2(int64_t cap, FastaCols c, Totals *__restrict__ tot,
                                                          const uint8_t *__restrict__ data, int64_t n_bytes, int64_t gbase) {
    const int64_t *__restrict__ hdr = c.hoff, *__restrict__ hdr_line = c.hdr_line;
    const int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t n_hdr = tot->n_hdr < cap ? tot->n_hdr : cap;
    int64_t s = 0;
    if (k < n_hdr) {
        if (c.dlen[k] >= 0) {
            int64_t hn, Ln;
            if (k + 1 < n_hdr) { hn = hdr[k + 1]; Ln = hdr_line[k + 1]; }
            else               { hn = tot->last_nl + 1; Ln = tot->n_nl; }     // EOF "position" (index.c:231)
            const int64_t nseq = Ln - hdr_line[k] - 1;     // sequence lines of this record
            const int64_t blen = hn - c.boff[k];
            s = blen - (int64_t)c.elen[k] * nseq;          // sum(line.l - line_end + 1)
            c.blen[k] = blen; c.slen[k] = s;
            const int norm = c.bad[k] > 1 ? 0 : 1;
            c.norm[k] = norm;
            // (the last record of a shard that is not the last one: provisional like its other columns, k_stitch_tail decides)
            int reg = line_regular(data, n_bytes, gbase, c.boff[k], blen, s, nseq > 0 ? c.llen[k] : 0, c.elen[k], norm);
            if (reg < 0) reg = c.bad[k] == 0;              // byte not held here: no odd line at all is regular whatever it says
            c.reg[k] = reg;
        } else { c.blen[k] = 0; c.slen[k] = 0; c.norm[k] = 1; c.reg[k] = 0; }
    }
    // one atomic per workgroup (per wave it was 78 k atomics on one address for 5 M records: 0.95 ms)
    __shared__ unsigned long long blk;
    if (threadIdx.x == 0) blk = 0;
    __syncthreads();
    s = wave_sum64(s);
    if (lane_id() == 0 && s) atomicAdd(&blk, (unsigned long long)s);
    __syncthreads();
    if (threadIdx.x == 0 && blk) atomicAdd((unsigned long long *)&tot->seq_len, blk);
}

// the line-regular column for a table that did not come from a scan (fx_fasta_set_table: rows of an existing .fxi)
__global__ __launch_bounds__(BLOCK) void k_line_regular(const uint8_t *__restrict__ data, int64_t n_bytes, int64_t gbase, int64_t n,
                                                       const int64_t *__restrict__ boff, const int64_t *__restrict__ blen,
                                                       const int64_t *__restrict__ slen, const int64_t *__restrict__ llen,
                                                       const int32_t *__restrict__ elen, const int32_t *__restrict__ norm,
                                                       int32_t *__restrict__ reg) {
    const int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (k < n) reg[k] = line_regular(data, n_bytes, gbase, boff[k], blen[k], slen[k], llen[k], elen[k], norm[k]) > 0 ? 1 : 0;
}

// last newline of the shard at a global offset < p (p wave-uniform), -1 if none: the granule that holds byte p - 1 is
// read by the whole wave (64 bytes per lane), below it prevnl[] answers.  Bytes of the stream only (not the virtual one).
__device__ __forceinline__ int64_t prev_nl_wave(const ScanCtx &x, int64_t p) {
    const int64_t y = (p - x.gbase < x.n) ? p - x.gbase : x.n;       // local offsets < y are searched
    if (y <= 0) return -1;
    const int64_t g = (y - 1) / GRAN;
    const int64_t a = g * (int64_t)GRAN + lane_id() * 64;
    unsigned long long m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (a + i * CHUNK < y) m |= (unsigned long long)eq_mask16(load16(x.data, a + i * CHUNK, x.n), 0x0A0A0A0Au) << (16 * i);
    if (y - a < 64) m &= (y - a <= 0) ? 0ull : (~0ull >> (64 - (y - a)));
    const unsigned long long b = __ballot(m != 0);
    if (!b) return x.prevnl[g];
    const int l = 63 - __clzll(b);
    const int hi = __shfl((int)(m >> 32), l, 64), lo = __shfl((int)(m & 0xFFFFFFFFull), l, 64);
    const unsigned long long ml = ((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo;
    return x.gbase + g * (int64_t)GRAN + l * 64 + (63 - __clzll(ml));
}

// ============================================================== shard boundary summary
// One workgroup.  Field order = fx_shard_summary in include/fxgpu.h.  The "lead" of a shard is the run
// of lines before its first header: they belong to a record that started in an earlier shard, whose
// llen is unknown here.  Because only `bad_line > 1` matters (index.c:237), the lead is summarised by
// the same two-distinct-values set as a granule: merge of the lead granules' sets, the lines that
// cross between them, and an exact walk of the part of the first header's granule before that header.
constexpr int SUMM_BLOCK = 1024;       // the lead of a shard can be a whole chromosome (10^5 granules): many loads in flight
__device__ __forceinline__ DiffSet diffset_shfl_xor(const DiffSet &d, int m) {
    return DiffSet{(uint32_t)__shfl_xor((int)d.v1, m, 64), (uint32_t)__shfl_xor((int)d.c1, m, 64), (uint32_t)__shfl_xor((int)d.v2, m, 64),
                   (uint32_t)__shfl_xor((int)d.c2, m, 64), (uint32_t)__shfl_xor((int)d.ovf, m, 64)};
}
__global__ __launch_bounds__(SUMM_BLOCK) void k_shard_summary2(ScanCtx x, int is_last, const Totals *__restrict__ tot, int64_t cap,
                                                              const int64_t *__restrict__ hdr, const int64_t *__restrict__ hdr_line,
                                                              FastaCols c, int64_t *__restrict__ S) {
    __shared__ DiffSet sets[SUMM_BLOCK / 64];
    __shared__ unsigned long long ws;
    const int tid = threadIdx.x;
    if (tid == 0) ws = ~0ull;
    __syncthreads();
    const int64_t n_hdr = tot->n_hdr, n_nl = tot->n_nl;
    const int64_t first_nl = next_nl(x, x.gbase - 1);
    const int64_t first_hdr = n_hdr ? hdr[0] : -1;
    // whitespace in the bytes before the first newline (a header line cut by the shard boundary)
    // (all of them, however long the line: a name that ends a megabyte behind the cut is found like any other; a thread stops
    // once somebody has found white space in front of it)
    const int64_t lim = first_nl >= 0 ? first_nl - x.gbase : x.n;
    for (int64_t j = tid; j < lim; j += SUMM_BLOCK) {
        if ((j & (64 * SUMM_BLOCK - 1)) < SUMM_BLOCK && *(volatile unsigned long long *)&ws < (unsigned long long)j) break;
        if (x.data[j] == ' ' || x.data[j] == '\t') { atomicMin(&ws, (unsigned long long)j); break; }
    }
    // lead: whole granules before the first header's granule, four independent loads per thread per step
    const int64_t g_first = n_hdr ? (first_hdr - x.gbase) / GRAN : x.ngran;
    DiffSet ds;
    ds.clear();
    for (int64_t g0 = tid; g0 < g_first; g0 += 4 * SUMM_BLOCK) {
        GranPk pk[4];
        int64_t pv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t g = g0 + u * SUMM_BLOCK;
            pk[u] = g < g_first ? x.go[g] : GranPk{0, 0, 0, 0};
            pv[u] = g < g_first ? x.prevnl[g] : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const GranOut o = gran_unpack(pk[u]);
            if (!o.n) continue;
            ds.add(o.v1, o.c1); ds.add(o.v2, o.c2); ds.ovf |= o.ovf;
            if (pv[u] >= 0) ds.add((uint32_t)(x.gbase + (g0 + u * SUMM_BLOCK) * (int64_t)GRAN + o.first - pv[u]), 1);
        }
    }
    // merge: butterfly inside the wave (the two-value set is a commutative, associative merge), then 16 wave results
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        const DiffSet o = diffset_shfl_xor(ds, m);
        ds.add(o.v1, o.c1); ds.add(o.v2, o.c2); ds.ovf |= o.ovf;
    }
    if ((tid & 63) == 0) sets[tid >> 6] = ds;
    __syncthreads();
    if (tid >= 64) return;
    // wave 0: the part of granule g_first before the header, 64 bytes per lane (four 16-byte loads, newline masks)
    DiffSet mine;
    mine.clear();
    int64_t lf = -1, ll = -1;                              // first / last newline seen by this lane (global)
    if (n_hdr) {
        const int64_t a = g_first * (int64_t)GRAN + tid * 64, b = first_hdr - x.gbase;
        unsigned long long m = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (a + i * CHUNK < b) m |= (unsigned long long)eq_mask16(load16(x.data, a + i * CHUNK, x.n), 0x0A0A0A0Au) << (16 * i);
        if (b - a < 64) m &= (b - a <= 0) ? 0ull : (~0ull >> (64 - (b - a)));       // only bytes before the header
        while (m) {
            const int k = __ffsll(m) - 1;
            m &= m - 1;
            const int64_t gp = x.gbase + a + k;
            if (ll >= 0) mine.add((uint32_t)(gp - ll), 1);
            if (lf < 0) lf = gp;
            ll = gp;
        }
    }
    DiffSet all;
    all.clear();
    if (tid == 0)
        for (int i = 0; i < SUMM_BLOCK / 64; ++i) { const DiffSet d = sets[i]; all.add(d.v1, d.c1); all.add(d.v2, d.c2); all.ovf |= d.ovf; }
    int64_t carry = n_hdr ? x.prevnl[g_first] : -1;
    for (int l = 0; l < 64; ++l) {                         // lane order = position order
        const DiffSet d{(uint32_t)__shfl((int)mine.v1, l, 64), (uint32_t)__shfl((int)mine.c1, l, 64), (uint32_t)__shfl((int)mine.v2, l, 64),
                        (uint32_t)__shfl((int)mine.c2, l, 64), (uint32_t)__shfl((int)mine.ovf, l, 64)};
        const int64_t f = ((int64_t)__shfl((int)(lf >> 32), l, 64) << 32) | (unsigned int)__shfl((int)(lf & 0xFFFFFFFFll), l, 64);
        const int64_t la = ((int64_t)__shfl((int)(ll >> 32), l, 64) << 32) | (unsigned int)__shfl((int)(ll & 0xFFFFFFFFll), l, 64);
        if (tid == 0 && f >= 0) {
            if (carry >= 0) all.add((uint32_t)(f - carry), 1);
            all.add(d.v1, d.c1); all.add(d.v2, d.c2); all.ovf |= d.ovf;
            carry = la;
        }
    }
    // the second-to-last newline of the lead and of the whole shard: what a record that ENDS here or in the next shard
    // needs to know the length of its last line (k_stitch_tail: line-regular test across the cut)
    const int64_t lead_nl = n_hdr ? hdr_line[0] : n_nl;
    const int64_t lead_last = n_hdr ? first_hdr - 1 : tot->last_nl;
    const int64_t lead_prev = lead_nl >= 2 ? prev_nl_wave(x, lead_last) : -1;
    const int64_t second_last = n_nl >= 2 ? prev_nl_wave(x, tot->last_nl) : -1;
    if (tid != 0) return;
    S[0] = x.gbase; S[1] = x.n; S[2] = is_last;
    S[3] = n_nl; S[4] = first_nl; S[5] = first_nl >= 0 ? next_nl(x, first_nl) : -1; S[6] = n_nl ? tot->last_nl : -1;
    S[7] = (first_nl > x.gbase) ? (int64_t)x.data[first_nl - 1 - x.gbase] : -1;
    S[8] = x.data[0]; S[9] = x.data[x.n - 1];
    S[10] = n_hdr; S[11] = first_hdr; S[12] = n_hdr ? hdr[(n_hdr < cap ? n_hdr : cap) - 1] : -1;
    S[13] = lead_nl;
    S[14] = (ws == ~0ull) ? -1 : x.gbase + (int64_t)ws;
    // v1 must be the first distance of the lead when there is one (stitch_tail reads second_nl - first_nl
    // separately; the pair order itself does not matter to count_ne)
    S[15] = all.c1 ? all.v1 : 0; S[16] = all.c1; S[17] = all.c2 ? all.v2 : 0; S[18] = all.c2;
    int64_t te = -1, tfe = -1, tna = 0, tbad = 0, telen = 0, tdlen = -1, tname = -1;
    if (n_hdr && n_hdr <= cap) {
        const int64_t k = n_hdr - 1;
        tdlen = c.dlen[k]; tname = c.name_len[k];
        if (tdlen >= 0) {
            te = c.boff[k] - 1; telen = c.elen[k];
            tna = n_nl - hdr_line[k] - 1;
            if (tna > 0) tfe = te + c.llen[k];
            tbad = c.bad[k];
        }
    }
    S[19] = te; S[20] = tfe; S[21] = tna; S[22] = tbad; S[23] = telen; S[24] = tdlen; S[25] = tname;
    S[26] = lead_prev; S[27] = second_last;
}

// ============================================================== stitch (multi-GPU)
// After the all-gather every rank holds the boundary summaries of all shards (world x 28 words, field
// order = fx_shard_summary).  One thread finishes the last record that starts in shard r -- the only
// one that can run past the cut -- with pure integer logic (index.c:234-353 across shard cuts) and
// rewrites that row of the resident table.  Same logic as pyfastx_amd/shard.py:stitch_tail, which the
// CPU tests exercise; this kernel keeps the whole sharded build on the device (no host round trip
// between the collective and the fetches).
enum { SS_BASE = 0, SS_NBYTES, SS_ISLAST, SS_NNL, SS_FIRSTNL, SS_SECONDNL, SS_LASTNL, SS_FIRSTNLPREV, SS_FIRSTBYTE, SS_LASTBYTE,
       SS_NHDR, SS_FIRSTHDR, SS_LASTHDR, SS_LEADNL, SS_LEADWS, SS_V1, SS_C1, SS_V2, SS_C2, SS_TE, SS_TFE, SS_TNA, SS_TBAD,
       SS_TELEN, SS_TDLEN, SS_TNAME, SS_LEADPREVNL, SS_SECONDLASTNL, SS_WORDS = 28 };

// min(2, number of FULL lead lines of shard u whose len+1 != llen)
__device__ __forceinline__ int64_t lead_count_ne(const int64_t *u, int64_t llen) {
    const int64_t full = u[SS_LEADNL] - 1;
    if (full <= 0) return 0;
    if (u[SS_C1] + u[SS_C2] < full) return 2;            // >= 3 distinct lengths: at least two differ from any llen
    const int64_t eq = (u[SS_V1] == llen ? u[SS_C1] : 0) + ((u[SS_C2] && u[SS_V2] == llen) ? u[SS_C2] : 0);
    return full - eq < 2 ? full - eq : 2;
}

__global__ void k_stitch_tail(const int64_t *__restrict__ S, int world, int r, int full_name, FastaCols c, int64_t cap,
                              int *__restrict__ err) {
    if (threadIdx.x || blockIdx.x) return;
    const int64_t *s = S + (int64_t)r * SS_WORDS;
    if (s[SS_NHDR] == 0 || s[SS_NHDR] > cap) return;
    const int64_t k = s[SS_NHDR] - 1;
    const int64_t h = s[SS_LASTHDR];
    int64_t e = s[SS_TE], elen = s[SS_TELEN], dlen = s[SS_TDLEN], name_len = s[SS_TNAME];
    bool have_e = e >= 0, have_llen = false;
    int64_t llen = 0, nseq = 0, bad = 0, last_nl = s[SS_LASTNL];
    if (have_e) {
        nseq = s[SS_TNA];
        if (s[SS_TFE] >= 0) { llen = s[SS_TFE] - e; have_llen = true; bad = s[SS_TBAD] < 2 ? s[SS_TBAD] : 2; }
    }
    int64_t hn = -1, ws = -1;                             // ws: white space seen in continuation shards while the header is unterminated
    int t_hdr = -1;                                       // the shard that holds the next header line
    for (int t = r + 1; t < world; ++t) {
        const int64_t *u = S + (int64_t)t * SS_WORDS;
        if (!have_e && ws < 0 && u[SS_LEADWS] >= 0) ws = u[SS_LEADWS];
        if (u[SS_LEADNL] > 0) {
            const int64_t first = u[SS_FIRSTNL], full = u[SS_LEADNL] - 1;
            if (!have_e) {                                // the header line itself crossed the cut
                const int64_t pb = u[SS_FIRSTNLPREV] >= 0 ? u[SS_FIRSTNLPREV] : (u - SS_WORDS)[SS_LASTBYTE];
                e = first;
                elen = pb == 13 ? 2 : 1;                  // index.c:266-269
                dlen = (e - h) - elen;                    // index.c:271
                if (full_name) name_len = dlen;
                else if (name_len < 0) {
                    name_len = (ws >= 0 && ws < e) ? ws - (h + 1) : dlen;
                }
                if (name_len > dlen) name_len = dlen;
                have_e = true;
                if (full >= 1) { llen = u[SS_SECONDNL] - u[SS_FIRSTNL]; have_llen = true; bad += lead_count_ne(u, llen); }
                nseq += full;
            } else if (!have_llen) {                      // the first sequence line crossed the cut
                llen = first - e; have_llen = true;
                nseq += u[SS_LEADNL];
                bad += lead_count_ne(u, llen);
            } else {                                      // an ordinary line crossed the cut
                bad += (first - last_nl != llen) ? 1 : 0;
                nseq += u[SS_LEADNL];
                bad += lead_count_ne(u, llen);
            }
            if (bad > 2) bad = 2;
            if (u[SS_NHDR] == 0) last_nl = u[SS_LASTNL];
        }
        if (u[SS_NHDR] > 0) { hn = u[SS_FIRSTHDR]; t_hdr = t; break; }
    }
    if (hn < 0) {                                         // `position` after the last line (index.c:231)
        hn = 0;
        for (int t = world - 1; t >= 0; --t) { const int64_t *u = S + (int64_t)t * SS_WORDS; if (u[SS_NNL] > 0) { hn = u[SS_LASTNL] + 1; break; } }
    }
    const int64_t boff = e + 1, blen = hn - boff;         // index.c:243,348
    const int64_t slen = blen - elen * nseq;
    if (nseq <= 0) llen = 0;
    // line-regular (fx_kernels.hpp: line_regular): the byte that decides it may sit on another rank, so the same
    // question is put to the summaries -- is the last line of the record (the two newlines before hn) x + elen long?
    int reg = 0;
    const int64_t bpl = llen - elen;
    if (bad <= 1 && bpl > 0 && elen > 0) {
        if (bad == 0 || slen <= bpl) reg = 1;
        else {
            const int64_t lines = (slen + bpl - 1) / bpl;
            if (lines == nseq) {
                const int64_t x = slen - (lines - 1) * bpl;
                int found = 0;
                int64_t prev = -1;
                for (int t = t_hdr >= 0 ? t_hdr : world - 1; t >= r && found < 2; --t) {
                    const int64_t *u = S + (int64_t)t * SS_WORDS;
                    const bool lead = t == t_hdr;         // only the newlines before its first header line count there
                    const int64_t cnt = lead ? u[SS_LEADNL] : u[SS_NNL];
                    if (cnt <= 0) continue;
                    if (found == 0) { if (cnt >= 2) { prev = lead ? u[SS_LEADPREVNL] : u[SS_SECONDLASTNL]; found = 2; } else found = 1; }
                    else { prev = lead ? u[SS_FIRSTHDR] - 1 : u[SS_LASTNL]; found = 2; }
                }
                reg = (found == 2 && (hn - 1) - prev == x + elen) ? 1 : 0;
            }
        }
    }
    c.boff[k] = boff; c.blen[k] = blen; c.slen[k] = slen; c.llen[k] = llen;
    c.elen[k] = (int32_t)elen; c.norm[k] = bad > 1 ? 0 : 1; c.dlen[k] = (int32_t)dlen; c.name_len[k] = (int32_t)name_len;
    c.reg[k] = reg;
}

// fx_fasta_set_row: one launch instead of eight 4/8-byte copies
__global__ void k_set_row(FastaCols c, int64_t k, int64_t boff, int64_t blen, int64_t slen, int64_t llen, int32_t elen,
                          int32_t norm, int32_t dlen, int32_t name_len, int32_t reg) {
    c.boff[k] = boff; c.blen[k] = blen; c.slen[k] = slen; c.llen[k] = llen;
    c.elen[k] = elen; c.norm[k] = norm & 1; c.dlen[k] = dlen; c.name_len[k] = name_len; c.reg[k] = reg;
}

}  // namespace fx
