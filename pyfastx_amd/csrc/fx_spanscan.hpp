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

// per-granule summary written by phase 1 (40 bytes, 8-byte aligned)
struct GranOut { uint32_t n, h, first, last, v1, c1, v2, c2, ovf, pad; };

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
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

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

template <bool FULL>
__device__ __forceinline__ void granule(const uint8_t *__restrict__ data, int64_t n, int prev_byte, int is_last,
                                        int64_t g, GranOut *__restrict__ out, const GranList &hgl, uint32_t &L) {
    const int lane = lane_id();
    const int64_t sbase = g * (int64_t)GRAN;

    uint4 v[GR_ROWS];
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

    WaveDiff wd;
    wd.clear();
    uint32_t n_w = 0, h_w = 0;                  // wave-uniform counts
    int first_w = -1, carry = -1;               // wave-uniform: first / latest newline (granule-local)

#pragma unroll
    for (int j = 0; j < GR_ROWS; ++j) {
        const uint32_t cb = j * 1024 + lane * CHUNK;
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
        if (__builtin_expect(__ballot(maybe_gt16(v[j])) != 0, 0)) {
            if (!nlm) nlm = flags4(t0) | (flags4(t1) << 4) | (flags4(t2) << 8) | (flags4(t3) << 12);
            h_w += wave_sum_small(__popc(header_mask16(v[j], nlm, data, sbase + cb, prev_byte)));
        }
    }
    if (lane == 0) {
        GranOut o;
        o.n = n_w; o.h = h_w; o.first = (uint32_t)first_w; o.last = (uint32_t)carry;
        o.v1 = wd.v1; o.c1 = wd.c1; o.v2 = wd.v2; o.c2 = wd.c2; o.ovf = wd.ovf; o.pad = 0;
        out[g] = o;
        if (h_w) hgl.g[atomicAdd(hgl.count, 1u)] = (uint32_t)g;
    }
}

// grid-stride over granules, one wave per granule per step.  FULL = true covers the n / GRAN granules
// that lie entirely inside the stream; the last, partial one (which also holds the virtual end-of-stream
// newline) is a single-wave launch of the FULL = false instantiation, so its bounds-checked loads
// cost the main kernel neither registers nor branches.
template <bool FULL>
__global__ __launch_bounds__(1024) void k_span_scan(const uint8_t *__restrict__ data, int64_t n, int prev_byte,
                                                    int is_last, int64_t g_begin, int64_t g_end,
                                                    GranOut *__restrict__ out, GranList hgl) {
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    uint32_t L = 0;
    for (int64_t g = g_begin + wave; g < g_end; g += nwaves) granule<FULL>(data, n, prev_byte, is_last, g, out, hgl, L);
}

// ============================================================== header collection
// One wave per granule that holds a header line (hgl): re-read its 4 KiB, exact masks, and write
// every header's offset and the number of stream newlines before it straight to their final,
// position-ordered slots: hdr[hdr_prefix[g] + rank], hdr_line[...] = nl_prefix[g] + newlines before.
__global__ __launch_bounds__(BLOCK) void k_hdr_collect(const uint8_t *__restrict__ data, int64_t n, int64_t gbase,
                                                      int prev_byte, int is_last, GranList hgl,
                                                      const int64_t *__restrict__ nl_prefix,
                                                      const int64_t *__restrict__ hdr_prefix,
                                                      int64_t *__restrict__ hdr, int64_t *__restrict__ hdr_line, int64_t cap) {
    const int lane = lane_id();
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t cnt = *hgl.count;
    for (int64_t i = wave; i < cnt; i += nwaves) {
        const int64_t g = hgl.g[i];
        const int64_t sbase = g * (int64_t)GRAN;
        int64_t hrank = hdr_prefix[g], nlb = nl_prefix[g];
        for (int j = 0; j < GR_ROWS; ++j) {
            const int64_t p = sbase + j * 1024 + lane * CHUNK;
            uint4 v = load16(data, p, n);
            if (is_last && n >= p && n < p + CHUNK && n > 0 && data[n - 1] != '\n') {
                const int k = (int)(n - p);
                const uint32_t b = 0x0Au << ((k & 3) * 8);
                if ((k >> 2) == 0) v.x |= b; else if ((k >> 2) == 1) v.y |= b; else if ((k >> 2) == 2) v.z |= b; else v.w |= b;
            }
            const uint32_t nlm = eq_mask16(v, 0x0A0A0A0Au);
            uint32_t hm = header_mask16(v, nlm, data, p, prev_byte);
            const uint32_t cn = __popc(nlm), ch = __popc(hm);
            const uint32_t in = wave_incl_scan(cn), ih = wave_incl_scan(ch);
            int64_t r = hrank + ih - ch;
            const int64_t nb0 = nlb + in - cn;
            while (hm) {
                const int k = __ffs(hm) - 1;
                hm &= hm - 1;
                if (r < cap) { hdr[r] = gbase + p + k; hdr_line[r] = nb0 + __popc(nlm & ((1u << k) - 1u)); }
                ++r;
            }
            nlb += (uint32_t)__shfl((int)in, 63, 64);
            hrank += (uint32_t)__shfl((int)ih, 63, 64);
        }
    }
}

// ============================================================== granule prefixes
// Exclusive prefixes over the granule summaries, three small kernels (no atomics, deterministic):
//   k_gran_reduce   one workgroup per CHUNK of 256 granules (1 MiB of stream): totals
//   k_chunk_scan    one workgroup: exclusive scan of the chunk totals (+ grand totals)
//   k_gran_prefix   one workgroup per chunk: per-granule prefixes = chunk base + local scan
// nl_prefix[g] / hdr_prefix[g] = newlines / header lines before granule g (entry [ngran] = totals),
// prevnl[g] = global offset of the last newline before granule g (-1: none in this shard).
constexpr int CHUNK_GRANS = 256;
struct ChunkTot { long long n, h, last, pad; };
struct Totals {                       // device-side scalars of one build, copied to the host once at the end
    long long n_nl, n_hdr, last_nl, seq_len, n_hdr_gran, n_irregular, pad0, pad1;
};

__device__ __forceinline__ long long shfl_xor64(long long v, int d) {
    const int lo = __shfl_xor((int)(v & 0xFFFFFFFFll), d, 64), hi = __shfl_xor((int)(v >> 32), d, 64);
    return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ long long shfl_up64(long long v, int d) {
    const int lo = __shfl_up((int)(v & 0xFFFFFFFFll), d, 64), hi = __shfl_up((int)(v >> 32), d, 64);
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
// inclusive scan of v over the threads of the workgroup (<= 1024 threads); *total = join of all
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
__device__ __forceinline__ Tri gran_tri(const GranOut *__restrict__ go, int64_t g, int64_t ngran, int64_t gbase) {
    if (g >= ngran) return Tri{0, 0, -1};
    const GranOut o = go[g];
    return Tri{(long long)o.n, (long long)o.h, o.n ? gbase + g * (long long)GRAN + o.last : -1};
}

__global__ __launch_bounds__(CHUNK_GRANS) void k_gran_reduce(const GranOut *__restrict__ go, int64_t ngran, int64_t gbase,
                                                            ChunkTot *__restrict__ ct) {
    __shared__ Tri lds[CHUNK_GRANS / 64];
    Tri tot;
    tri_block_incl(gran_tri(go, (int64_t)blockIdx.x * CHUNK_GRANS + threadIdx.x, ngran, gbase), lds, &tot);
    if (threadIdx.x == 0) ct[blockIdx.x] = ChunkTot{tot.n, tot.h, tot.last, 0};
}

// in place: ct[c] becomes the join of the chunks before c; tot gets the grand totals
__global__ __launch_bounds__(1024) void k_chunk_scan(ChunkTot *__restrict__ ct, int64_t nchunks, Totals *__restrict__ tot) {
    __shared__ Tri lds[16];
    __shared__ Tri carry_s;
    if (threadIdx.x == 0) carry_s = Tri{0, 0, -1};
    __syncthreads();
    for (int64_t c0 = 0; c0 < nchunks; c0 += 1024) {
        const int64_t c = c0 + threadIdx.x;
        Tri v{0, 0, -1};
        if (c < nchunks) { const ChunkTot t = ct[c]; v = Tri{t.n, t.h, t.last}; }
        Tri total;
        const Tri inc = tri_block_incl(v, lds, &total);
        const Tri carry = carry_s;
        // exclusive = carry + (inclusive of the previous thread); recover it from the inclusive value of lane-1
        Tri prev{shfl_up64(inc.n, 1), shfl_up64(inc.h, 1), shfl_up64(inc.last, 1)};
        if (lane_id() == 0) {            // first lane of a wave: inclusive value of the last lane of the previous wave
            prev = Tri{0, 0, -1};
            for (int i = 0; i < (int)(threadIdx.x >> 6); ++i) prev = tri_join(prev, lds[i]);
        }
        const Tri ex = tri_join(carry, prev);
        if (c < nchunks) ct[c] = ChunkTot{ex.n, ex.h, ex.last, 0};
        __syncthreads();
        if (threadIdx.x == 0) carry_s = tri_join(carry, total);
        __syncthreads();
    }
    if (threadIdx.x == 0) { const Tri t = carry_s; tot->n_nl = t.n; tot->n_hdr = t.h; tot->last_nl = t.last; }
}

__global__ __launch_bounds__(CHUNK_GRANS) void k_gran_prefix(const GranOut *__restrict__ go, int64_t ngran, int64_t gbase,
                                                            const ChunkTot *__restrict__ ct, const Totals *__restrict__ tot,
                                                            int64_t *__restrict__ nl_prefix, int64_t *__restrict__ hdr_prefix,
                                                            int64_t *__restrict__ prevnl) {
    __shared__ Tri lds[CHUNK_GRANS / 64];
    const int64_t g = (int64_t)blockIdx.x * CHUNK_GRANS + threadIdx.x;
    const Tri v = gran_tri(go, g, ngran, gbase);
    Tri total;
    const Tri inc = tri_block_incl(v, lds, &total);
    const ChunkTot b = ct[blockIdx.x];
    if (g < ngran) {
        nl_prefix[g] = b.n + inc.n - v.n;
        hdr_prefix[g] = b.h + inc.h - v.h;
        // exclusive max: the inclusive value of the previous thread
        long long pm = shfl_up64(inc.last, 1);
        if (lane_id() == 0) { pm = -1; for (int i = 0; i < (int)(threadIdx.x >> 6); ++i) pm = lds[i].last > pm ? lds[i].last : pm; }
        if (threadIdx.x == 0) pm = -1;
        prevnl[g] = b.last > pm ? b.last : pm;
    }
    if (g == ngran - 1) { nl_prefix[ngran] = tot->n_nl; hdr_prefix[ngran] = tot->n_hdr; prevnl[ngran] = tot->last_nl; }
}

// ============================================================== navigation by granule
struct ScanCtx {
    const uint8_t *data; int64_t n, gbase, ngran;
    const GranOut *go;
    const int64_t *nl_prefix, *hdr_prefix, *prevnl;      // [ngran + 1]
};

// first newline at a global offset > x (x >= gbase - 1), or -1 when the shard holds none.  The virtual
// end-of-stream newline (offset gbase + n) counts: it is in the last granule's summary.
__device__ __forceinline__ int64_t next_nl(const ScanCtx &c, int64_t x) {
    const int64_t y = x + 1 - c.gbase;                    // local offset of the first candidate byte
    if (y > c.n) return -1;
    const int64_t g = y / GRAN;
    if (g < c.ngran) {
        const GranOut o = c.go[g];
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
    return c.gbase + lo * (int64_t)GRAN + c.go[lo].first;
}

// ============================================================== record table
// One thread per header: the columns of index.c:234-339 from the header offset, its line index,
// and the two newlines that follow it (end of the header line, end of the first sequence line).
__global__ __launch_bounds__(BLOCK) void k_fasta_rec2(ScanCtx x, const Totals *__restrict__ tot, int64_t cap,
                                                     const int64_t *__restrict__ hdr, const int64_t *__restrict__ hdr_line,
                                                     int full_name, FastaCols c) {
    const int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t n_hdr = tot->n_hdr < cap ? tot->n_hdr : cap;
    if (k >= n_hdr) return;
    const int64_t h = hdr[k], L = hdr_line[k];
    const int64_t e = next_nl(x, h);                       // newline that ends the header line
    if (e < 0) {
        // only possible for the LAST header of a non-final shard: its line ends in a later shard.
        // Leave a stub (dlen = -1) for the host-side stitch; name_len = local whitespace hit or -1.
        int nlen = -1;
        if (!full_name) {
            const int64_t lim = x.n - (h + 1 - x.gbase);
            const uint8_t *s = x.data + (h + 1 - x.gbase);
            for (int64_t j = 0; j < lim; ++j) if (s[j] == ' ' || s[j] == '\t') { nlen = (int)j; break; }
        }
        c.boff[k] = 0; c.blen[k] = 0; c.slen[k] = 0; c.llen[k] = 0;
        c.elen[k] = 0; c.dlen[k] = -1; c.name_len[k] = nlen; c.bad[k] = 0;
        return;
    }
    const int64_t boff = e + 1;                            // index.c:258  start = position
    const int elen = (x.data[e - 1 - x.gbase] == '\r') ? 2 : 1;   // index.c:266-269
    const int dlen = (int)(e - h) - elen;                  // index.c:271
    int name_len = dlen;
    if (!full_name) {                                      // index.c:289-293: cut at ' ' or '\t'
        const uint8_t *s = x.data + (h + 1 - x.gbase);
        for (name_len = 0; name_len < dlen; ++name_len)
            if (s[name_len] == ' ' || s[name_len] == '\t') break;
    }
    int64_t hn, Ln;
    if (k + 1 < tot->n_hdr && k + 1 < cap) { hn = hdr[k + 1]; Ln = hdr_line[k + 1]; }
    else                                   { hn = tot->last_nl + 1; Ln = tot->n_nl; }   // EOF "position" (index.c:231)
    const int64_t nseq = Ln - L - 1;                       // sequence lines of this record
    const int64_t blen = hn - boff;                        // index.c:243,348
    int64_t llen = 0;
    if (nseq > 0) llen = next_nl(x, e) - e;                // first line length + 1, index.c:330-332
    c.boff[k] = boff; c.blen[k] = blen;
    c.slen[k] = blen - (int64_t)elen * nseq;               // sum(line.l - line_end + 1), index.c:335-338
    c.llen[k] = llen;
    c.elen[k] = elen; c.dlen[k] = dlen; c.name_len[k] = name_len;
    c.bad[k] = 0;
}

// ============================================================== bad_line (index.c:325-327)
// One thread per granule.  A granule that lies inside one record's body past its first sequence line,
// without a header line and with <= 2 distinct line lengths, is answered from its summary: lines that
// differ from the record's llen = those of the two summarised lengths that differ, plus the line that
// crosses into the granule.  Everything else goes to the irregular list for k_gran_exact.
struct RecView { const int64_t *boff, *llen; const int32_t *dlen; uint32_t *bad; };

__global__ __launch_bounds__(BLOCK) void k_gran_lines(ScanCtx x, RecView rv, int64_t cap, GranList irr) {
    const int64_t g = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= x.ngran) return;
    const GranOut o = x.go[g];
    if (!o.n) return;
    const int64_t r = x.hdr_prefix[g] - 1;                 // record that owns the first byte of the granule
    if (o.h == 0) {
        if (r < 0 || r >= cap) return;                     // before the first header: the shard summary handles the lead
        if (rv.dlen[r] < 0) return;
        const int64_t ll = rv.llen[r];
        if (!ll) return;                                   // no sequence line: the only newline after the header ends it
        const int64_t e1 = rv.boff[r] - 1 + ll;            // end of the first sequence line
        const int64_t gs = x.gbase + g * (int64_t)GRAN;
        if (e1 >= gs + GRAN) return;                       // nothing but the header line can end here
        if (e1 < gs && !o.ovf) {
            DiffSet ds;
            ds.v1 = o.v1; ds.c1 = o.c1; ds.v2 = o.v2; ds.c2 = o.c2; ds.ovf = 0;
            const uint32_t mism = ds.count_ne((uint32_t)ll) + ((gs + o.first - x.prevnl[g]) != ll ? 1u : 0u);
            if (mism) atomicAdd(&rv.bad[r], mism);
            return;
        }
    }
    irr.g[atomicAdd(irr.count, 1u)] = (uint32_t)g;
}

// One wave per irregular granule: walk its newlines exactly.  For the newline at p with predecessor q
// (in the lane, in a lower lane, in an earlier row, or prevnl[g]) the record is the last header <= p
// (headers of this granule are hdr[hdr_prefix[g] .. hdr_prefix[g+1])); the header line and the first
// sequence line are skipped, any other line with p - q != llen counts.
__global__ __launch_bounds__(BLOCK) void k_gran_exact(ScanCtx x, RecView rv, int64_t cap, GranList irr, int is_last,
                                                     const int64_t *__restrict__ hdr) {
    const int lane = lane_id();
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t cnt = *irr.count;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int64_t i = wave; i < cnt; i += nwaves) {
        const int64_t g = irr.g[i];
        const int64_t sbase = g * (int64_t)GRAN;
        const int64_t hb = x.hdr_prefix[g];
        int64_t he = x.hdr_prefix[g + 1];
        if (he > cap) he = cap;
        int64_t carry = x.prevnl[g];                       // wave-uniform: latest newline so far (global, -1 none)
        for (int j = 0; j < GR_ROWS; ++j) {
            const int64_t p0 = sbase + j * 1024 + lane * CHUNK;
            uint4 v = load16(x.data, p0, x.n);
            if (is_last && x.n >= p0 && x.n < p0 + CHUNK && x.n > 0 && x.data[x.n - 1] != '\n') {
                const int k = (int)(x.n - p0);
                const uint32_t b = 0x0Au << ((k & 3) * 8);
                if ((k >> 2) == 0) v.x |= b; else if ((k >> 2) == 1) v.y |= b; else if ((k >> 2) == 2) v.z |= b; else v.w |= b;
            }
            uint32_t m = eq_mask16(v, 0x0A0A0A0Au);
            const unsigned long long bal = __ballot(m != 0);
            if (!bal) continue;
            const int64_t pl = x.gbase + p0 + (31 - __clz(m | 1u));          // last newline of this lane
            const unsigned long long lower = bal & lt;
            const int prevlane = 63 - __clzll(lower | 1ull);
            const int plo = __builtin_amdgcn_ds_bpermute(prevlane << 2, (int)(pl & 0xFFFFFFFFll));
            const int phi = __builtin_amdgcn_ds_bpermute(prevlane << 2, (int)(pl >> 32));
            int64_t q = lower ? (((int64_t)phi << 32) | (unsigned int)plo) : carry;
            while (m) {
                const int k = __ffs(m) - 1;
                m &= m - 1;
                const int64_t p = x.gbase + p0 + k;
                // record of p: headers of this granule that start before p
                int64_t lo = hb, hi = he;
                while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (hdr[mid] < p) lo = mid + 1; else hi = mid; }
                const int64_t r = lo - 1;
                if (r >= 0 && r < cap && rv.dlen[r] >= 0) {
                    const int64_t e = rv.boff[r] - 1, ll = rv.llen[r];
                    if (p != e && p != e + ll && q >= 0 && p - q != ll) atomicAdd(&rv.bad[r], 1u);
                }
                q = p;
            }
            const int last_lane = 63 - __clzll(bal);
            carry = ((int64_t)__builtin_amdgcn_readlane((int)(pl >> 32), last_lane) << 32) |
                    (unsigned int)__builtin_amdgcn_readlane((int)(pl & 0xFFFFFFFFll), last_lane);
        }
    }
}

// norm (index.c:237,342), stat.seqlen (index.c:253-254, 360-369)
__global__ __launch_bounds__(BLOCK) void k_fasta_finalize2(const uint32_t *__restrict__ bad, const int64_t *__restrict__ slen,
                                                          int64_t cap, int32_t *__restrict__ norm, Totals *__restrict__ tot) {
    const int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t n_hdr = tot->n_hdr < cap ? tot->n_hdr : cap;
    int64_t s = 0;
    if (k < n_hdr) { norm[k] = bad[k] > 1 ? 0 : 1; s = slen[k]; }
    s = wave_sum64(s);
    if (lane_id() == 0 && s) atomicAdd((unsigned long long *)&tot->seq_len, (unsigned long long)s);
}

// ============================================================== shard boundary summary
// One workgroup.  Field order = fx_shard_summary in include/fxgpu.h.  The "lead" of a shard is the run
// of lines before its first header: they belong to a record that started in an earlier shard, whose
// llen is unknown here.  Because only `bad_line > 1` matters (index.c:237), the lead is summarised by
// the same two-distinct-values set as a granule: merge of the lead granules' sets, the lines that
// cross between them, and an exact walk of the part of the first header's granule before that header.
__global__ __launch_bounds__(BLOCK) void k_shard_summary2(ScanCtx x, int is_last, const Totals *__restrict__ tot, int64_t cap,
                                                         const int64_t *__restrict__ hdr, const int64_t *__restrict__ hdr_line,
                                                         FastaCols c, int64_t *__restrict__ S) {
    __shared__ DiffSet sets[BLOCK];
    __shared__ unsigned long long ws;
    const int tid = threadIdx.x;
    if (tid == 0) ws = ~0ull;
    __syncthreads();
    const int64_t n_hdr = tot->n_hdr, n_nl = tot->n_nl;
    const int64_t first_nl = next_nl(x, x.gbase - 1);
    const int64_t first_hdr = n_hdr ? hdr[0] : -1;
    // whitespace in the bytes before the first newline (a header line cut by the shard boundary)
    int64_t lim = first_nl >= 0 ? first_nl - x.gbase : x.n;
    if (lim > 65536) lim = 65536;
    for (int64_t j = tid; j < lim; j += BLOCK)
        if (x.data[j] == ' ' || x.data[j] == '\t') { atomicMin(&ws, (unsigned long long)j); break; }
    // lead: whole granules before the first header's granule
    const int64_t g_first = n_hdr ? (first_hdr - x.gbase) / GRAN : x.ngran;
    DiffSet ds;
    ds.clear();
    for (int64_t g = tid; g < g_first; g += BLOCK) {
        const GranOut o = x.go[g];
        if (!o.n) continue;
        ds.add(o.v1, o.c1); ds.add(o.v2, o.c2); ds.ovf |= o.ovf;
        const int64_t pv = x.prevnl[g];
        if (pv >= 0) ds.add((uint32_t)(x.gbase + g * (int64_t)GRAN + o.first - pv), 1);
    }
    sets[tid] = ds;
    __syncthreads();
    if (tid >= 64) return;
    // wave 0: the part of granule g_first before the header, 64 bytes per lane
    DiffSet mine;
    mine.clear();
    int64_t lf = -1, ll = -1;                              // first / last newline seen by this lane (global)
    if (n_hdr) {
        const int64_t a = g_first * (int64_t)GRAN + tid * 64, b = first_hdr - x.gbase;
        for (int64_t p = a; p < a + 64 && p < b; ++p)
            if (x.data[p] == '\n') {
                const int64_t gp = x.gbase + p;
                if (ll >= 0) mine.add((uint32_t)(gp - ll), 1);
                if (lf < 0) lf = gp;
                ll = gp;
            }
    }
    DiffSet all;
    all.clear();
    if (tid == 0) for (int i = 0; i < BLOCK; ++i) { const DiffSet d = sets[i]; all.add(d.v1, d.c1); all.add(d.v2, d.c2); all.ovf |= d.ovf; }
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
    if (tid != 0) return;
    const int64_t lead_nl = n_hdr ? hdr_line[0] : n_nl;
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
    S[26] = 0; S[27] = 0;
}

}  // namespace fx
