// fx_kernels.hpp -- hand-written HIP kernels for gfx950 (MI355X, wave64).
//
// The reference (lmdu/pyfastx) walks the file line by line on one CPU thread
// (kseq.c:59-109 feeding index.c:230-339 / fastq.c:89-149).  None of that
// structure survives here: the stream is resident in HBM and is processed as
//
//   K1 k_scan          bytes -> 1 bit/byte newline mask + per-tile counts      (HBM-bound, reads the file once)
//   K2 k_tile_scan     exclusive prefix over the per-tile counts               (tiny)
//   K3 k_linetable     newline mask -> int64 line table nl[]                   (reads n/8 bytes, writes 8 B/line)
//   K4 k_hdr_scatter   '>' at line start -> hdr[] (only tiles that have one)
//   K5 k_fasta_rec     one thread per record: gathers from nl[]/hdr[]          (index.c:234-339 columns)
//   K6 k_fasta_lines   one thread per line: bad-line count per record          (index.c:325-327)
//   K7 k_fetch         one wave per query: gather, despace, upper, revcomp     (index.c:683-707, util.c:157-269)
//   ... FASTQ and composition kernels below.
//
// Integer/byte work only: no MFMA anywhere; the roofline is HBM bandwidth.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fx {

constexpr int BLOCK = 256;                      // 4 waves of 64
constexpr int UNROLL = 8;                       // 16-byte loads in flight per lane
constexpr int CHUNK = 16;                       // bytes per lane per load (global_load_dwordx4)
constexpr int TILE = BLOCK * CHUNK * UNROLL;    // 32 KiB of file per workgroup
constexpr int TILE_CHUNKS = BLOCK * UNROLL;     // 2048 mask words (u16) per tile

// ---------------------------------------------------------------- SWAR bytes
// 0x80 in every byte of x that is zero (exact, no borrow false positives).
__device__ __forceinline__ uint32_t zero_bytes(uint32_t x) {
    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}
// gather the four 0x80 flags of a word into bits 0..3
__device__ __forceinline__ uint32_t flags4(uint32_t t) {
    return (((t >> 7) & 0x01010101u) * 0x00204081u >> 21) & 0xFu;
}
__device__ __forceinline__ uint32_t eq_mask16(const uint4 &v, uint32_t pat) {
    return flags4(zero_bytes(v.x ^ pat)) | (flags4(zero_bytes(v.y ^ pat)) << 4) |
           (flags4(zero_bytes(v.z ^ pat)) << 8) | (flags4(zero_bytes(v.w ^ pat)) << 12);
}
__device__ __forceinline__ uint32_t any_eq16(const uint4 &v, uint32_t pat) {
    return zero_bytes(v.x ^ pat) | zero_bytes(v.y ^ pat) | zero_bytes(v.z ^ pat) | zero_bytes(v.w ^ pat);
}

// 16 bytes at data[p..p+16); bytes at or beyond n read as 0 (never '\n' or '>').
__device__ __forceinline__ uint4 load16(const uint8_t *__restrict__ data, int64_t p, int64_t n) {
    if (p + CHUNK <= n) return *reinterpret_cast<const uint4 *>(data + p);
    uint32_t w[4] = {0, 0, 0, 0};
    for (int k = 0; k < CHUNK; ++k)
        if (p + k < n) w[k >> 2] |= (uint32_t)data[p + k] << ((k & 3) * 8);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// ------------------------------------------------------------- wave / block
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (l >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ int64_t wave_sum64(int64_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
// exclusive prefix of v over the 256 threads of the block (thread order); *total = block sum.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *lds4, uint32_t *total) {
    const int w = threadIdx.x >> 6, l = lane_id();
    uint32_t inc = wave_incl_scan(v);
    __syncthreads();                       // protect lds4 reuse across calls
    if (l == 63) lds4[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < BLOCK / 64; ++i) { uint32_t s = lds4[i]; if (i < w) base += s; tot += s; }
    *total = tot;
    return base + inc - v;
}
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t *lds4) {
    const int w = threadIdx.x >> 6, l = lane_id();
    v = wave_sum(v);
    __syncthreads();
    if (l == 0) lds4[w] = v;
    __syncthreads();
    return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

// first index i in [0,n) with a[i] >= key  (n if none)
__device__ __forceinline__ int64_t lower_bound(const int64_t *__restrict__ a, int64_t n, int64_t key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}
// first index i in [0,n) with a[i] > key  (n if none)
__device__ __forceinline__ int64_t upper_bound(const int64_t *__restrict__ a, int64_t n, int64_t key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (a[mid] <= key) lo = mid + 1; else hi = mid; }
    return lo;
}

// ======================================================================= K1
// Delimiter scan.  One workgroup per 32 KiB tile; every lane issues UNROLL
// independent 16-byte loads (a wave covers 1 KiB contiguous per instruction),
// turns each into a 16-bit "byte == '\n'" mask with SWAR arithmetic, stores the
// mask (2 B/lane, coalesced) and counts.  With HDR the lane also tests for '>'
// and, only when one is present (rare outside header lines), checks the byte
// before it: a FASTA header is a '>' that follows '\n' or starts the stream
// (index.c:234, line.s[0] == 62).
// Replaces: ks_getuntil's byte loop kseq.c:78-80 and the memcpy kseq.c:94.
template <bool HDR>
__global__ __launch_bounds__(BLOCK) void k_scan(const uint8_t *__restrict__ data, int64_t n, int prev_byte,
                                               uint16_t *__restrict__ nlmask, uint32_t *__restrict__ tile_nl,
                                               uint32_t *__restrict__ tile_hdr) {
    __shared__ uint32_t lds4[4];
    const int64_t tile = blockIdx.x;
    const int64_t tbase = tile * (int64_t)TILE;
    const int tid = threadIdx.x;
    uint4 v[UNROLL];
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) v[j] = load16(data, tbase + (int64_t)(j * BLOCK + tid) * CHUNK, n);
    uint32_t cnt = 0, hcnt = 0;
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
        const uint32_t m = eq_mask16(v[j], 0x0A0A0A0Au);
        nlmask[tile * TILE_CHUNKS + j * BLOCK + tid] = (uint16_t)m;
        cnt += __popc(m);
        if (HDR) {
            if (any_eq16(v[j], 0x3E3E3E3Eu)) {
                uint32_t g = eq_mask16(v[j], 0x3E3E3E3Eu);
                const int64_t p = tbase + (int64_t)(j * BLOCK + tid) * CHUNK;
                while (g) {
                    const int k = __ffs(g) - 1;
                    g &= g - 1;
                    const int64_t pos = p + k;
                    const int prev = pos ? (int)data[pos - 1] : prev_byte;
                    hcnt += (prev == '\n');
                }
            }
        }
    }
    const uint32_t tot = block_sum(cnt, lds4);
    if (tid == 0) tile_nl[tile] = tot;
    if (HDR) {
        const uint32_t htot = block_sum(hcnt, lds4);
        if (tid == 0) tile_hdr[tile] = htot;
    }
}

// ======================================================================= K2
// Exclusive prefix sums over the per-tile counts (one workgroup; the arrays are
// ~n/32768 entries).  off[i] = sum(cnt[0..i)), off[ntiles] = total.
__global__ __launch_bounds__(1024) void k_tile_scan(const uint32_t *__restrict__ cnt, int64_t ntiles,
                                                   int64_t *__restrict__ off) {
    __shared__ int64_t part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (ntiles + 1023) / 1024;
    const int64_t lo = (int64_t)tid * per, hi = (lo + per < ntiles) ? lo + per : ntiles;
    int64_t s = 0;
    for (int64_t i = lo; i < hi; ++i) s += cnt[i];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {          // Hillis-Steele inclusive scan in LDS
        int64_t t = (tid >= d) ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += t;
        __syncthreads();
    }
    int64_t run = part[tid] - s;
    for (int64_t i = lo; i < hi; ++i) { off[i] = run; run += cnt[i]; }
    if (tid == 1023) off[ntiles] = part[1023];
}

// ======================================================================= K3
// Newline mask -> line table.  One workgroup per tile; each lane takes 8
// consecutive 16-bit masks (one 16-byte load = 128 bytes of file), the block
// does an exclusive scan of the popcounts, and every lane writes the GLOBAL
// offsets of its newlines at nl[tile_off + rank].
// nl[i] is the offset of the '\n' that terminates line i; it carries
// `position += line.l + 1` (index.c:231, fastq.c:148) for every line at once.
__global__ __launch_bounds__(BLOCK) void k_linetable(const uint16_t *__restrict__ nlmask,
                                                    const int64_t *__restrict__ tile_off, int64_t gbase,
                                                    int64_t *__restrict__ nl) {
    __shared__ uint32_t lds4[4];
    const int64_t tile = blockIdx.x;
    const int tid = threadIdx.x;
    const uint4 mv = *reinterpret_cast<const uint4 *>(nlmask + tile * TILE_CHUNKS + tid * 8);
    const uint32_t w[4] = {mv.x, mv.y, mv.z, mv.w};
    const uint32_t cnt = __popc(w[0]) + __popc(w[1]) + __popc(w[2]) + __popc(w[3]);
    uint32_t total;
    uint32_t r = block_excl_scan(cnt, lds4, &total);
    if (cnt == 0) return;
    int64_t *dst = nl + tile_off[tile] + r;
    const int64_t p0 = gbase + tile * (int64_t)TILE + (int64_t)tid * 8 * CHUNK;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t m = w[q];                        // two 16-bit masks = 32 bytes of file, bit i = byte i
        while (m) {
            const int k = __ffs(m) - 1;
            m &= m - 1;
            *dst++ = p0 + q * 32 + k;
        }
    }
}

// ======================================================================= K4
// Header offsets.  Tiles without a header exit immediately (for a genome that
// is all but ~n_seq of them); the others are re-read row by row in position
// order so hdr[] comes out sorted.
__global__ __launch_bounds__(BLOCK) void k_hdr_scatter(const uint8_t *__restrict__ data, int64_t n, int prev_byte,
                                                      const uint32_t *__restrict__ tile_hdr,
                                                      const int64_t *__restrict__ tile_hdr_off, int64_t gbase,
                                                      int64_t *__restrict__ hdr) {
    __shared__ uint32_t lds4[4];
    const int64_t tile = blockIdx.x;
    if (tile_hdr[tile] == 0) return;
    const int tid = threadIdx.x;
    const int64_t tbase = tile * (int64_t)TILE;
    int64_t run = tile_hdr_off[tile];
    for (int j = 0; j < UNROLL; ++j) {
        const int64_t p = tbase + (int64_t)(j * BLOCK + tid) * CHUNK;
        const uint4 v = load16(data, p, n);
        uint32_t g = eq_mask16(v, 0x3E3E3E3Eu), hm = 0;
        while (g) {
            const int k = __ffs(g) - 1;
            g &= g - 1;
            const int64_t pos = p + k;
            const int prev = pos ? (int)data[pos - 1] : prev_byte;
            if (prev == '\n') hm |= 1u << k;
        }
        uint32_t total;
        uint32_t r = block_excl_scan(__popc(hm), lds4, &total);
        while (hm) {
            const int k = __ffs(hm) - 1;
            hm &= hm - 1;
            hdr[run + r++] = gbase + p + k;
        }
        run += total;
    }
}

// ======================================================================= K5
// FASTA record table: one thread per header.  Everything is a gather from the
// line table; column semantics follow index.c:234-339 exactly, including the
// quirks (elen taken from the header line only, index.c:266-269; blen/boff in
// "position" units that over-count by one when the stream lacks a final '\n',
// index.c:231 -- the virtual newline appended to nl[] reproduces that).
//   n_nl counts the virtual EOF newline when present; all offsets are global,
//   data is indexed with (offset - gbase).
struct FastaCols {
    int64_t *hoff, *boff, *blen, *slen, *llen, *hdr_line;
    int32_t *elen, *dlen, *name_len;
    uint32_t *bad;
};

__global__ __launch_bounds__(BLOCK) void k_fasta_rec(const uint8_t *__restrict__ data, int64_t gbase, int64_t n_bytes,
                                                    const int64_t *__restrict__ nl, int64_t n_nl,
                                                    const int64_t *__restrict__ hdr, int64_t n_hdr, int full_name,
                                                    FastaCols c) {
    const int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n_hdr) return;
    const int64_t h = hdr[k];
    const int64_t L = lower_bound(nl, n_nl, h);          // line index of the header line
    if (L >= n_nl) {
        // only possible for the LAST header of a non-final shard: its line ends in a later
        // shard.  Leave a stub (dlen = -1) for the host-side stitch; name_len = local
        // whitespace hit or -1.
        int nlen = -1;
        if (!full_name) {
            const int64_t lim = n_bytes - (h + 1 - gbase);
            const uint8_t *s = data + (h + 1 - gbase);
            for (int64_t j = 0; j < lim; ++j) if (s[j] == ' ' || s[j] == '\t') { nlen = (int)j; break; }
        }
        c.hoff[k] = h; c.boff[k] = 0; c.blen[k] = 0; c.slen[k] = 0; c.llen[k] = 0; c.hdr_line[k] = n_nl;
        c.elen[k] = 0; c.dlen[k] = -1; c.name_len[k] = nlen; c.bad[k] = 0;
        return;
    }
    const int64_t e = nl[L];                             // its terminating newline
    const int64_t boff = e + 1;                          // index.c:258  start = position
    const int elen = (data[e - 1 - gbase] == '\r') ? 2 : 1;   // index.c:266-269
    const int dlen = (int)(e - h) - elen;                // index.c:271
    int name_len = dlen;
    if (!full_name) {                                    // index.c:289-293: cut at ' ' or '\t'
        const uint8_t *s = data + (h + 1 - gbase);
        for (name_len = 0; name_len < dlen; ++name_len)
            if (s[name_len] == ' ' || s[name_len] == '\t') break;
    }
    int64_t hn, Ln;
    if (k + 1 < n_hdr) { hn = hdr[k + 1]; Ln = lower_bound(nl, n_nl, hn); }
    else               { hn = nl[n_nl - 1] + 1; Ln = n_nl; }       // EOF "position"
    const int64_t nseq = Ln - L - 1;                     // sequence lines of this record
    const int64_t blen = hn - boff;                      // index.c:243,348
    c.hoff[k] = h; c.boff[k] = boff; c.blen[k] = blen;
    c.slen[k] = blen - (int64_t)elen * nseq;             // sum(line.l - line_end + 1), index.c:335-338
    c.llen[k] = nseq > 0 ? nl[L + 1] - nl[L] : 0;        // first line length + 1, index.c:330-332
    c.hdr_line[k] = L;
    c.elen[k] = elen; c.dlen[k] = dlen; c.name_len[k] = name_len;
    c.bad[k] = 0;
}

// ======================================================================= K6
// bad_line (index.c:325-327): lines after the first of a record whose length
// differs from the first.  One thread per line; the record is found by binary
// search over hdr_line[] (wave-uniform in the common case of long records).
// Bad lines are rare in well-formed files (the short last line of each record),
// so the global atomic is rarely taken.
__global__ __launch_bounds__(BLOCK) void k_fasta_lines(const int64_t *__restrict__ nl, int64_t n_nl,
                                                      const int64_t *__restrict__ hdr_line, int64_t n_hdr,
                                                      const int64_t *__restrict__ llen, uint32_t *__restrict__ bad) {
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t i0 = (int64_t)blockIdx.x * BLOCK + (threadIdx.x & ~63); i0 < n_nl; i0 += stride) {
        const int64_t i = i0 + lane_id();
        // records of the first and last line of this wave's 64-line window
        const int64_t ilast = (i0 + 63 < n_nl) ? i0 + 63 : n_nl - 1;
        const int64_t r0 = upper_bound(hdr_line, n_hdr, i0) - 1;
        int64_t r1 = r0;
        if (r0 + 1 < n_hdr && hdr_line[r0 + 1] <= ilast) r1 = upper_bound(hdr_line, n_hdr, ilast) - 1;
        if (i >= n_nl) continue;
        int64_t rec = r0;
        if (r1 != r0) rec = r0 + upper_bound(hdr_line + (r0 + 1), r1 - r0, i);   // search inside [r0+1, r1]
        if (rec < 0) continue;                                  // before the first header
        const int64_t hl = hdr_line[rec];
        if (i <= hl + 1) continue;                              // header line or first sequence line
        if (nl[i] - nl[i - 1] != llen[rec]) atomicAdd(&bad[rec], 1u);
    }
}

// Shard lead statistics (multi-GPU stitch).  The "lead" of a shard is
// the run of lines before its first header: they belong to a record that
// started in an earlier shard, whose first-line length (llen) is unknown here.
// Because only `bad_line > 1` matters (index.c:237), the lead is summarised by
// its two first distinct line lengths and their counts: with <= 2 distinct
// values the owner can compute its bad-line count exactly, with >= 3 it is
// >= 2 whatever llen turns out to be.
//   lines considered: i in [1, lead_nl) (both delimiting newlines in the shard)
//   pass 0: v = nl[1]-nl[0];  out[0] += count(d == v), out[1] = min i with d != v
//   pass 1: v = d at out[1];  out[2] += count(d == v)
__global__ __launch_bounds__(BLOCK) void k_lead_stats(const int64_t *__restrict__ nl, int64_t lead_nl, int pass,
                                                     unsigned long long *__restrict__ out) {
    if (lead_nl < 2) return;
    int64_t v;
    if (pass == 0) v = nl[1] - nl[0];
    else {
        const int64_t j = (int64_t)out[1];
        if (j >= lead_nl) return;
        v = nl[j] - nl[j - 1];
    }
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    int64_t cnt = 0;
    unsigned long long first_ne = ~0ull;
    for (int64_t i = 1 + (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < lead_nl; i += stride) {
        const int64_t d = nl[i] - nl[i - 1];
        if (d == v) ++cnt;
        else if (first_ne == ~0ull) first_ne = (unsigned long long)i;
    }
    cnt = wave_sum64(cnt);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        unsigned long long o = __shfl_xor(first_ne, d, 64);
        first_ne = o < first_ne ? o : first_ne;
    }
    if (lane_id() == 0) {
        if (cnt) atomicAdd(&out[pass == 0 ? 0 : 2], (unsigned long long)cnt);
        if (pass == 0 && first_ne != ~0ull) atomicMin(&out[1], first_ne);
    }
}

// Collect the boundary summary of a shard into S[0..FX_SUMMARY_WORDS) (one
// workgroup; scalars by thread 0, the bounded whitespace search by all).
// Field order = fx_shard_summary in include/fxgpu.h.
constexpr int FX_SUMMARY_WORDS = 28;
__global__ __launch_bounds__(BLOCK) void k_shard_summary(const uint8_t *__restrict__ data, int64_t n, int64_t gbase,
                                                        int is_last, const int64_t *__restrict__ nl, int64_t n_nl,
                                                        const int64_t *__restrict__ hdr, int64_t n_hdr, FastaCols c,
                                                        const unsigned long long *__restrict__ stats, int64_t lead_nl,
                                                        int64_t *__restrict__ S) {
    __shared__ unsigned long long ws;
    if (threadIdx.x == 0) ws = ~0ull;
    __syncthreads();
    const int64_t first_nl = n_nl ? nl[0] : -1;
    int64_t lim = (first_nl >= 0 ? first_nl - gbase : n);
    if (lim > 65536) lim = 65536;
    for (int64_t j = threadIdx.x; j < lim; j += BLOCK)
        if (data[j] == ' ' || data[j] == '\t') { atomicMin(&ws, (unsigned long long)j); break; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    S[0] = gbase; S[1] = n; S[2] = is_last;
    S[3] = n_nl; S[4] = first_nl; S[5] = n_nl > 1 ? nl[1] : -1; S[6] = n_nl ? nl[n_nl - 1] : -1;
    S[7] = (first_nl > gbase) ? (int64_t)data[first_nl - 1 - gbase] : -1;
    S[8] = data[0]; S[9] = data[n - 1];
    S[10] = n_hdr; S[11] = n_hdr ? hdr[0] : -1; S[12] = n_hdr ? hdr[n_hdr - 1] : -1;
    S[13] = lead_nl;
    S[14] = (ws == ~0ull) ? -1 : gbase + (int64_t)ws;
    int64_t v1 = 0, c1 = 0, v2 = 0, c2 = 0;
    if (lead_nl >= 2) {
        v1 = nl[1] - nl[0]; c1 = (int64_t)stats[0];
        const int64_t j = (int64_t)stats[1];
        if (j < lead_nl) { v2 = nl[j] - nl[j - 1]; c2 = (int64_t)stats[2]; }
    }
    S[15] = v1; S[16] = c1; S[17] = v2; S[18] = c2;
    int64_t te = -1, tfe = -1, tna = 0, tbad = 0, telen = 0, tdlen = -1, tname = -1;
    if (n_hdr) {
        const int64_t k = n_hdr - 1;
        tdlen = c.dlen[k]; tname = c.name_len[k];
        if (tdlen >= 0) {
            const int64_t L = c.hdr_line[k];
            te = nl[L]; telen = c.elen[k];
            tna = n_nl - L - 1;
            if (L + 1 < n_nl) tfe = nl[L + 1];
            tbad = c.bad[k];
        }
    }
    S[19] = te; S[20] = tfe; S[21] = tna; S[22] = tbad; S[23] = telen; S[24] = tdlen; S[25] = tname;
    S[26] = 0; S[27] = 0;
}

// norm (index.c:237,342) and stat.seqlen (index.c:253-254, 360-369)
__global__ __launch_bounds__(BLOCK) void k_fasta_finalize(const uint32_t *__restrict__ bad,
                                                         const int64_t *__restrict__ slen, int64_t n_hdr,
                                                         int32_t *__restrict__ norm,
                                                         unsigned long long *__restrict__ seqlen_total) {
    const int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    int64_t s = 0;
    if (k < n_hdr) { norm[k] = bad[k] > 1 ? 0 : 1; s = slen[k]; }
    s = wave_sum64(s);
    if (lane_id() == 0 && s) atomicAdd(seqlen_total, (unsigned long long)s);
}

// ======================================================================= K7
// Batched fetch.  One wave per query: lanes read consecutive bytes of
// [off, off+blen), keep-mask = byte not in {10,13,32} (jump_table, util.c:157-164),
// __ballot + popcount of lower lanes gives each kept byte its rank, ranks in
// [skip, skip+take) are emitted at dst[rank-skip] (mirrored for FX_REVERSE),
// optionally upper-cased (Py_TOUPPER, util.c:181-194) and complemented through
// the IUPAC LUT in LDS (comp_map, util.c:228-237).
// Replaces per query: fseeko+fread (index.c:688-689), remove_space*
// (util.c:166-194), reverse/complement (util.c:239-269), memcpy
// (sequence.c:346-347).
struct FetchQ {
    const int64_t *off, *blen, *skip, *take;     // skip may be null (0)
    const int64_t *seq_id, *start, *stop;        // alternative: resolve against the FASTA table
    const uint8_t *qflags;                       // per-query flags or null
    const int64_t *dst_off;
    int64_t *out_len;
};
struct FastaTab { const int64_t *boff, *blen, *slen, *llen; const int32_t *elen, *norm; int64_t n_seq; };

__device__ __forceinline__ void build_comp_lut(uint8_t *lut) {
    // IUPAC complement (util.c:204-237): A<->T C<->G M<->K R<->Y V<->B H<->D, U->A, case kept,
    // W S N and everything else map to themselves.  Bytes >= 128: identity.
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        uint8_t c = (uint8_t)i, u = c & 0xDF, r = c;
        const bool letter = (u >= 'A' && u <= 'Z') && (c < 128);
        if (letter) {
            uint8_t m = u;
            switch (u) {
            case 'A': m = 'T'; break; case 'T': m = 'A'; break; case 'U': m = 'A'; break;
            case 'C': m = 'G'; break; case 'G': m = 'C'; break;
            case 'M': m = 'K'; break; case 'K': m = 'M'; break;
            case 'R': m = 'Y'; break; case 'Y': m = 'R'; break;
            case 'V': m = 'B'; break; case 'B': m = 'V'; break;
            case 'H': m = 'D'; break; case 'D': m = 'H'; break;
            default: break;
            }
            r = m | (c & 0x20);
        }
        lut[i] = r;
    }
}

template <bool BY_ID>
__global__ __launch_bounds__(BLOCK) void k_fetch(const uint8_t *__restrict__ data, int64_t gbase, int64_t n_bytes,
                                                FetchQ q, FastaTab tab, int64_t nq, int flags_all,
                                                uint8_t *__restrict__ dst) {
    __shared__ uint8_t lut[256];
    build_comp_lut(lut);
    __syncthreads();
    const int lane = lane_id();
    const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
    for (int64_t i = wave; i < nq; i += nwaves) {
        int64_t off, blen, skip, take;
        if (BY_ID) {
            const int64_t id = q.seq_id[i], a = q.start[i], b = q.stop[i];
            if (id < 0 || id >= tab.n_seq || a < 0 || b < a || b > tab.slen[id]) { // caller validates; stay safe
                if (lane == 0 && q.out_len) q.out_len[i] = -1;
                continue;
            }
            take = b - a;
            const int64_t bpl = tab.llen[id] - tab.elen[id];
            if (tab.norm[id] && bpl > 0) {                 // sequence.c:498-510
                const int64_t bs = a / bpl, be = b / bpl;
                off = tab.boff[id] + a + (int64_t)tab.elen[id] * bs;
                blen = take + (be - bs) * tab.elen[id];
                skip = 0;
            } else {                                       // sequence.c:100-110: despace whole record, then slice
                off = tab.boff[id]; blen = tab.blen[id]; skip = a;
            }
        } else {
            off = q.off[i]; blen = q.blen[i]; take = q.take[i]; skip = q.skip ? q.skip[i] : 0;
        }
        const int fl = q.qflags ? q.qflags[i] : flags_all;
        // clamp to the bytes we hold (fread past EOF returns short, index.c:689)
        int64_t lo = off - gbase, hi = lo + blen;
        if (lo < 0) lo = 0;
        if (hi > n_bytes) hi = n_bytes;
        uint8_t *out = dst + q.dst_off[i];
        int64_t rank = 0;                                  // kept bytes before this window
        const int64_t end = skip + take;
        for (int64_t p = lo; p < hi && rank < end; p += 64) {
            const int64_t pp = p + lane;
            uint8_t c = (pp < hi) ? data[pp] : (uint8_t)'\n';
            const bool keep = (fl & 8) ? (pp < hi) : !(c == 10 || c == 13 || c == 32);   // FX_RAW keeps every byte
            const unsigned long long bal = __ballot(keep);
            const int64_t r = rank + __popcll(bal & ((1ull << lane) - 1ull));
            if (keep && r >= skip && r < end) {
                if ((fl & 1) && c >= 'a' && c <= 'z') c -= 32;
                if (fl & 4) c = lut[c];
                const int64_t o = r - skip;
                out[(fl & 2) ? (take - 1 - o) : o] = c;
            }
            rank += __popcll(bal);
        }
        if (lane == 0 && q.out_len) {
            int64_t got = rank - skip;
            q.out_len[i] = got < 0 ? 0 : (got > take ? take : got);
        }
    }
}

// In-place reverse / complement of one buffer (pyfastx.reverse_complement, module.c:44-59).
__global__ __launch_bounds__(BLOCK) void k_revcomp(uint8_t *__restrict__ buf, int64_t n, int mode) {
    __shared__ uint8_t lut[256];
    build_comp_lut(lut);
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    const int64_t half = (mode & 2) ? (n + 1) / 2 : n;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < half; i += stride) {
        if (mode & 2) {
            const int64_t j = n - 1 - i;
            uint8_t a = buf[i], b = buf[j];
            if (mode & 4) { a = lut[a]; b = lut[b]; }
            buf[i] = b; buf[j] = a;
        } else if (mode & 4) {
            buf[i] = lut[buf[i]];
        }
    }
}

// ================================================================ FASTQ (K4')
// One thread per read k: the four lines of record k are nl[4k-1]+1 .. nl[4k+3]
// (fastq.c:89-149, `line_num % 4` state machine becomes a gather).  Also
// reduces stat.size (sum of rlen, including an incomplete trailing record's
// sequence line, fastq.c:125) and the min/max quality-line length for
// meta.maxlen/minlen (fastq.c:747-751).
struct FastqCols {
    int64_t *name_off, *rlen, *soff, *qoff;
    int32_t *name_len, *dlen;
};
struct FastqAcc {            // device accumulators
    unsigned long long size;
    unsigned long long a, c, g, t, n;
    long long maxlen, minlen;
    int minqs, maxqs;
};

__global__ __launch_bounds__(BLOCK) void k_fastq_rec(const uint8_t *__restrict__ data, int64_t gbase,
                                                    const int64_t *__restrict__ nl, int64_t n_nl, int64_t n_reads,
                                                    FastqCols c, FastqAcc *acc) {
    const int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    int64_t rl = 0;
    long long qmax = 0, qmin = 10000000000LL;
    if (4 * k + 1 < n_nl) {                               // sequence line exists
        const int64_t s0 = k ? nl[4 * k - 1] + 1 : gbase; // start of header line
        const int64_t e0 = nl[4 * k];
        const int64_t soff = e0 + 1, e1 = nl[4 * k + 1];
        const int64_t l1 = e1 - soff;
        rl = (l1 > 0 && data[e1 - 1 - gbase] == '\r') ? l1 - 1 : l1;     // fastq.c:124-128
        if (k < n_reads) {
            const int dlen = (int)(e0 - s0);              // fastq.c:103 (includes '@' and '\r')
            int64_t nlen = dlen - 1;
            if (nlen > 0 && data[e0 - 1 - gbase] == '\r') --nlen;        // fastq.c:107-109
            const uint8_t *s = data + (s0 + 1 - gbase);
            for (int64_t j = 0; j < nlen; ++j) if (s[j] == ' ') { nlen = j; break; }   // fastq.c:112-117
            const int64_t qoff = nl[4 * k + 2] + 1, e3 = nl[4 * k + 3];
            long long ql = e3 - qoff;
            if (ql > 0 && data[e3 - 1 - gbase] == '\r') --ql;            // fastq.c:734-737 (trailing CR)
            qmax = ql; qmin = ql;
            c.name_off[k] = s0 + 1; c.name_len[k] = (int32_t)nlen; c.dlen[k] = dlen;
            c.rlen[k] = rl; c.soff[k] = soff; c.qoff[k] = qoff;
        }
    }
    rl = wave_sum64(rl);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        long long a = __shfl_xor(qmax, d, 64), b = __shfl_xor(qmin, d, 64);
        qmax = a > qmax ? a : qmax; qmin = b < qmin ? b : qmin;
    }
    if (lane_id() == 0) {
        if (rl) atomicAdd(&acc->size, (unsigned long long)rl);
        atomicMax(&acc->maxlen, qmax);
        atomicMin(&acc->minlen, qmin);
    }
}

// FASTQ composition (fastq.c:715-753): one wave per read, lanes stride the
// sequence line (count A/C/G/T uppercase, '\r' ignored, everything else N) and
// the quality line (min/max byte, '\r' ignored).
__global__ __launch_bounds__(BLOCK) void k_fastq_comp(const uint8_t *__restrict__ data, int64_t gbase,
                                                     const int64_t *__restrict__ nl, int64_t n_nl,
                                                     int64_t n_lines4, FastqAcc *acc) {
    const int lane = lane_id();
    const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
    unsigned long long ca = 0, cc = 0, cg = 0, ct = 0, cn = 0;
    int qmin = 104, qmax = 33;                             // fastq.c:667-668
    for (int64_t k = wave; k < n_lines4; k += nwaves) {
        if (4 * k + 1 < n_nl) {                            // line_num % 4 == 2
            const int64_t s = nl[4 * k] + 1 - gbase, e = nl[4 * k + 1] - gbase;
            for (int64_t p = s + lane; p < e; p += 64) {
                const uint8_t c = data[p];
                ca += (c == 'A'); cc += (c == 'C'); cg += (c == 'G'); ct += (c == 'T');
                cn += !(c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 13);
            }
        }
        if (4 * k + 3 < n_nl) {                            // line_num % 4 == 0
            const int64_t s = nl[4 * k + 2] + 1 - gbase, e = nl[4 * k + 3] - gbase;
            for (int64_t p = s + lane; p < e; p += 64) {
                const int c = (int)(signed char)data[p];
                if (c != 13) { qmin = c < qmin ? c : qmin; qmax = c > qmax ? c : qmax; }
            }
        }
    }
    ca = wave_sum64(ca); cc = wave_sum64(cc); cg = wave_sum64(cg); ct = wave_sum64(ct); cn = wave_sum64(cn);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        int a = __shfl_xor(qmin, d, 64), b = __shfl_xor(qmax, d, 64);
        qmin = a < qmin ? a : qmin; qmax = b > qmax ? b : qmax;
    }
    if (lane == 0) {
        if (ca) atomicAdd(&acc->a, ca); if (cc) atomicAdd(&acc->c, cc); if (cg) atomicAdd(&acc->g, cg);
        if (ct) atomicAdd(&acc->t, ct); if (cn) atomicAdd(&acc->n, cn);
        atomicMin(&acc->minqs, qmin); atomicMax(&acc->maxqs, qmax);
    }
}

// FASTQ read fetch (read.c:37-45,152-167,237-278): one wave per read copies
// rlen bytes at soff and at qoff; quali = qual - phred as int8.
__global__ __launch_bounds__(BLOCK) void k_fastq_fetch(const uint8_t *__restrict__ data, int64_t gbase,
                                                      const int64_t *__restrict__ rlen, const int64_t *__restrict__ soff,
                                                      const int64_t *__restrict__ qoff, int64_t n_reads,
                                                      const int64_t *__restrict__ ids, int64_t nq, int phred, int flags,
                                                      uint8_t *__restrict__ seq, uint8_t *__restrict__ qual,
                                                      int8_t *__restrict__ quali, const int64_t *__restrict__ dst_off) {
    __shared__ uint8_t lut[256];
    build_comp_lut(lut);
    __syncthreads();
    const int lane = lane_id();
    const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
    for (int64_t i = wave; i < nq; i += nwaves) {
        const int64_t id = ids ? ids[i] : i;           // ids == null: the arrays are per query already
        if (id < 0 || id >= n_reads) continue;
        const int64_t n = rlen[id], so = soff[id] - gbase, qo = qoff[id] - gbase, d = dst_off[i];
        for (int64_t j = lane; j < n; j += 64) {
            if (seq) {
                uint8_t c = data[so + j];
                if (flags & 4) c = lut[c];
                seq[d + ((flags & 2) ? (n - 1 - j) : j)] = c;
            }
            const uint8_t qc = data[qo + j];
            if (qual) qual[d + j] = qc;
            if (quali) quali[d + j] = (int8_t)((int)(signed char)qc - phred);
        }
    }
}

// ============================================================ FASTA composition
// fasta.c:901-950: per-record histogram of the bytes on sequence lines ('\n'
// excluded, '\r' included, header lines excluded, bytes before the first header
// dropped).  One workgroup per tile.  Fast path (tile lies inside one record's
// sequence block): SWAR compare+popcount for the ten bytes that make up
// essentially all of a genome (ACGTN acgtn); any other byte value falls to an
// LDS histogram.  Slow path (tile touches a header line or a record boundary):
// per byte record lookup.  Results are flushed with 64-bit global atomics, a
// handful per tile.
__device__ __forceinline__ uint32_t cnt_eq16(const uint4 &v, uint32_t pat) {
    return __popc(zero_bytes(v.x ^ pat)) + __popc(zero_bytes(v.y ^ pat)) + __popc(zero_bytes(v.z ^ pat)) +
           __popc(zero_bytes(v.w ^ pat));
}

__global__ __launch_bounds__(BLOCK) void k_fasta_comp(const uint8_t *__restrict__ data, int64_t n, int64_t gbase,
                                                     const int64_t *__restrict__ hdr, const int64_t *__restrict__ boff,
                                                     int64_t n_hdr, const uint32_t *__restrict__ tile_hdr,
                                                     unsigned long long *__restrict__ comp) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t lds4[4];
    const int tid = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t tbase = tile * (int64_t)TILE;
    const int64_t tend = (tbase + TILE < n) ? tbase + TILE : n;
    hist[tid] = 0;
    __syncthreads();
    // record that owns the first byte of the tile
    const int64_t rec0 = upper_bound(hdr, n_hdr, gbase + tbase) - 1;
    const bool fast = rec0 >= 0 && tile_hdr[tile] == 0 && (gbase + tbase) >= boff[rec0];
    if (fast) {
        const uint32_t pats[10] = {0x41414141u, 0x43434343u, 0x47474747u, 0x54545454u, 0x4E4E4E4Eu,
                                   0x61616161u, 0x63636363u, 0x67676767u, 0x74747474u, 0x6E6E6E6Eu};
        uint32_t c10[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t other = 0;
        for (int j = 0; j < UNROLL; ++j) {
            const int64_t p = tbase + (int64_t)(j * BLOCK + tid) * CHUNK;
            if (p >= tend) break;
            const uint4 v = load16(data, p, n);
            const int valid = (int)((tend - p < CHUNK) ? (tend - p) : CHUNK);
            uint32_t known = cnt_eq16(v, 0x0A0A0A0Au);
#pragma unroll
            for (int a = 0; a < 10; ++a) { const uint32_t c = cnt_eq16(v, pats[a]); c10[a] += c; known += c; }
            if (known != (uint32_t)valid) {                // some other byte value: exact per-byte pass
                other = 1;
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                for (int k = 0; k < valid; ++k) {
                    const uint32_t b = (w[k >> 2] >> ((k & 3) * 8)) & 0xFF;
                    const uint32_t u = b & 0xDF;
                    const bool common = (b == 10) || u == 'A' || u == 'C' || u == 'G' || u == 'T' || u == 'N';
                    if (!common && b < 128) atomicAdd(&hist[b], 1u);
                }
            }
        }
        const uint8_t sym[10] = {'A', 'C', 'G', 'T', 'N', 'a', 'c', 'g', 't', 'n'};
#pragma unroll
        for (int a = 0; a < 10; ++a) {
            const uint32_t s = block_sum(c10[a], lds4);
            if (tid == 0 && s) atomicAdd(&comp[rec0 * 128 + sym[a]], (unsigned long long)s);
        }
        const uint32_t any_other = block_sum(other, lds4);
        if (any_other && tid < 128 && hist[tid]) atomicAdd(&comp[rec0 * 128 + tid], (unsigned long long)hist[tid]);
        return;
    }
    // slow path: byte by byte with record / header-line awareness
    for (int64_t p = tbase + tid; p < tend; p += BLOCK) {
        const uint8_t b = data[p];
        if (b == '\n' || b >= 128) continue;
        const int64_t rec = upper_bound(hdr, n_hdr, gbase + p) - 1;
        if (rec < 0) continue;
        if (gbase + p < boff[rec]) continue;               // inside the header line
        atomicAdd(&comp[rec * 128 + b], 1ull);
    }
}

}  // namespace fx
