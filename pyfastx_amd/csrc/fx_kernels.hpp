// fx_kernels.hpp -- hand-written HIP kernels for gfx950 (MI355X, wave64).
//
// The reference (lmdu/pyfastx) walks the file line by line on one CPU thread
// (kseq.c:59-109 feeding index.c:230-339 / fastq.c:89-149).  None of that
// structure survives here: the stream is resident in HBM and is processed as
//
//   FASTA index   fx_spanscan.hpp: one read of the stream, per-4-KiB summaries, no line table
//   FASTQ index   K1 k_scan       bytes -> 1 bit/byte newline mask + per-tile counts   (reads the file once)
//                 K2 k_group_*    exclusive prefix over the per-tile counts            (tiny)
//                 K3 k_linetable  newline mask -> int64 line table nl[]               (n/8 bytes in, 8 B/line out)
//                 k_fastq_rec / k_fastq_comp / k_fastq_fetch
//   fetch         K7 k_fetch      G lanes per query: gather, despace, upper, revcomp   (index.c:683-707, util.c:157-269)
//   composition   k_fasta_comp
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
constexpr int GROUP = BLOCK;                    // tiles per prefix group (one count per thread of a consumer block)

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
// Launch shape (tuned with tools/scanbench.hip on MI355X, 3 GB input): 1024-thread
// workgroups, 4 loads in flight per lane, non-temporal loads (the stream is read
// exactly once; nt keeps it from displacing L2/MALL lines): 5.6 TB/s including the
// mask store, vs 4.7 TB/s for 256 threads x 8 loads with default-policy loads.  A
// workgroup covers SCAN_TILES (2) consecutive 32 KiB tiles and emits one count per tile.
constexpr int SCAN_BLOCK = 1024;
constexpr int SCAN_UNROLL = 4;
constexpr int SCAN_SPAN = SCAN_BLOCK * CHUNK * SCAN_UNROLL;     // 64 KiB per workgroup
constexpr int SCAN_TILES = SCAN_SPAN / TILE;                    // 2
constexpr int ROWS_PER_TILE = SCAN_UNROLL / SCAN_TILES;         // 2 rows of 16 KiB per tile

__device__ __forceinline__ uint4 load16_nt(const uint8_t *__restrict__ data, int64_t p, int64_t n) {
    if (p + CHUNK <= n) {
        const uint4 *q = reinterpret_cast<const uint4 *>(data + p);
        uint4 v;
        v.x = __builtin_nontemporal_load(&q->x); v.y = __builtin_nontemporal_load(&q->y);
        v.z = __builtin_nontemporal_load(&q->z); v.w = __builtin_nontemporal_load(&q->w);
        return v;
    }
    return load16(data, p, n);
}

template <bool HDR>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan(const uint8_t *__restrict__ data, int64_t n, int prev_byte,
                                                    uint16_t *__restrict__ nlmask, uint32_t *__restrict__ tile_nl,
                                                    uint32_t *__restrict__ tile_hdr, int64_t ntiles) {
    __shared__ uint32_t red[2][SCAN_TILES][SCAN_BLOCK / 64];
    const int64_t span = blockIdx.x;
    const int64_t sbase = span * (int64_t)SCAN_SPAN;
    const int tid = threadIdx.x;
    uint4 v[SCAN_UNROLL];
#pragma unroll
    for (int j = 0; j < SCAN_UNROLL; ++j) v[j] = load16_nt(data, sbase + (int64_t)(j * SCAN_BLOCK + tid) * CHUNK, n);
    uint32_t cnt[SCAN_TILES], hcnt[SCAN_TILES];
#pragma unroll
    for (int t = 0; t < SCAN_TILES; ++t) { cnt[t] = 0; hcnt[t] = 0; }
#pragma unroll
    for (int j = 0; j < SCAN_UNROLL; ++j) {
        const uint32_t m = eq_mask16(v[j], 0x0A0A0A0Au);
        nlmask[span * (SCAN_SPAN / CHUNK) + j * SCAN_BLOCK + tid] = (uint16_t)m;
        cnt[j / ROWS_PER_TILE] += __popc(m);
        if (HDR) {
            if (any_eq16(v[j], 0x3E3E3E3Eu)) {
                uint32_t g = eq_mask16(v[j], 0x3E3E3E3Eu);
                const int64_t p = sbase + (int64_t)(j * SCAN_BLOCK + tid) * CHUNK;
                while (g) {
                    const int k = __ffs(g) - 1;
                    g &= g - 1;
                    const int64_t pos = p + k;
                    const int prev = pos ? (int)data[pos - 1] : prev_byte;
                    hcnt[j / ROWS_PER_TILE] += (prev == '\n');
                }
            }
        }
    }
    const int w = tid >> 6, l = tid & 63;
#pragma unroll
    for (int t = 0; t < SCAN_TILES; ++t) {
        const uint32_t a = wave_sum(cnt[t]);
        if (l == 0) red[0][t][w] = a;
        if (HDR) { const uint32_t b = wave_sum(hcnt[t]); if (l == 0) red[1][t][w] = b; }
    }
    __syncthreads();
    if (tid < SCAN_TILES * (HDR ? 2 : 1)) {
        const int t = tid % SCAN_TILES, which = tid / SCAN_TILES;
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < SCAN_BLOCK / 64; ++i) s += red[which][t][i];
        const int64_t tile = span * SCAN_TILES + t;
        if (tile < ntiles) (which ? tile_hdr : tile_nl)[tile] = s;
    }
}

// ======================================================================= K2
// The global prefix of the per-tile counts is two-level: k_group_sum adds up each
// GROUP (256) of tile counts (one wave per group), k_group_scan scans those sums
// (ntiles/256 words: 364 for a 3 GB file), and every consumer workgroup adds the
// counts of the tiles before it inside its own group (tile_prefix below).
__global__ __launch_bounds__(BLOCK) void k_group_sum(const uint32_t *__restrict__ tile_a, const uint32_t *__restrict__ tile_b,
                                                    int64_t ntiles, int64_t ngroups,
                                                    unsigned long long *__restrict__ grp_cnt) {
    const int lane = lane_id();
    const int64_t g = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    if (g >= ngroups) return;
    uint32_t a = 0, b = 0;
#pragma unroll
    for (int k = 0; k < GROUP / 64; ++k) {
        const int64_t t = g * GROUP + k * 64 + lane;
        if (t < ntiles) { a += tile_a[t]; if (tile_b) b += tile_b[t]; }
    }
    a = wave_sum(a);
    if (tile_b) b = wave_sum(b);
    if (lane == 0) { grp_cnt[g] = a; if (tile_b) grp_cnt[ngroups + g] = b; }
}

// Exclusive prefix over the per-GROUP counts (ntiles/256 words: 364 for a 3 GB
// file), one workgroup, `nsets` independent arrays back to back.
// off[s*(ngroups+1) + g] = sum(cnt[s*ngroups + 0..g)), last entry = total.
__global__ __launch_bounds__(1024) void k_group_scan(const unsigned long long *__restrict__ cnt, int64_t ngroups,
                                                    int nsets, int64_t *__restrict__ off) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int s = 0; s < nsets; ++s) {
        if (tid == 0) carry = 0;
        __syncthreads();
        for (int64_t g0 = 0; g0 < ngroups; g0 += 1024) {
            const int64_t g = g0 + tid;
            const unsigned long long v = g < ngroups ? cnt[s * ngroups + g] : 0ull;
            unsigned long long inc = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { unsigned long long t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
            if (lane == 63) wsum[w] = inc;
            __syncthreads();
            unsigned long long base = carry;
            for (int i = 0; i < w; ++i) base += wsum[i];
            if (g < ngroups) off[s * (ngroups + 1) + g] = (int64_t)(base + inc - v);
            __syncthreads();
            if (tid == 1023) carry = base + inc;
            __syncthreads();
        }
        if (tid == 0) off[s * (ngroups + 1) + ngroups] = (int64_t)carry;
        __syncthreads();
    }
}

// prefix of a tile inside its group: sum of the counts of the tiles before it (<= 255 loads, one per thread)
__device__ __forceinline__ int64_t tile_prefix(const uint32_t *__restrict__ tile_cnt, const int64_t *__restrict__ grp_off,
                                               int64_t tile, uint32_t *lds4) {
    const int64_t g = tile / GROUP, idx = g * GROUP + threadIdx.x;
    const uint32_t v = idx < tile ? tile_cnt[idx] : 0u;
    return grp_off[g] + (int64_t)block_sum(v, lds4);
}

// ======================================================================= K3
// Newline mask -> line table.  One workgroup per tile; each lane takes 8
// consecutive 16-bit masks (one 16-byte load = 128 bytes of file), the block
// does an exclusive scan of the popcounts, and every lane writes the GLOBAL
// offsets of its newlines at nl[tile_off + rank].
// nl[i] is the offset of the '\n' that terminates line i; it carries
// `position += line.l + 1` (index.c:231, fastq.c:148) for every line at once.
__global__ __launch_bounds__(BLOCK) void k_linetable(const uint16_t *__restrict__ nlmask,
                                                    const uint32_t *__restrict__ tile_nl,
                                                    const int64_t *__restrict__ grp_off, int64_t gbase,
                                                    int64_t *__restrict__ nl, int64_t cap) {
    __shared__ uint32_t lds4[4];
    const int64_t tile = blockIdx.x;
    const int tid = threadIdx.x;
    if (tile_nl[tile] == 0) return;
    const int64_t tbase_rank = tile_prefix(tile_nl, grp_off, tile, lds4);
    const uint4 mv = *reinterpret_cast<const uint4 *>(nlmask + tile * TILE_CHUNKS + tid * 8);
    const uint32_t w[4] = {mv.x, mv.y, mv.z, mv.w};
    const uint32_t cnt = __popc(w[0]) + __popc(w[1]) + __popc(w[2]) + __popc(w[3]);
    uint32_t total;
    uint32_t r = block_excl_scan(cnt, lds4, &total);
    if (cnt == 0) return;
    // `cap`: the table is allocated from an estimate before the host knows the total (it learns it
    // while this kernel runs); ranks beyond it are dropped and the host re-runs with the exact size
    int64_t rk = tbase_rank + r;
    const int64_t p0 = gbase + tile * (int64_t)TILE + (int64_t)tid * 8 * CHUNK;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t m = w[q];                        // two 16-bit masks = 32 bytes of file, bit i = byte i
        while (m) {
            const int k = __ffs(m) - 1;
            m &= m - 1;
            if (rk < cap) nl[rk] = p0 + q * 32 + k;
            ++rk;
        }
    }
}

__global__ void k_set_i64(int64_t *p, int64_t v) { *p = v; }

// FASTA record table columns (SoA, one entry per header line).  Column semantics follow
// index.c:234-339 exactly, including the quirks (elen taken from the header line only,
// index.c:266-269; blen/boff in "position" units that over-count by one when the stream lacks a
// final '\n', index.c:231 -- the virtual end-of-stream newline reproduces that).  The kernels
// that fill it live in fx_spanscan.hpp.
struct FastaCols {
    int64_t *hoff, *boff, *blen, *slen, *llen, *hdr_line;
    int32_t *elen, *dlen, *name_len, *norm;
    uint32_t *bad;
};

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

// ---- helpers of the line-arithmetic fetch path (k_fetch, BY_ID, norm = 1 records) ------------------
typedef uint4 __attribute__((aligned(1))) uint4_u;       // 16 bytes at any byte address (gfx9 unaligned access mode)
typedef uint2 __attribute__((aligned(1))) uint2_u;
typedef uint32_t __attribute__((aligned(1))) uint32_u;
typedef uint16_t __attribute__((aligned(1))) uint16_u;

// 32-bit word i of the 128-bit mask whose low T bytes (0 <= T <= 16) are 0xFF
__device__ __forceinline__ uint32_t lowbytes_word(int T, int i) {
    const int tt = T - 4 * i;
    return tt >= 4 ? 0xFFFFFFFFu : (tt <= 0 ? 0u : ((1u << (8 * tt)) - 1u));
}
// 0x80 in every byte of w that is < 0x80 and <= 0x20 (white space and control bytes)
__device__ __forceinline__ uint32_t le20_bytes(uint32_t w) { return ~(((w & 0x7F7F7F7Fu) + 0x5F5F5F5Fu) | w) & 0x80808080u; }
// Py_TOUPPER on four bytes (util.c:181-194): 'a'..'z' -> 'A'..'Z', everything else unchanged
__device__ __forceinline__ uint32_t upper4(uint32_t w) {
    const uint32_t x = w & 0x7F7F7F7Fu;
    const uint32_t lower = (x + 0x1F1F1F1Fu) & ~(x + 0x05050505u) & ~w & 0x80808080u;     // 0x61 <= c <= 0x7A
    return w - (lower >> 2);
}
__device__ __forceinline__ uint32_t lut4(const uint8_t *lut, uint32_t w) {
    return (uint32_t)lut[w & 0xFF] | ((uint32_t)lut[(w >> 8) & 0xFF] << 8) | ((uint32_t)lut[(w >> 16) & 0xFF] << 16) |
           ((uint32_t)lut[w >> 24] << 24);
}
__device__ __forceinline__ bool is_space3(uint32_t c) { return c == 10u || c == 13u || c == 32u; }   // jump_table, util.c:157-164
// the low `len` (1..16) bytes of v to p, any alignment
__device__ __forceinline__ void store_low_bytes(uint8_t *p, uint4 v, int len) {
    if (len >= 16) { *reinterpret_cast<uint4_u *>(p) = v; return; }
    if (len & 8) { *reinterpret_cast<uint2_u *>(p) = make_uint2(v.x, v.y); p += 8; v.x = v.z; v.y = v.w; }
    if (len & 4) { *reinterpret_cast<uint32_u *>(p) = v.x; p += 4; v.x = v.y; }
    if (len & 2) { *reinterpret_cast<uint16_u *>(p) = (uint16_t)v.x; p += 2; v.x >>= 16; }
    if (len & 1) *p = (uint8_t)v.x;
}

// G lanes cooperate on one query (64/G queries in flight per wave); each lane
// loads V aligned bytes per step, so a step covers a G*V-byte window:
//   < 8,16>  128-byte window, 8 queries per wave  -- ~100-bp random access
//   <64,16>  1 KiB window, 1 query per wave       -- long ranges (whole records)
// General path: the keep mask of a lane's V bytes is SWAR, the rank of its first kept byte is
// an exclusive prefix over the G lanes (__shfl_up, width G), kept bytes whose
// rank falls in [skip, skip+take) are stored (mirrored for FX_REVERSE).
// Line-arithmetic path (BY_ID, norm = 1, >= 16 bases per line): base i of a record sits at byte
// boff + i + elen * (i / bases_per_line) (sequence.c:498-510), so a lane produces 16 OUTPUT bytes
// from one unaligned 16-byte load -- two when a line end falls inside, merged with a byte mask --
// and writes them with one unaligned 16-byte store: no keep mask, no ranks, no per-byte loop.  It
// checks that the skipped bytes are line terminators and that no white space hides inside the
// lines; if either fails the query is redone by the general path, so results always equal
// "read the byte range, drop 10/13/32" (index.c:694-707, util.c:157-194).
template <bool BY_ID, int G, int V>
__global__ __launch_bounds__(BLOCK) void k_fetch(const uint8_t *__restrict__ data, int64_t gbase, int64_t n_bytes,
                                                FetchQ q, FastaTab tab, int64_t nq, int flags_all,
                                                uint8_t *__restrict__ dst) {
    __shared__ uint8_t lut[256];
    // BY_ID: the record table of a genome (a few hundred rows) is copied to LDS once per workgroup, so
    // resolving a query costs one LDS read instead of a second dependent trip to memory
    constexpr int TABCAP = BY_ID ? 512 : 1;
    __shared__ int64_t s_boff[TABCAP], s_slen[TABCAP], s_blen[TABCAP];
    __shared__ int32_t s_llen[TABCAP], s_en[TABCAP];           // s_en = elen | norm << 8
    const bool tab_lds = BY_ID && tab.n_seq <= TABCAP && tab.n_seq > 0;
    build_comp_lut(lut);
    if (tab_lds)
        for (int r = threadIdx.x; r < (int)tab.n_seq; r += BLOCK) {
            s_boff[r] = tab.boff[r]; s_slen[r] = tab.slen[r]; s_blen[r] = tab.blen[r];
            const int64_t ll = tab.llen[r];
            s_llen[r] = ll > 0x7FFFFFFFll ? 0x7FFFFFFF : (int32_t)ll;
            s_en[r] = tab.elen[r] | (tab.norm[r] << 8);
        }
    __syncthreads();
    constexpr int QPW = 64 / G, NW = V / 4;
    const int lane = lane_id(), sub = lane & (G - 1), grp = lane / G;
    const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
    // query descriptors are fetched one iteration ahead (the loop is a chain of dependent memory round
    // trips: descriptor -> table -> sequence bytes; this takes the first one off the critical path)
    int64_t d_a0 = 0, d_a1 = 0, d_a2 = 0, d_a3 = 0, d_off = 0;
    int d_fl = flags_all;
    auto fetch_desc = [&](int64_t i) {
        if (i >= nq) return;
        if (BY_ID) { d_a0 = q.seq_id[i]; d_a1 = q.start[i]; d_a2 = q.stop[i]; }
        else       { d_a0 = q.off[i]; d_a1 = q.blen[i]; d_a2 = q.take[i]; d_a3 = q.skip ? q.skip[i] : 0; }
        d_off = q.dst_off[i];
        d_fl = q.qflags ? q.qflags[i] : flags_all;
    };
    fetch_desc(wave * QPW + grp);
    for (int64_t i0 = wave * QPW; i0 < nq; i0 += nwaves * QPW) {
        const int64_t i = i0 + grp;
        bool ok = i < nq;
        const int64_t c_a0 = d_a0, c_a1 = d_a1, c_a2 = d_a2, c_a3 = d_a3, c_off = d_off;
        const int fl = d_fl;
        fetch_desc(i + nwaves * QPW);
        int64_t off = 0, blen = 0, skip = 0, take = 0;
        int64_t r_boff = 0, r_el = 0, r_bpl = 0;
        bool r_norm = false;
        if (ok) {
            if (BY_ID) {
                const int64_t id = c_a0, a = c_a1, b = c_a2;
                int64_t r_slen = 0, r_blen = 0;
                if (id >= 0 && id < tab.n_seq) {
                    if (tab_lds) { r_boff = s_boff[id]; r_slen = s_slen[id]; r_blen = s_blen[id]; r_el = s_en[id] & 0xFF; r_norm = (s_en[id] >> 8) != 0;
                                   r_bpl = (int64_t)s_llen[id] - r_el; if (s_llen[id] == 0x7FFFFFFF) r_bpl = tab.llen[id] - r_el; }
                    else         { r_boff = tab.boff[id]; r_slen = tab.slen[id]; r_blen = tab.blen[id]; r_el = tab.elen[id]; r_norm = tab.norm[id] != 0;
                                   r_bpl = tab.llen[id] - r_el; }
                }
                if (id < 0 || id >= tab.n_seq || a < 0 || b < a || b > r_slen) {   // caller validates; stay safe
                    if (sub == 0 && q.out_len) q.out_len[i] = -1;
                    ok = false;
                } else {
                    take = b - a;
                    if (r_norm && r_bpl > 0) {                 // sequence.c:498-510
                        int64_t bs, be;
                        if (((uint64_t)b | (uint64_t)r_bpl) >> 32) { bs = a / r_bpl; be = b / r_bpl; }
                        else { bs = (uint32_t)a / (uint32_t)r_bpl; be = (uint32_t)b / (uint32_t)r_bpl; }
                        off = r_boff + a + r_el * bs;
                        blen = take + (be - bs) * r_el;
                    } else {                                   // sequence.c:100-110: despace whole record, then slice
                        off = r_boff; blen = r_blen; skip = a;
                    }
                }
            } else {
                off = c_a0; blen = c_a1; take = c_a2; skip = c_a3;
            }
        }
        uint8_t *out = dst + (ok ? c_off : 0);
        if (BY_ID) {
            // ---- line-arithmetic path
            bool fast = false;
            int64_t a = 0, el = 0, bpl = 0, in_a = 0;
            if (ok && take > 0 && !(fl & 8)) {
                a = c_a1; el = r_el; bpl = r_bpl;
                in_a = off - gbase;                               // local offset of base `a` (off was computed above)
                fast = r_norm && bpl >= 16 && bpl < (1ll << 31) && a + take < (1ll << 31) &&
                       in_a >= 16 && in_a + blen + 32 <= n_bytes;
            }
            bool redo = false;
            if (fast) {
                const uint32_t bpl32 = (uint32_t)bpl;
                const uint32_t r0 = (uint32_t)a % bpl32;          // column of base `a` in its line
                const bool rev = (fl & 2) != 0;
                bool irregular = false;
                for (int64_t s0 = 0; s0 < take; s0 += G * 16) {
                    const int64_t oc = s0 + 16 * sub;             // this lane writes out[oc, oc + len)
                    const int len = (int)(take - oc < 16 ? take - oc : 16);
                    if (len <= 0) continue;
                    // forward indices [f0, f0 + len) of the query; `lead` unused bytes in front of them in the
                    // lane's 16-byte window (reverse strand: the partial chunk is the head of the query)
                    const int64_t f0 = rev ? take - oc - len : oc;
                    const int lead = rev ? 16 - len : 0;
                    const uint32_t xx = r0 + (uint32_t)f0;
                    const uint32_t k = xx / bpl32, t = bpl32 - (xx - k * bpl32);   // line of f0 (relative), bases to its end
                    const uint8_t *p1 = data + in_a + f0 + el * (int64_t)k - lead;
                    uint4 v = *reinterpret_cast<const uint4_u *>(p1);
                    if (t < (uint32_t)len) {                      // a line ends inside: bytes from index lead + t on come from el further
                        const uint4 w = *reinterpret_cast<const uint4_u *>(p1 + el);
                        const int T = lead + (int)t;
                        const uint32_t m0 = lowbytes_word(T, 0), m1 = lowbytes_word(T, 1), m2 = lowbytes_word(T, 2), m3 = lowbytes_word(T, 3);
                        v.x = (v.x & m0) | (w.x & ~m0); v.y = (v.y & m1) | (w.y & ~m1);
                        v.z = (v.z & m2) | (w.z & ~m2); v.w = (v.w & m3) | (w.w & ~m3);
                    }
                    if (t < (uint32_t)len || (t == (uint32_t)len && f0 + len < take)) {   // the terminator after base f0 + t - 1 is skipped
                        const uint8_t *gp = p1 + lead + t;
                        irregular |= !is_space3(gp[0]) || (el == 2 && !is_space3(gp[1]));
                    }
                    {   // no white space among the bases taken
                        const int lo_b = lead, hi_b = lead + len;
                        uint32_t f = le20_bytes(v.x) & lowbytes_word(hi_b, 0) & ~lowbytes_word(lo_b, 0);
                        f |= le20_bytes(v.y) & lowbytes_word(hi_b, 1) & ~lowbytes_word(lo_b, 1);
                        f |= le20_bytes(v.z) & lowbytes_word(hi_b, 2) & ~lowbytes_word(lo_b, 2);
                        f |= le20_bytes(v.w) & lowbytes_word(hi_b, 3) & ~lowbytes_word(lo_b, 3);
                        irregular |= f != 0;
                    }
                    if (fl & 1) { v.x = upper4(v.x); v.y = upper4(v.y); v.z = upper4(v.z); v.w = upper4(v.w); }
                    if (fl & 4) { v.x = lut4(lut, v.x); v.y = lut4(lut, v.y); v.z = lut4(lut, v.z); v.w = lut4(lut, v.w); }
                    if (rev) v = make_uint4(__builtin_bswap32(v.w), __builtin_bswap32(v.z), __builtin_bswap32(v.y), __builtin_bswap32(v.x));
                    store_low_bytes(out + oc, v, len);
                }
                const unsigned long long ib = __ballot(irregular);
                redo = ((ib >> (grp * G)) & (G == 64 ? ~0ull : ((1ull << G) - 1ull))) != 0;
                if (!redo && sub == 0 && q.out_len) q.out_len[i] = take;
            }
            if (fast && !redo) ok = false;                         // done: nothing left for the general path
            if (!ok) { take = 0; blen = 0; }
        }
        // ---- general path
        // clamp to the bytes we hold (fread past EOF returns short, index.c:689)
        int64_t lo = off - gbase, hi = lo + blen;
        if (lo < 0) lo = 0;
        if (hi > n_bytes) hi = n_bytes;
        if (!ok) { lo = 0; hi = 0; }
        const int64_t end = skip + take;
        int64_t rank = 0;                                      // kept bytes before this window
        for (int64_t p = lo & ~(int64_t)(V - 1); p < hi && rank < end; p += G * V) {
            const int64_t pp = p + (int64_t)sub * V;
            uint32_t w[NW];
#pragma unroll
            for (int k = 0; k < NW; ++k) w[k] = 0;
            if (pp < hi) {
                if (V == 16) { const uint4 t = *reinterpret_cast<const uint4 *>(data + pp); w[0] = t.x; w[1] = t.y; w[NW - 2] = t.z; w[NW - 1] = t.w; }
                else         { const uint2 t = *reinterpret_cast<const uint2 *>(data + pp); w[0] = t.x; w[NW - 1] = t.y; }
            }
            // bits of the lane's V bytes that are inside [lo,hi) and not white space
            int64_t a0 = lo - pp, a1 = hi - pp;
            a0 = a0 < 0 ? 0 : (a0 > V ? V : a0);
            a1 = a1 < 0 ? 0 : (a1 > V ? V : a1);
            uint32_t km = (a1 > a0) ? (((1u << a1) - 1u) & ~((1u << a0) - 1u)) & ((V == 16) ? 0xFFFFu : 0xFFu) : 0u;
            if (!(fl & 8)) {                                   // jump_table (util.c:157-164): drop 10, 13, 32
                uint32_t sp = 0;
#pragma unroll
                for (int k = 0; k < NW; ++k)
                    sp |= flags4(zero_bytes(w[k] ^ 0x0A0A0A0Au) | zero_bytes(w[k] ^ 0x0D0D0D0Du) |
                                 zero_bytes(w[k] ^ 0x20202020u)) << (4 * k);
                km &= ~sp;
            }
            const int kcnt = __popc(km);
            int inc = kcnt;
#pragma unroll
            for (int d = 1; d < G; d <<= 1) { const int t = __shfl_up(inc, d, G); if (sub >= d) inc += t; }
            const int total = __shfl(inc, G - 1, G);
            int64_t r = rank + inc - kcnt;
            while (km) {
                const int j = __ffs(km) - 1;
                km &= km - 1;
                if (r >= skip && r < end) {
                    uint32_t word = w[0];
#pragma unroll
                    for (int k = 1; k < NW; ++k) word = (j >> 2) == k ? w[k] : word;
                    uint8_t c = (uint8_t)(word >> ((j & 3) * 8));
                    if ((fl & 1) && c >= 'a' && c <= 'z') c -= 32;
                    if (fl & 4) c = lut[c];
                    const int64_t o = r - skip;
                    out[(fl & 2) ? (take - 1 - o) : o] = c;
                }
                ++r;
            }
            rank += total;
        }
        if (ok && sub == 0 && q.out_len) {
            const int64_t got = rank - skip;
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
// The `line_num % 4` state machine of fastq.c:89-149 becomes a gather from the
// line table: record k owns lines 4k .. 4k+3.  Shard context (FqCtx) makes the
// same kernels serve byte-range shards: nl[] is local (index i = global line
// loff + i), a record is owned by the shard in whose core its header line STARTS,
// and the bytes of its remaining lines may lie in the halo that follows the core.
struct FastqCols {
    int64_t *name_off, *rlen, *soff, *qoff;
    int32_t *name_len, *dlen;
};
struct FastqAcc {            // device accumulators
    unsigned long long size;
    unsigned long long a, c, g, t, n;
    unsigned long long n_owned;
    long long maxlen, minlen;
    int minqs, maxqs;
    int err;                 // 1: a record ran past the halo
    int pad;
};
struct FqCtx {
    int64_t gbase;           // global offset of data[0]
    int64_t core_end;        // global offset one past the shard's core
    int64_t loff;            // global line index of nl[0]
    int64_t prev_nl;         // global offset of newline loff-1 (-1: none)
    int64_t k_first;         // first record owned by this shard (global id of local row 0)
    int is_last;
};

// global offset where line j (0..3) of record k starts / the newline that ends it
__device__ __forceinline__ int64_t fq_line_end(const FqCtx &x, const int64_t *nl, int64_t k, int j) { return nl[4 * k + j - x.loff]; }
__device__ __forceinline__ int64_t fq_line_start(const FqCtx &x, const int64_t *nl, int64_t k, int j) {
    const int64_t i = 4 * k + j - 1 - x.loff;
    return (i >= 0 ? nl[i] : x.prev_nl) + 1;
}

// One thread per candidate record: columns of the `read` table (fastq.c:99-145), stat.size
// (sum of rlen, including an incomplete trailing record's sequence line, fastq.c:125) and the
// min/max quality-line length for meta.maxlen/minlen (fastq.c:747-751).
__global__ __launch_bounds__(BLOCK) void k_fastq_rec(const uint8_t *__restrict__ data, FqCtx x,
                                                    const int64_t *__restrict__ nl, int64_t n_nl, int64_t n_cand,
                                                    FastqCols c, FastqAcc *acc) {
    __shared__ long long red[4][BLOCK / 64];
    int64_t rl_sum = 0;
    long long qmax = 0, qmin = 10000000000LL;
    unsigned owned = 0;
    int err = 0;
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < n_cand; t += stride) {
        const int64_t k = x.k_first + t;
        const int64_t i0 = 4 * k - x.loff;                // local index of the newline ending the header line
        if (i0 - 1 >= n_nl) continue;                     // header start unknown: no such line
        const int64_t s0 = fq_line_start(x, nl, k, 0);
        if (s0 >= x.core_end) continue;                   // owned by a later shard (or past the data)
        if (i0 + 3 < n_nl) {                              // all four lines present
            const int64_t e0 = nl[i0], soff = e0 + 1, e1 = nl[i0 + 1];
            const int64_t l1 = e1 - soff;
            const int64_t rl = (l1 > 0 && data[e1 - 1 - x.gbase] == '\r') ? l1 - 1 : l1;   // fastq.c:124-128
            const int dlen = (int)(e0 - s0);              // fastq.c:103 (includes '@' and '\r')
            int64_t nlen = dlen - 1;
            if (nlen > 0 && data[e0 - 1 - x.gbase] == '\r') --nlen;        // fastq.c:107-109
            // first ' ' of the name (fastq.c:112-117), 8 aligned bytes at a time
            const int64_t nb = s0 + 1 - x.gbase, ne = nb + nlen;
            for (int64_t p = nb & ~7ll; p < ne; p += 8) {
                const uint2 w = *reinterpret_cast<const uint2 *>(data + p);
                uint32_t m = flags4(zero_bytes(w.x ^ 0x20202020u)) | (flags4(zero_bytes(w.y ^ 0x20202020u)) << 4);
                if (p < nb) m &= 0xFFu << (nb - p);
                if (m) { const int64_t hit = p + __ffs(m) - 1; if (hit < ne) nlen = hit - nb; break; }
            }
            const int64_t qoff = nl[i0 + 2] + 1, e3 = nl[i0 + 3];
            long long ql = e3 - qoff;
            if (ql > 0 && data[e3 - 1 - x.gbase] == '\r') --ql;            // fastq.c:734-737 (trailing CR)
            qmax = ql > qmax ? ql : qmax; qmin = ql < qmin ? ql : qmin;
            c.name_off[t] = s0 + 1; c.name_len[t] = (int32_t)nlen; c.dlen[t] = dlen;
            c.rlen[t] = rl; c.soff[t] = soff; c.qoff[t] = qoff;
            rl_sum += rl;
            ++owned;
        } else if (!x.is_last) {
            err = 1;                                      // the record runs past the halo
        } else if (i0 + 1 < n_nl) {                       // incomplete trailing record: its sequence line still counts (fastq.c:125)
            const int64_t soff = nl[i0] + 1, e1 = nl[i0 + 1];
            const int64_t l1 = e1 - soff;
            rl_sum += (l1 > 0 && data[e1 - 1 - x.gbase] == '\r') ? l1 - 1 : l1;
        }
    }
    // one set of atomics per workgroup
    rl_sum = wave_sum64(rl_sum);
    long long own = wave_sum64((int64_t)owned);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        long long a = __shfl_xor(qmax, d, 64), b = __shfl_xor(qmin, d, 64);
        qmax = a > qmax ? a : qmax; qmin = b < qmin ? b : qmin;
    }
    if (__any(err)) acc->err = 1;
    const int w = threadIdx.x >> 6;
    if (lane_id() == 0) { red[0][w] = rl_sum; red[1][w] = own; red[2][w] = qmax; red[3][w] = qmin; }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long s = 0, o = 0, mx = 0, mn = 10000000000LL;
        for (int i = 0; i < BLOCK / 64; ++i) { s += red[0][i]; o += red[1][i]; mx = red[2][i] > mx ? red[2][i] : mx; mn = red[3][i] < mn ? red[3][i] : mn; }
        if (s) atomicAdd(&acc->size, (unsigned long long)s);
        if (o) atomicAdd(&acc->n_owned, (unsigned long long)o);
        if (o) { atomicMax(&acc->maxlen, mx); atomicMin(&acc->minlen, mn); }
    }
}

// FASTQ composition (fastq.c:715-753).  16 lanes per record, 4 records per wave; a lane
// reads 16 aligned bytes per step (256-byte window per record).  Sequence line: SWAR
// compare+popcount for 'A','C','G','T' (upper case only) and '\r' (ignored); every other
// byte is N.  Quality line: min / max byte, '\r' ignored.
__device__ __forceinline__ uint32_t valid16(int64_t pp, int64_t lo, int64_t hi) {
    int64_t a0 = lo - pp, a1 = hi - pp;
    a0 = a0 < 0 ? 0 : (a0 > 16 ? 16 : a0);
    a1 = a1 < 0 ? 0 : (a1 > 16 ? 16 : a1);
    return a1 > a0 ? (((1u << a1) - 1u) & ~((1u << a0) - 1u)) : 0u;
}
__global__ __launch_bounds__(BLOCK) void k_fastq_comp(const uint8_t *__restrict__ data, FqCtx x,
                                                     const int64_t *__restrict__ nl, int64_t n_nl,
                                                     int64_t n_cand, FastqAcc *acc) {
    const int lane = lane_id(), sub = lane & 15, grp = lane >> 4;
    const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
    uint32_t ca = 0, cc = 0, cg = 0, ct = 0, cn = 0;       // per lane, flushed per record batch (no overflow: <= 16 per step)
    unsigned long long ta = 0, tc = 0, tg = 0, tt = 0, tn = 0;
    int qmin = 104, qmax = 33;                             // fastq.c:667-668
    for (int64_t t0 = wave * 4; t0 < n_cand; t0 += nwaves * 4) {
        const int64_t k = x.k_first + t0 + grp;
        const int64_t i0 = 4 * k - x.loff;
        bool mine = (t0 + grp) < n_cand && i0 - 1 < n_nl && i0 < n_nl;
        if (mine) mine = fq_line_start(x, nl, k, 0) < x.core_end;
        if (mine && i0 + 1 < n_nl) {                       // line_num % 4 == 2
            const int64_t s = nl[i0] + 1 - x.gbase, e = nl[i0 + 1] - x.gbase;
            for (int64_t p = (s & ~15ll) + sub * 16; p < e; p += 256) {
                const uint4 v = *reinterpret_cast<const uint4 *>(data + p);
                const uint32_t ok = valid16(p, s, e);
                const uint32_t ma = eq_mask16(v, 0x41414141u) & ok, mc = eq_mask16(v, 0x43434343u) & ok;
                const uint32_t mg = eq_mask16(v, 0x47474747u) & ok, mt = eq_mask16(v, 0x54545454u) & ok;
                const uint32_t mr = eq_mask16(v, 0x0D0D0D0Du) & ok;
                ca += __popc(ma); cc += __popc(mc); cg += __popc(mg); ct += __popc(mt);
                cn += __popc(ok & ~(ma | mc | mg | mt | mr));
            }
        }
        if (mine && i0 + 3 < n_nl) {                       // line_num % 4 == 0
            const int64_t s = nl[i0 + 2] + 1 - x.gbase, e = nl[i0 + 3] - x.gbase;
            for (int64_t p = (s & ~15ll) + sub * 16; p < e; p += 256) {
                const uint4 v = *reinterpret_cast<const uint4 *>(data + p);
                uint32_t ok = valid16(p, s, e) & ~eq_mask16(v, 0x0D0D0D0Du);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                while (ok) {
                    const int j = __ffs(ok) - 1;
                    ok &= ok - 1;
                    const int q = (int)(signed char)(w[j >> 2] >> ((j & 3) * 8));
                    qmin = q < qmin ? q : qmin; qmax = q > qmax ? q : qmax;
                }
            }
        }
        ta += ca; tc += cc; tg += cg; tt += ct; tn += cn;
        ca = cc = cg = ct = cn = 0;
    }
    ta = wave_sum64(ta); tc = wave_sum64(tc); tg = wave_sum64(tg); tt = wave_sum64(tt); tn = wave_sum64(tn);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        int a = __shfl_xor(qmin, d, 64), b = __shfl_xor(qmax, d, 64);
        qmin = a < qmin ? a : qmin; qmax = b > qmax ? b : qmax;
    }
    if (lane == 0) {
        if (ta) atomicAdd(&acc->a, ta); if (tc) atomicAdd(&acc->c, tc); if (tg) atomicAdd(&acc->g, tg);
        if (tt) atomicAdd(&acc->t, tt); if (tn) atomicAdd(&acc->n, tn);
        atomicMin(&acc->minqs, qmin); atomicMax(&acc->maxqs, qmax);
    }
}

// number of entries of the sorted table that are < key, and the last such entry (-1 if none)
__global__ void k_count_below(const int64_t *__restrict__ a, int64_t n, int64_t key, int64_t *out) {
    const int64_t c = lower_bound(a, n, key);
    out[0] = c;
    out[1] = c ? a[c - 1] : -1;
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
                                                     int64_t n_hdr, const int64_t *__restrict__ hdr_prefix,
                                                     int64_t ngran, int gran_per_tile,
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
    // does a header line start inside this tile?  (hdr_prefix: header lines before each 4 KiB granule)
    const int64_t g0 = tile * gran_per_tile, g1 = (g0 + gran_per_tile < ngran) ? g0 + gran_per_tile : ngran;
    const bool fast = rec0 >= 0 && hdr_prefix[g1] == hdr_prefix[g0] && (gbase + tbase) >= boff[rec0];
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
