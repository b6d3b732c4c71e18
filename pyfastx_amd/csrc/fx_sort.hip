// fx_sort.hip -- order of the record names for the UNIQUE INDEX of a .fxi.
//
// The reference leaves this to SQLite: `CREATE UNIQUE INDEX readidx ON read (name)` (fastq.c:152, index.c:363) runs
// an external merge sort over every (name, rowid) on one CPU thread after the inserts -- for the 10^8 reads of a
// sequencing run that is minutes.  Here the names are already in HBM (they are slices of the resident stream), so
// the order is an LSD radix sort over them in place: the names are cut into 8-byte big-endian chunks (zero padded),
// a permutation is stable-sorted by name length first and then by chunk  ceil(maxlen/8)-1, ..., 1, 0; after the last
// pass it is ordered by (chunk 0, chunk 1, ..., length) = memcmp order with the shorter name first on a tie, which is
// SQLite's BINARY collation.  The chunks of all names are written once (k_sort_chunks), together with the OR and the AND of
// every chunk over all names: the bits in which names DIFFER.  Only those are sorted by -- they are packed, up to 64 at a
// time and across chunk borders, into the key of a round (k_sort_gather: a random 8-byte read per name and chunk touched),
// and the (key, index) pairs are sorted with stable 8-bit counting passes (k_rs_hist / k_rs_scan / k_rs_scatter below).
// The shared prefix of sequencer read names costs nothing and an ASCII digit four bits: the 31-byte names of C3 are
// 72 bits = 9 passes where whole chunks took 18 (plus 14 histograms that found a constant digit).  A final kernel compares
// neighbours to count duplicate names.  The permutation goes to the index kernels of fx_fxi_dev.hpp.
#include <hip/hip_runtime.h>
#include <cstring>
#include <vector>

#include "fx_sort.hpp"

namespace fx {

static constexpr int SB = 256;
typedef uint64_t __attribute__((aligned(1))) u64_unal;

// grid-stride; the longest name reaches max_len with ONE atomic per workgroup (one per wave on one address was 3.5 ms
// of atomics for 20 M names)
__global__ __launch_bounds__(SB) void k_sort_init(const int32_t *__restrict__ name_len, int64_t n, uint64_t *__restrict__ keys,
                                                  uint32_t *__restrict__ vals, unsigned *__restrict__ max_len) {
    __shared__ unsigned blk;
    if (threadIdx.x == 0) blk = 0;
    __syncthreads();
    int len = 0;
    for (int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x; i < n; i += (int64_t)gridDim.x * SB) {
        const int l = name_len[i] > 0 ? name_len[i] : 0;
        keys[i] = (uint64_t)l;
        vals[i] = (uint32_t)i;
        len = l > len ? l : len;
    }
    for (int o = 32; o; o >>= 1) len = max(len, __shfl_xor(len, o));
    if ((threadIdx.x & 63) == 0 && len) atomicMax(&blk, (unsigned)len);
    __syncthreads();
    if (threadIdx.x == 0 && blk) atomicMax(max_len, blk);
}

// Round 6.  The 8-byte chunks of ALL names, once, in record order: chunk c of name i = its bytes [8c, 8c+8) as a big-endian
// number, zero padded past the end -> kc[c * n + i].  The names are read where they lie in the stream, one after the
// other (a line per name whatever the number of chunks); the passes below then fetch ONE 8-byte key per name through the
// permutation.  (Round 5 went to the stream in every pass: name_off[r], name_len[r] and the bytes, three dependent random
// reads per name and chunk -- 6.9 ms per chunk for 10^8 names, 21 of the sort's 51 ms.)
// OR and AND of every chunk over all names: a bit that is equal in both is the same in every name.  A workgroup keeps its own
// pair of words per chunk in LDS (a lane goes to the LDS atomic only with a bit the words do not show yet: after the first few
// names nobody does), walks over several tiles and leaves its words in part[workgroup][chunk][2]; k_sort_orand folds them.
// (Global atomics on ONE pair of words per chunk, guarded by a load of them: 3 of the kernel's 6.5 ms for 10^8 names -- the
// L2 of an XCD keeps the line as it first saw it, so the guard never learns; read coherently the words are a hot spot: 8.7 ms.)
// Names of more than SORT_LDS_CHUNKS chunks (4 KiB) take the global atomics.
constexpr int SORT_LDS_CHUNKS = 512;
__global__ __launch_bounds__(SB) void k_sort_chunks(const uint8_t *__restrict__ data, int64_t gbase, const int64_t *__restrict__ name_off,
                                                    const int32_t *__restrict__ name_len, int64_t n, int nchunk, uint64_t *__restrict__ kc,
                                                    unsigned long long *__restrict__ part, unsigned long long *orand) {
    // one thread per (name, chunk), the chunks of a name in neighbouring lanes: one load instruction per wave reads 64 / nchunk
    // names, a name's lanes share its line (a thread per name issued nchunk loads over 64 different lines each: 8.2 ms for 10^8 names)
    __shared__ unsigned long long s_or[SORT_LDS_CHUNKS], s_and[SORT_LDS_CHUNKS];
    const bool local = nchunk <= SORT_LDS_CHUNKS;
    if (local) {
        for (int x = threadIdx.x; x < nchunk; x += SB) { s_or[x] = 0ull; s_and[x] = ~0ull; }
        __syncthreads();
    }
    const int64_t ntiles = (n * nchunk + SB - 1) / SB;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t t0 = tile * SB;                      // (the 64-bit division once per tile, on the scalar unit)
        const int64_t i0 = t0 / nchunk;
        const uint32_t x = (uint32_t)(t0 - i0 * nchunk) + threadIdx.x;
        const int64_t i = i0 + x / (uint32_t)nchunk;
        if (i >= n) continue;
        const int c = (int)(x % (uint32_t)nchunk);
        const int len = name_len[i] > 0 ? name_len[i] : 0;
        const uint8_t *p = data + (name_off[i] - gbase);
        const int rest = len - 8 * c;
        uint64_t k = 0;
        if (rest >= 8) k = __builtin_bswap64(*reinterpret_cast<const u64_unal *>(p + 8 * c));
        else if (rest > 0 && len >= 8) k = __builtin_bswap64(*reinterpret_cast<const u64_unal *>(p + len - 8)) << (8 * (8 - rest));   // the 8 bytes that end with the name
        else if (rest > 0) for (int b = 0; b < rest; ++b) k |= (uint64_t)p[b] << (56 - 8 * b);
        kc[(int64_t)c * n + i] = k;
        if (local) {
            if (k & ~s_or[c]) atomicOr(&s_or[c], (unsigned long long)k);
            if (~k & s_and[c]) atomicAnd(&s_and[c], (unsigned long long)k);
        } else {
            if (k & ~orand[2 * c]) atomicOr(&orand[2 * c], (unsigned long long)k);
            if (~k & orand[2 * c + 1]) atomicAnd(&orand[2 * c + 1], (unsigned long long)k);
        }
    }
    if (local) {
        __syncthreads();
        unsigned long long *mine = part + (int64_t)blockIdx.x * nchunk * 2;
        for (int x = threadIdx.x; x < nchunk; x += SB) { mine[2 * x] = s_or[x]; mine[2 * x + 1] = s_and[x]; }
    }
}
// one workgroup per chunk: the words of all workgroups of k_sort_chunks -> orand[2c], orand[2c + 1]
__global__ __launch_bounds__(SB) void k_sort_orand(const unsigned long long *__restrict__ part, int nblk, int nchunk, unsigned long long *__restrict__ orand) {
    __shared__ unsigned long long w_or[SB / 64], w_and[SB / 64];
    const int c = blockIdx.x;
    unsigned long long o = 0ull, a = ~0ull;
    for (int b = threadIdx.x; b < nblk; b += SB) { o |= part[((int64_t)b * nchunk + c) * 2]; a &= part[((int64_t)b * nchunk + c) * 2 + 1]; }
    for (int d = 32; d; d >>= 1) { o |= __shfl_xor(o, d); a &= __shfl_xor(a, d); }
    if ((threadIdx.x & 63) == 0) { w_or[threadIdx.x >> 6] = o; w_and[threadIdx.x >> 6] = a; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < SB / 64; ++q) { o |= w_or[q]; a &= w_and[q]; }
        orand[2 * c] = o; orand[2 * c + 1] = a;
    }
}

// The key of one round of passes: up to 64 of the bits that DIFFER between names, taken from one or more chunks and packed
// without the bits every name shares (`SYN:1:FC:1:` costs nothing, an ASCII digit four bits instead of eight).  Dropping bits
// that are equal everywhere keeps the order of any two keys, so sorting by the packed groups, least significant first, is
// sorting by the chunks.  A run = `width` consecutive bits of chunk `chunk` from bit `lo`, placed at bit `dst` of the key.
constexpr int SORT_RUNS = 64;                             // a run is at least one bit: a group is full by its bits before it is by its runs
struct SortGroup { int n, bits; uint32_t run[SORT_RUNS]; };      // run: chunk (16 bits) | lo << 16 (6) | (width - 1) << 22 (6); dst = the widths before it
constexpr int SORT_PACK_GROUPS = 32;                      // groups per launch of k_sort_pack (their descriptions lie in LDS)

// The packed keys of ALL names, once, in record order and IN PLACE: the m-th most significant group goes where chunk m was
// (its bits come from chunks >= m -- a chunk holds at most 64 differing bits --, and a thread has read them when it writes;
// launches take the groups most significant first).  groups: least significant first, as the rounds take them; this launch
// writes the slots [m0, m1).  A round then gathers one word per name (k_sort_gather), the first of them -- while the
// permutation is still the identity or close to it -- nearly in order.
__global__ __launch_bounds__(SB) void k_sort_pack(uint64_t *kc, int64_t N, int64_t n, const SortGroup *__restrict__ groups, int G, int m0, int m1) {
    __shared__ SortGroup sg[SORT_PACK_GROUPS];            // (read from global memory run by run, every run was a round trip of every wave: 2.7 ms for 10^8 names)
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(groups + (G - m1));      // slots m0..m1-1 = groups G-1-m0 .. G-m1
        uint32_t *dst = reinterpret_cast<uint32_t *>(sg);
        const int words = (m1 - m0) * (int)(sizeof(SortGroup) / 4);
        for (int x = threadIdx.x; x < words; x += SB) dst[x] = src[x];
    }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (i >= n) return;
    for (int m = m0; m < m1; ++m) {
        const SortGroup &g = sg[m1 - 1 - m];
        uint64_t key = 0, k = 0;
        int cc = -1, dst = 0;
        for (int r = 0; r < g.n; ++r) {
            const uint32_t d = g.run[r];
            const int c = (int)(d & 0xFFFFu), lo = (int)((d >> 16) & 63u), width = (int)((d >> 22) & 63u) + 1;
            if (c != cc) { cc = c; k = kc[(int64_t)cc * N + i]; }
            const uint64_t msk = width >= 64 ? ~0ull : (1ull << width) - 1ull;
            key |= ((k >> lo) & msk) << dst;
            dst += width;
        }
        kc[(int64_t)m * N + i] = key;
    }
}
__global__ __launch_bounds__(SB) void k_sort_gather(const uint64_t *__restrict__ pk, const uint32_t *__restrict__ vals, int64_t n,
                                                    uint64_t *__restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (i < n) keys[i] = pk[vals[i]];
}

// the groups, least significant first.  The FIRST one takes what is left over (total mod 64), so that the last -- whose keys
// k_sort_finish compares neighbours by -- holds the 64 most significant differing bits.
static std::vector<SortGroup> sort_groups(const uint64_t *orand, int nchunk) {
    struct Run { int chunk, lo, width; };
    std::vector<Run> runs;                                 // least significant first: last chunk, low bits
    int total = 0;
    for (int c = nchunk - 1; c >= 0; --c) {
        const uint64_t vary = orand[2 * c] ^ orand[2 * c + 1];
        for (int b = 0; b < 64;) {
            if (!((vary >> b) & 1ull)) { ++b; continue; }
            int e = b;
            while (e < 64 && ((vary >> e) & 1ull)) ++e;
            runs.push_back(Run{c, b, e - b});
            total += e - b;
            b = e;
        }
    }
    std::vector<SortGroup> out;
    if (!total) return out;
    SortGroup g;
    memset(&g, 0, sizeof g);
    int room = total % 64 ? total % 64 : 64;
    auto close = [&]() { out.push_back(g); memset(&g, 0, sizeof g); room = 64; };
    for (size_t r = 0; r < runs.size();) {
        const int w = runs[r].width < room ? runs[r].width : room;
        g.run[g.n++] = (uint32_t)runs[r].chunk | ((uint32_t)runs[r].lo << 16) | ((uint32_t)(w - 1) << 22);
        g.bits += w; room -= w;
        if (w == runs[r].width) ++r;
        else { runs[r].lo += w; runs[r].width -= w; }
        if (room == 0) close();
    }
    if (g.n) close();
    return out;
}

// ------------------------------------------------------------------ stable 8-bit counting pass over (key, value) pairs
// A workgroup of 4 waves owns a tile of RS_TILE = 2048 consecutive pairs, wave w the 512 pairs [w * 512, ...) of it in
// 8 rounds of 64 consecutive pairs -- so "earlier in the input" is (workgroup, wave, round, lane), and a pair's place
// in the output is   start of its digit's bucket  +  pairs with that digit in earlier workgroups  (k_rs_scan)
//                  + ... in earlier waves of the workgroup + ... in earlier rounds of the wave + ... in lower lanes.
#ifndef FX_RS_ROUNDS
#define FX_RS_ROUNDS 8                                   // 2048 pairs per tile (28 KB of LDS, five workgroups per CU): 10^8 names 19.0-19.5 ms; 12: 19.6; 16 (53 KB, two per CU): 20.2-21.0
#endif
constexpr int RS_BLOCK = 256, RS_ROUNDS = FX_RS_ROUNDS, RS_TILE = RS_BLOCK * RS_ROUNDS;

__device__ __forceinline__ uint32_t rs_incl_scan64(uint32_t v) {     // inclusive prefix sum over the wave (DPP, as wave_incl_scan)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
// lanes of the wave that hold the same 8-bit digit as this one (and are valid): eight ballots
__device__ __forceinline__ uint64_t rs_peers(uint32_t d, bool valid) {
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

// hist[bin * nblk + blk] = pairs of tile blk whose digit is bin
__global__ __launch_bounds__(RS_BLOCK) void k_rs_hist(const uint64_t *__restrict__ keys, int64_t n, int shift, int64_t nblk,
                                                      uint32_t *__restrict__ hist) {
    __shared__ uint32_t h[256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    h[tid] = 0;
    __syncthreads();
    const int64_t start = (int64_t)blockIdx.x * RS_TILE + w * (RS_TILE / 4);
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const int64_t i = start + r * 64 + lane;
        const bool valid = i < n;
        const uint32_t d = valid ? (uint32_t)(keys[i] >> shift) & 255u : 0u;
        const uint64_t peers = rs_peers(d, valid);     // one LDS atomic per distinct digit of the wave, not per pair
        if (valid && (peers & ((1ull << lane) - 1)) == 0) atomicAdd(&h[d], (uint32_t)__popcll(peers));
    }
    __syncthreads();
    hist[(int64_t)tid * nblk + blockIdx.x] = h[tid];
}

// one workgroup per bin: hist[bin][*] -> exclusive prefix over the tiles (in place), totals[bin] = its sum
__global__ __launch_bounds__(RS_BLOCK) void k_rs_scan(uint32_t *__restrict__ hist, int64_t nblk, uint32_t *__restrict__ totals) {
    __shared__ uint32_t wsum[RS_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint32_t *row = hist + (int64_t)blockIdx.x * nblk;
    const int64_t per = (nblk + RS_BLOCK - 1) / RS_BLOCK, a = tid * per, b = a + per < nblk ? a + per : nblk;
    uint32_t s = 0;
    for (int64_t j = a; j < b; ++j) s += row[j];
    const uint32_t inc = rs_incl_scan64(s);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = inc - s;
    for (int k = 0; k < w; ++k) base += wsum[k];
    for (int64_t j = a; j < b; ++j) { const uint32_t v = row[j]; row[j] = base; base += v; }
    if (tid == RS_BLOCK - 1) totals[blockIdx.x] = base;
}

// The tile is put in digit order in LDS first and leaves from there: neighbouring lanes then write neighbouring places of a
// bucket (a tile holds 16 pairs per digit on average: runs of 128 + 64 bytes) where, straight from the registers, every
// store instruction went to 64 different places (2.0 ms per pass for 10^8 pairs).
__global__ __launch_bounds__(RS_BLOCK) void k_rs_scatter(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                         uint64_t *__restrict__ kout, uint32_t *__restrict__ vout, int64_t n, int shift,
                                                         int64_t nblk, const uint32_t *__restrict__ offs, const uint32_t *__restrict__ totals) {
    __shared__ uint64_t lk[RS_TILE];
    __shared__ uint32_t lv[RS_TILE];
    __shared__ uint32_t cnt[RS_BLOCK / 64][256];          // per wave: pairs seen so far per digit; then: pairs of the digit in the waves before
    __shared__ uint32_t lbase[256];                       // where the digit starts in the tile once it is in digit order
    __shared__ uint32_t gdelta[256];                      // place in the output - place in the ordered tile, per digit
    __shared__ uint32_t wsum[RS_BLOCK / 64], wsum2[RS_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int k = 0; k < RS_BLOCK / 64; ++k) cnt[k][tid] = 0;
    uint32_t gbase;
    {                                                     // bucket starts: exclusive scan of the 256 totals
        const uint32_t t = totals[tid], inc = rs_incl_scan64(t);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t b = inc - t;
        for (int k = 0; k < w; ++k) b += wsum[k];
        gbase = b + offs[(int64_t)tid * nblk + blockIdx.x];    // where this tile's pairs of digit `tid` start in the output
    }
    uint64_t k[RS_ROUNDS];
    uint32_t v[RS_ROUNDS], dr[RS_ROUNDS];                 // digit << 16 | rank among the wave's pairs of that digit
    const int64_t start = (int64_t)blockIdx.x * RS_TILE + w * (RS_TILE / 4);
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const int64_t i = start + r * 64 + lane;
        const bool valid = i < n;
        k[r] = valid ? kin[i] : 0ull;
        v[r] = valid ? vin[i] : 0u;
        const uint32_t d = (uint32_t)(k[r] >> shift) & 255u;
        const uint64_t peers = rs_peers(d, valid);
        const uint32_t below = (uint32_t)__popcll(peers & ((1ull << lane) - 1));
        const uint32_t prior = cnt[w][d];                 // all lanes read before the first lane of each digit adds
        if (valid && below == 0) cnt[w][d] = prior + (uint32_t)__popcll(peers);
        dr[r] = (d << 16) | (prior + below);
    }
    __syncthreads();
    {                                                     // thread `tid` = digit: the waves' counts become prefixes, the tile's total is scanned over the digits
        uint32_t run = 0;
#pragma unroll
        for (int q = 0; q < RS_BLOCK / 64; ++q) { const uint32_t c = cnt[q][tid]; cnt[q][tid] = run; run += c; }
        const uint32_t inc = rs_incl_scan64(run);
        if (lane == 63) wsum2[w] = inc;
        __syncthreads();
        uint32_t b = inc - run;
        for (int q = 0; q < w; ++q) b += wsum2[q];
        lbase[tid] = b;
        gdelta[tid] = gbase - b;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const int64_t i = start + r * 64 + lane;
        if (i < n) {
            const uint32_t d = dr[r] >> 16;
            const uint32_t lp = lbase[d] + cnt[w][d] + (dr[r] & 0xFFFFu);
            lk[lp] = k[r];
            lv[lp] = v[r];
        }
    }
    __syncthreads();
    const int64_t left = n - (int64_t)blockIdx.x * RS_TILE;
    const uint32_t count = left < RS_TILE ? (uint32_t)left : (uint32_t)RS_TILE;
#pragma unroll 4
    for (uint32_t j = tid; j < count; j += RS_BLOCK) {
        const uint64_t key = lk[j];
        const uint32_t pos = j + gdelta[(uint32_t)(key >> shift) & 255u];
        kout[pos] = key;
        vout[pos] = lv[j];
    }
}

// order[i] = row of the i-th smallest name; s_off / s_len (may be null): the offset and the length of that name, in sorted
// order -- the index kernels of fx_fxi_dev.hpp then gather nothing but the name itself; *ndup += 1 per adjacent equal pair
// (equal length, equal in every packed group = equal in every bit that differs anywhere: keys0 = what the last round sorted
// by -- the most significant group, or the lengths when no bit differs --, coalesced; only neighbours equal in it go to the
// other groups, kc[1..nchunk) after k_sort_pack)
__global__ __launch_bounds__(SB) void k_sort_finish(const uint64_t *__restrict__ kc, int nchunk, const uint64_t *__restrict__ keys0,
                                                    const int64_t *__restrict__ name_off, const int32_t *__restrict__ name_len,
                                                    const uint32_t *__restrict__ vals, int64_t n, int64_t *__restrict__ order,
                                                    int64_t *__restrict__ s_off, int32_t *__restrict__ s_len, int64_t *__restrict__ ndup) {
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = vals[i];
    order[i] = (int64_t)a;
    const int la = name_len[a] > 0 ? name_len[a] : 0;
    if (s_off) { s_off[i] = name_off[a]; s_len[i] = la; }
    if (i + 1 >= n) return;
    if (keys0[i] != keys0[i + 1]) return;
    const uint32_t b = vals[i + 1];
    const int lb = name_len[b] > 0 ? name_len[b] : 0;
    if (la != lb) return;
    for (int c = 1; c < nchunk; ++c)
        if (kc[(int64_t)c * n + a] != kc[(int64_t)c * n + b]) return;
    atomicAdd(reinterpret_cast<unsigned long long *>(ndup), 1ull);
}

#define SORTCHK(expr, what)                      \
    do {                                         \
        hipError_t e__ = (expr);                 \
        if (e__ != hipSuccess) { *where = what; cleanup(); return (int)e__; } \
    } while (0)

struct RadixScratch { uint32_t *hist = nullptr, *totals = nullptr; int64_t nblk = 0; };

// sort the pairs by bits [begin_bit, end_bit) of the key, stable; cur: index of the buffer pair that holds the data,
// updated.  Digits that are equal in all keys cost a histogram but no data movement.
static hipError_t radix_sort_pairs(uint64_t *keys[2], uint32_t *vals[2], int &cur, int64_t n, int begin_bit, int end_bit,
                                   const RadixScratch &sc, hipStream_t s) {
    uint32_t totals[256];
    for (int shift = begin_bit; shift < end_bit; shift += 8) {
        hipLaunchKernelGGL(k_rs_hist, dim3((unsigned)sc.nblk), dim3(RS_BLOCK), 0, s, keys[cur], n, shift, sc.nblk, sc.hist);
        hipLaunchKernelGGL(k_rs_scan, dim3(256), dim3(RS_BLOCK), 0, s, sc.hist, sc.nblk, sc.totals);
        hipError_t e = hipMemcpyAsync(totals, sc.totals, sizeof totals, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return e;
        bool single = false;
        for (int b = 0; b < 256; ++b) single = single || (int64_t)totals[b] == n;
        if (single) continue;
        hipLaunchKernelGGL(k_rs_scatter, dim3((unsigned)sc.nblk), dim3(RS_BLOCK), 0, s, keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1],
                           n, shift, sc.nblk, sc.hist, sc.totals);
        cur ^= 1;
    }
    return hipGetLastError();
}

int sort_names(const uint8_t *data, int64_t gbase, const int64_t *name_off, const int32_t *name_len, int64_t n,
               int64_t *d_order, int64_t *d_ndup, hipStream_t s, const char **where, int64_t *d_soff, int32_t *d_slen) {
    uint64_t *keys[2] = {nullptr, nullptr}, *kc = nullptr;
    uint32_t *vals[2] = {nullptr, nullptr};
    RadixScratch sc;
    unsigned *d_max = nullptr;
    // (out of the library's scratch pool: 2.4 GB of hipMalloc + hipFree per sort of 10^8 names were milliseconds of waiting for the device)
    struct Blk { void *p; size_t cap; };
    std::vector<Blk> held;
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto take = [&](void **p, size_t bytes) -> hipError_t {
        size_t cap = 0;
        *p = scratch_get(dev, bytes, &cap);
        if (!*p) return hipErrorOutOfMemory;
        held.push_back(Blk{*p, cap});
        return hipSuccess;
    };
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(s);
        for (auto &b : held) scratch_put(dev, b.p, b.cap);
        held.clear();
    };
    *where = "";
    SORTCHK(hipMemsetAsync(d_ndup, 0, 8, s), "memset");
    if (n <= 0) return 0;
    const size_t N = (size_t)n;
    for (int k = 0; k < 2; ++k) {
        SORTCHK(take((void **)&keys[k], N * 8), "hipMalloc(sort keys)");
        SORTCHK(take((void **)&vals[k], N * 4), "hipMalloc(sort values)");
    }
    sc.nblk = (n + RS_TILE - 1) / RS_TILE;
    SORTCHK(take((void **)&sc.hist, (size_t)sc.nblk * 256 * 4), "hipMalloc(sort histograms)");
    SORTCHK(take((void **)&sc.totals, 256 * 4), "hipMalloc");
    SORTCHK(take((void **)&d_max, 4), "hipMalloc");
    SORTCHK(hipMemsetAsync(d_max, 0, 4, s), "memset");
    const unsigned nb = (unsigned)((n + SB - 1) / SB);
    hipLaunchKernelGGL(k_sort_init, dim3(nb < 4096u ? nb : 4096u), dim3(SB), 0, s, name_len, n, keys[0], vals[0], d_max);
    unsigned max_len = 0;
    SORTCHK(hipMemcpyAsync(&max_len, d_max, 4, hipMemcpyDeviceToHost, s), "memcpy");
    SORTCHK(hipStreamSynchronize(s), "k_sort_init");
    const int nchunk = (int)((max_len + 7) / 8);
    if (nchunk > 0xFFFF) { *where = "a name of more than 512 KiB"; cleanup(); return (int)hipErrorInvalidValue; }
    std::vector<uint64_t> orand((size_t)nchunk * 2);
    if (nchunk) {
        unsigned long long *d_orand = nullptr;
        SORTCHK(take((void **)&kc, N * 8 * (size_t)nchunk), "hipMalloc(name chunks)");
        SORTCHK(take((void **)&d_orand, orand.size() * 8), "hipMalloc");
        for (int c = 0; c < nchunk; ++c) { orand[2 * c] = 0; orand[2 * c + 1] = ~0ull; }
        SORTCHK(hipMemcpyAsync(d_orand, orand.data(), orand.size() * 8, hipMemcpyHostToDevice, s), "memcpy");
        const int64_t ntiles = (n * nchunk + SB - 1) / SB;
        const int nblk = (int)(ntiles < 8192 ? ntiles : 8192);
        unsigned long long *d_part = nullptr;
        if (nchunk <= SORT_LDS_CHUNKS) SORTCHK(take((void **)&d_part, (size_t)nblk * nchunk * 16), "hipMalloc");
        hipLaunchKernelGGL(k_sort_chunks, dim3((unsigned)nblk), dim3(SB), 0, s, data, gbase, name_off, name_len, n, nchunk, kc, d_part, d_orand);
        if (d_part) hipLaunchKernelGGL(k_sort_orand, dim3((unsigned)nchunk), dim3(SB), 0, s, d_part, nblk, nchunk, d_orand);
        SORTCHK(hipMemcpyAsync(orand.data(), d_orand, orand.size() * 8, hipMemcpyDeviceToHost, s), "memcpy");
    }
    int cur = 0;
    int len_bits = 8;
    while (len_bits < 32 && (max_len >> len_bits)) len_bits += 8;
    SORTCHK(radix_sort_pairs(keys, vals, cur, n, 0, len_bits, sc, s), "radix passes (length)");     // (synchronises: orand is on the host)
    const std::vector<SortGroup> groups = sort_groups(orand.data(), nchunk);
    const int G = (int)groups.size();                      // <= nchunk
    if (G) {
        SortGroup *d_groups = nullptr;
        SORTCHK(take((void **)&d_groups, groups.size() * sizeof(SortGroup)), "hipMalloc");
        SORTCHK(hipMemcpyAsync(d_groups, groups.data(), groups.size() * sizeof(SortGroup), hipMemcpyHostToDevice, s), "memcpy");
        for (int m0 = 0; m0 < G; m0 += SORT_PACK_GROUPS)
            hipLaunchKernelGGL(k_sort_pack, dim3(nb), dim3(SB), 0, s, kc, (int64_t)N, n, d_groups, G, m0, m0 + SORT_PACK_GROUPS < G ? m0 + SORT_PACK_GROUPS : G);
    }
    for (int g = 0; g < G; ++g) {
        hipLaunchKernelGGL(k_sort_gather, dim3(nb), dim3(SB), 0, s, kc + (size_t)(G - 1 - g) * N, vals[cur], n, keys[cur]);
        SORTCHK(radix_sort_pairs(keys, vals, cur, n, 0, groups[g].bits, sc, s), "radix passes (names)");
    }
    hipLaunchKernelGGL(k_sort_finish, dim3(nb), dim3(SB), 0, s, kc, G, keys[cur], name_off, name_len, vals[cur], n, d_order, d_soff, d_slen, d_ndup);
    SORTCHK(hipGetLastError(), "k_sort_finish");
    SORTCHK(hipStreamSynchronize(s), "sort");
    cleanup();
    return 0;
}

// ------------------------------------------------------------------ statistics of the record lengths
__global__ __launch_bounds__(SB) void k_len_init(const int64_t *__restrict__ slen, int64_t n, uint64_t *__restrict__ keys,
                                                 uint32_t *__restrict__ vals, unsigned long long *__restrict__ mx) {
    __shared__ unsigned long long blk;
    if (threadIdx.x == 0) blk = 0;
    __syncthreads();
    unsigned long long m = 0;
    for (int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x; i < n; i += (int64_t)gridDim.x * SB) {
        const unsigned long long v = slen[i] > 0 ? (unsigned long long)slen[i] : 0ull;
        keys[i] = v; vals[i] = (uint32_t)i;
        m = v > m ? v : m;
    }
    if (m) atomicMax(&blk, m);
    __syncthreads();
    if (threadIdx.x == 0 && blk) atomicMax(mx, blk);
}

constexpr int LS_CHUNK = 2048;                            // sorted lengths per workgroup of the scan kernels (8 per thread)
__global__ __launch_bounds__(SB) void k_len_chunk_sums(const uint64_t *__restrict__ a, int64_t n, unsigned long long *__restrict__ sums) {
    __shared__ unsigned long long blk;
    if (threadIdx.x == 0) blk = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * LS_CHUNK;
    unsigned long long s = 0;
    for (int k = threadIdx.x; k < LS_CHUNK; k += SB) if (base + k < n) s += a[base + k];
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(&blk, s);
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = blk;
}
// one workgroup: sums[0..nchunks) -> exclusive prefix in place, sums[nchunks] = total
__global__ __launch_bounds__(SB) void k_len_chunk_bases(unsigned long long *__restrict__ sums, int64_t nchunks) {
    __shared__ unsigned long long part[SB];
    const int64_t per = (nchunks + SB - 1) / SB, a = threadIdx.x * per, b = a + per < nchunks ? a + per : nchunks;
    unsigned long long s = 0;
    for (int64_t j = a; j < b; ++j) s += sums[j];
    part[threadIdx.x] = s;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
    for (int k = 0; k < SB; ++k) { if (k < (int)threadIdx.x) base += part[k]; tot += part[k]; }
    for (int64_t j = a; j < b; ++j) { const unsigned long long v = sums[j]; sums[j] = base; base += v; }
    __syncthreads();
    if (threadIdx.x == 0) sums[nchunks] = tot;
}
// every question is a boundary in the sorted array: the thread that sits on it writes the answer
__global__ __launch_bounds__(SB) void k_len_probe(const uint64_t *__restrict__ a, const uint32_t *__restrict__ idx, int64_t n,
                                                  const unsigned long long *__restrict__ bases, int64_t nchunks, unsigned long long count_min,
                                                  double half, LenStats *__restrict__ out) {
    __shared__ unsigned long long wtot[SB / 64];
    const int64_t base = (int64_t)blockIdx.x * LS_CHUNK;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long v[8], s = 0;                         // thread t owns elements base + 8t .. 8t+7
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int64_t i = base + 8 * threadIdx.x + k; v[k] = i < n ? a[i] : 0ull; s += v[k]; }
    unsigned long long inc = s;
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    unsigned long long excl = bases[blockIdx.x] + inc - s;  // sum of the sorted lengths before this thread's first element
    for (int k = 0; k < w; ++k) excl += wtot[k];
    const unsigned long long total = bases[nchunks];
    const uint64_t amax = a[n - 1];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int64_t i = base + 8 * threadIdx.x + k;
        if (i < n) {
            const uint64_t prev = i ? (k ? v[k - 1] : a[i - 1]) : 0;
            if (i == 0) { out->shortest_id = idx[0]; out->shortest_len = (long long)v[k]; out->n = n; out->sum = (long long)total; }
            if (v[k] == amax && (i == 0 || prev != amax)) { out->longest_id = idx[i]; out->longest_len = (long long)amax; }
            if (v[k] >= count_min && (i == 0 || prev < count_min)) out->count_ge = n - i;
            if (i == (n - 1) / 2) { out->med_lo = (long long)v[k]; out->med_hi = (long long)((n % 2 == 0) ? a[i + 1] : v[k]); }
            // descending walk: this element is number n - i; running sum with it = total - excl, without it = that - v
            const unsigned long long with = total - excl, without = with - v[k];
            if ((double)with >= half && (i == n - 1 || !((double)without >= half))) { out->nx_len = (long long)v[k]; out->nx_count = n - i; }
        }
        excl += v[k];
    }
}

int len_stats(const int64_t *d_slen, int64_t n, int64_t count_min, double half, LenStats *host_out, hipStream_t s, const char **where) {
    uint64_t *keys[2] = {nullptr, nullptr};
    uint32_t *vals[2] = {nullptr, nullptr};
    RadixScratch sc;
    unsigned long long *d_max = nullptr, *d_sums = nullptr;
    LenStats *d_out = nullptr;
    auto cleanup = [&]() {
        for (int k = 0; k < 2; ++k) { if (keys[k]) (void)hipFree(keys[k]); if (vals[k]) (void)hipFree(vals[k]); }
        if (sc.hist) (void)hipFree(sc.hist);
        if (sc.totals) (void)hipFree(sc.totals);
        if (d_max) (void)hipFree(d_max);
        if (d_sums) (void)hipFree(d_sums);
        if (d_out) (void)hipFree(d_out);
    };
    *where = "";
    memset(host_out, 0, sizeof *host_out);
    if (n <= 0) return 0;
    if (n >= 0xFFFFFFFFll) { *where = "too many records for the 32-bit sort index"; return (int)hipErrorInvalidValue; }
    const size_t N = (size_t)n;
    for (int k = 0; k < 2; ++k) {
        SORTCHK(pool_malloc((void **)&keys[k], N * 8), "hipMalloc(sort keys)");
        SORTCHK(pool_malloc((void **)&vals[k], N * 4), "hipMalloc(sort values)");
    }
    sc.nblk = (n + RS_TILE - 1) / RS_TILE;
    const int64_t nchunks = (n + LS_CHUNK - 1) / LS_CHUNK;
    SORTCHK(pool_malloc((void **)&sc.hist, (size_t)sc.nblk * 256 * 4), "hipMalloc(sort histograms)");
    SORTCHK(pool_malloc((void **)&sc.totals, 256 * 4), "hipMalloc");
    SORTCHK(pool_malloc((void **)&d_max, 8), "hipMalloc");
    SORTCHK(pool_malloc((void **)&d_sums, (size_t)(nchunks + 1) * 8), "hipMalloc");
    SORTCHK(pool_malloc((void **)&d_out, sizeof(LenStats)), "hipMalloc");
    SORTCHK(hipMemsetAsync(d_max, 0, 8, s), "memset");
    SORTCHK(hipMemsetAsync(d_out, 0, sizeof(LenStats), s), "memset");
    const unsigned nb = (unsigned)((n + SB - 1) / SB);
    hipLaunchKernelGGL(k_len_init, dim3(nb < 4096u ? nb : 4096u), dim3(SB), 0, s, d_slen, n, keys[0], vals[0], d_max);
    unsigned long long mx = 0;
    SORTCHK(hipMemcpyAsync(&mx, d_max, 8, hipMemcpyDeviceToHost, s), "memcpy");
    SORTCHK(hipStreamSynchronize(s), "k_len_init");
    int bits = 8;
    while (bits < 64 && (mx >> bits)) bits += 8;
    int cur = 0;
    SORTCHK(radix_sort_pairs(keys, vals, cur, n, 0, bits, sc, s), "radix passes (lengths)");
    hipLaunchKernelGGL(k_len_chunk_sums, dim3((unsigned)nchunks), dim3(SB), 0, s, keys[cur], n, d_sums);
    hipLaunchKernelGGL(k_len_chunk_bases, dim3(1), dim3(SB), 0, s, d_sums, nchunks);
    hipLaunchKernelGGL(k_len_probe, dim3((unsigned)nchunks), dim3(SB), 0, s, keys[cur], vals[cur], n, d_sums, nchunks,
                       (unsigned long long)(count_min > 0 ? count_min : 0), half, d_out);
    SORTCHK(hipGetLastError(), "length statistics kernels");
    SORTCHK(hipMemcpyAsync(host_out, d_out, sizeof(LenStats), hipMemcpyDeviceToHost, s), "memcpy");
    SORTCHK(hipStreamSynchronize(s), "length statistics");
    cleanup();
    return 0;
}

}  // namespace fx
