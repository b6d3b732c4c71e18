// fx_sort.hip -- order of the record names for the UNIQUE INDEX of a .fxi.
//
// The reference leaves this to SQLite: `CREATE UNIQUE INDEX readidx ON read (name)` (fastq.c:152, index.c:363) runs
// an external merge sort over every (name, rowid) on one CPU thread after the inserts -- for the 10^8 reads of a
// sequencing run that is minutes.  Here the names are already in HBM (they are slices of the resident stream), so
// the order is an LSD radix sort over them in place: the names are cut into 8-byte big-endian chunks (zero padded),
// a permutation is stable-sorted by name length first and then by chunk  ceil(maxlen/8)-1, ..., 1, 0; after the last
// pass it is ordered by (chunk 0, chunk 1, ..., length) = memcmp order with the shorter name first on a tie, which is
// SQLite's BINARY collation.  Each chunk gathers one 8-byte key per name through the current permutation (a random
// 8-byte read per name) and sorts the (key, index) pairs by it with up to eight stable 8-bit counting passes
// (k_rs_hist / k_rs_scan / k_rs_scatter below); a pass whose digit is the same in every key -- the shared prefix of
// sequencer read names -- is detected from its histogram and skipped.  A final kernel compares neighbours to count
// duplicate names.  The permutation goes to fx_fxi_bulk_index, which writes the index b-tree from it.
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
__global__ __launch_bounds__(SB) void k_sort_chunks(const uint8_t *__restrict__ data, int64_t gbase, const int64_t *__restrict__ name_off,
                                                    const int32_t *__restrict__ name_len, int64_t n, int nchunk, uint64_t *__restrict__ kc) {
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (i >= n) return;
    const int len = name_len[i] > 0 ? name_len[i] : 0;
    const uint8_t *p = data + (name_off[i] - gbase);
    for (int c = 0; c < nchunk; ++c) {
        const int rest = len - 8 * c;
        uint64_t k = 0;
        if (rest >= 8) k = __builtin_bswap64(*reinterpret_cast<const u64_unal *>(p + 8 * c));
        else if (rest > 0) for (int b = 0; b < rest; ++b) k |= (uint64_t)p[8 * c + b] << (56 - 8 * b);
        kc[(int64_t)c * n + i] = k;
    }
}
__global__ __launch_bounds__(SB) void k_sort_gather(const uint64_t *__restrict__ kc, const uint32_t *__restrict__ vals, int64_t n,
                                                    uint64_t *__restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (i < n) keys[i] = kc[vals[i]];
}

// ------------------------------------------------------------------ stable 8-bit counting pass over (key, value) pairs
// A workgroup of 4 waves owns a tile of RS_TILE = 4096 consecutive pairs, wave w the 1024 pairs [w * 1024, ...) of it in
// 16 rounds of 64 consecutive pairs -- so "earlier in the input" is (workgroup, wave, round, lane), and a pair's place
// in the output is   start of its digit's bucket  +  pairs with that digit in earlier workgroups  (k_rs_scan)
//                  + ... in earlier waves of the workgroup + ... in earlier rounds of the wave + ... in lower lanes.
constexpr int RS_BLOCK = 256, RS_ROUNDS = 16, RS_TILE = RS_BLOCK * RS_ROUNDS;

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

__global__ __launch_bounds__(RS_BLOCK) void k_rs_scatter(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                         uint64_t *__restrict__ kout, uint32_t *__restrict__ vout, int64_t n, int shift,
                                                         int64_t nblk, const uint32_t *__restrict__ offs, const uint32_t *__restrict__ totals) {
    __shared__ uint32_t cnt[RS_BLOCK / 64][256];          // per wave: pairs seen so far per digit
    __shared__ uint32_t base[256];                        // where this tile's pairs of a digit start in the output
    __shared__ uint32_t wsum[RS_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int k = 0; k < RS_BLOCK / 64; ++k) cnt[k][tid] = 0;
    {                                                     // bucket starts: exclusive scan of the 256 totals
        const uint32_t t = totals[tid], inc = rs_incl_scan64(t);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t b = inc - t;
        for (int k = 0; k < w; ++k) b += wsum[k];
        base[tid] = b + offs[(int64_t)tid * nblk + blockIdx.x];
    }
    __syncthreads();
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
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const int64_t i = start + r * 64 + lane;
        if (i < n) {
            const uint32_t d = dr[r] >> 16;
            uint32_t pos = base[d] + (dr[r] & 0xFFFFu);
            for (int q = 0; q < w; ++q) pos += cnt[q][d];
            kout[pos] = k[r];
            vout[pos] = v[r];
        }
    }
}

// order[i] = row of the i-th smallest name; s_off / s_len (may be null): the offset and the length of that name, in sorted
// order -- the index kernels of fx_fxi_dev.hpp then gather nothing but the name itself; *ndup += 1 per adjacent equal pair
// (equal length, equal in every chunk: chunk 0 is what the last pass sorted by and lies in keys0, coalesced)
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
    if (nchunk > 0 && keys0[i] != keys0[i + 1]) return;
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
    if (nchunk) {
        SORTCHK(take((void **)&kc, N * 8 * (size_t)nchunk), "hipMalloc(name chunks)");
        hipLaunchKernelGGL(k_sort_chunks, dim3(nb), dim3(SB), 0, s, data, gbase, name_off, name_len, n, nchunk, kc);
    }
    int cur = 0;
    int len_bits = 8;
    while (len_bits < 32 && (max_len >> len_bits)) len_bits += 8;
    SORTCHK(radix_sort_pairs(keys, vals, cur, n, 0, len_bits, sc, s), "radix passes (length)");
    for (int c = nchunk - 1; c >= 0; --c) {
        hipLaunchKernelGGL(k_sort_gather, dim3(nb), dim3(SB), 0, s, kc + (size_t)c * N, vals[cur], n, keys[cur]);
        SORTCHK(radix_sort_pairs(keys, vals, cur, n, 0, 64, sc, s), "radix passes (chunk)");
    }
    hipLaunchKernelGGL(k_sort_finish, dim3(nb), dim3(SB), 0, s, kc, nchunk, keys[cur], name_off, name_len, vals[cur], n, d_order, d_soff, d_slen, d_ndup);
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
