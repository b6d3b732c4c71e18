// fx_names.hpp -- name -> record id in HBM.
//
// The reference resolves `fa['chr1']` / `fq['read name']` with one SQLite probe per call
// (index.c:527-566, fastq.c:486-519: `SELECT * FROM seq WHERE chrom=? LIMIT 1` over the UNIQUE
// INDEX).  For a batch that is a million B-tree descents on one CPU thread.  Here the names
// never leave the resident stream: an open-addressing table of record ids (uint32, load <= 0.5)
// is built from (name_off, name_len) of the record table -- the key of a slot IS the bytes of
// the stream it points to -- and a batch of query names is resolved by one thread per query:
// hash, linear probe, compare against the stream.  Duplicate names resolve to the lowest id
// (what `LIMIT 1` returns from the rowid-ordered table when the unique index could not be made).
#pragma once
#include "fx_kernels.hpp"

namespace fx {

typedef uint64_t __attribute__((aligned(1))) u64_any;

// 64-bit hash of n bytes at p (8 at a time, tail masked); the same function hashes stream names and queries
__device__ __forceinline__ uint64_t name_hash(const uint8_t *p, int64_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
    int64_t i = 0;
    for (; i + 8 <= n; i += 8) {
        h ^= *reinterpret_cast<const u64_any *>(p + i);
        h *= 0xFF51AFD7ED558CCDull; h ^= h >> 32;
    }
    if (i < n) {
        uint64_t w = 0;
        if (n >= 8) w = *reinterpret_cast<const u64_any *>(p + n - 8) >> (8 * (8 - (n - i)));      // the 8 bytes that end with the name: one load, not a loop of byte loads
        else for (int k = 0; i + k < n; ++k) w |= (uint64_t)p[i + k] << (8 * k);
        h ^= w;
        h *= 0xFF51AFD7ED558CCDull; h ^= h >> 32;
    }
    h *= 0xC4CEB9FE1A85EC53ull; h ^= h >> 29;
    return h;
}
__device__ __forceinline__ bool bytes_equal(const uint8_t *a, const uint8_t *b, int64_t n) {
    int64_t i = 0;
    for (; i + 8 <= n; i += 8)
        if (*reinterpret_cast<const u64_any *>(a + i) != *reinterpret_cast<const u64_any *>(b + i)) return false;
    if (i < n && n >= 8) return *reinterpret_cast<const u64_any *>(a + n - 8) == *reinterpret_cast<const u64_any *>(b + n - 8);   // the tail: the 8 bytes that end both
    for (; i < n; ++i) if (a[i] != b[i]) return false;
    return true;
}

// name_len: int32 per record; name_off: global offsets (gbase subtracted for the blob)
__global__ __launch_bounds__(BLOCK) void k_names_build(const uint8_t *__restrict__ data, int64_t gbase,
                                                      const int64_t *__restrict__ name_off, const int32_t *__restrict__ name_len,
                                                      int64_t n, uint32_t *__restrict__ table, uint64_t mask) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const int64_t len = name_len[i];
    if (len < 0) return;
    uint64_t s = name_hash(data + (name_off[i] - gbase), len) & mask;
    while (atomicCAS(&table[s], 0u, (uint32_t)(i + 1)) != 0u) s = (s + 1) & mask;
}

// out[q] = lowest record id whose name equals query q, or -1
__global__ __launch_bounds__(BLOCK) void k_names_lookup(const uint8_t *__restrict__ data, int64_t gbase,
                                                       const int64_t *__restrict__ name_off, const int32_t *__restrict__ name_len,
                                                       const uint32_t *__restrict__ table, uint64_t mask,
                                                       const uint8_t *__restrict__ qbytes, const int64_t *__restrict__ qoff,
                                                       int64_t nq, int64_t *__restrict__ out) {
    const int64_t q = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (q >= nq) return;
    const uint8_t *key = qbytes + qoff[q];
    const int64_t len = qoff[q + 1] - qoff[q];
    int64_t best = -1;
    for (uint64_t s = name_hash(key, len) & mask;; s = (s + 1) & mask) {
        const uint32_t v = table[s];
        if (!v) break;
        const int64_t id = (int64_t)v - 1;
        if (name_len[id] == len && (best < 0 || id < best) && bytes_equal(data + (name_off[id] - gbase), key, len)) best = id;
    }
    out[q] = best;
}

}  // namespace fx
