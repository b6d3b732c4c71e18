// fx_sort.hpp -- interface of fx_sort.hip (order of the record names: a hand-written LSD radix sort, its own unit).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace fx {

// hipMalloc that empties the library's idle scratch pool and tries again when memory is short (fxgpu.hip)
hipError_t pool_malloc(void **p, size_t bytes);
// a block of the library's scratch pool (fxgpu.hip: ScratchPool) / back into it
void *scratch_get(int device, size_t bytes, size_t *cap);
void scratch_put(int device, void *p, size_t cap);

// Sorted order of n names that live in the resident stream (name i = name_len[i] bytes at data + name_off[i] - gbase)
// in SQLite's BINARY collation: memcmp over the common length, the shorter name first on a tie.  d_order[i] (device,
// int64) = 0-based index of the i-th smallest name, equal names in index order; *d_ndup (device) = number of
// adjacent equal pairs in that order (0 <=> all names distinct).  d_soff / d_slen (device, n each; both or neither): the
// offset and the length of the i-th smallest name -- name_off / name_len through the order, so that what formats the index
// from it gathers nothing but the names.  Enqueued on `s`; returns a hipError_t.  *where names the failing step.
int sort_names(const uint8_t *data, int64_t gbase, const int64_t *name_off, const int32_t *name_len, int64_t n,
               int64_t *d_order, int64_t *d_ndup, hipStream_t s, const char **where, int64_t *d_soff = nullptr, int32_t *d_slen = nullptr);

// Statistics of the record lengths (fasta.c:573-849: count(n), nl(p), longest, shortest, mean, median --
// the reference asks SQLite to sort / scan the seq table for each of them): ONE stable radix sort of (slen, id) on the
// GPU, a scan of the sorted lengths and one probe kernel answer them all.  d_slen: n lengths (>= 0) in HBM.
struct LenStats {
    long long n, sum;
    long long longest_id, longest_len;      // FIRST record with the maximum length (SQLite's MAX() keeps the first row it met)
    long long shortest_id, shortest_len;    // first record with the minimum length
    long long count_ge;                     // records with slen >= count_min
    long long med_lo, med_hi;               // sorted[(n-1)/2], sorted[(n-1)/2 + 1] (= med_lo when n is odd): fasta.c:786-849
    long long nx_len, nx_count;             // walking the lengths in descending order: the first (length, lengths so far)
                                            // whose running sum >= half (a double, fasta.c:630-647); 0, 0 when none does
};
int len_stats(const int64_t *d_slen, int64_t n, int64_t count_min, double half, LenStats *host_out, hipStream_t s, const char **where);

}  // namespace fx
