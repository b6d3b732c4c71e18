// fx_sort.hpp -- interface of fx_sort.hip (order of the record names: a hand-written LSD radix sort, its own unit).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace fx {

// Sorted order of n names that live in the resident stream (name i = name_len[i] bytes at data + name_off[i] - gbase)
// in SQLite's BINARY collation: memcmp over the common length, the shorter name first on a tie.  d_order[i] (device,
// int64) = 0-based index of the i-th smallest name, equal names in index order; *d_ndup (device) = number of
// adjacent equal pairs in that order (0 <=> all names distinct).  Enqueued on `s`; returns a hipError_t.
// *where names the failing step.
int sort_names(const uint8_t *data, int64_t gbase, const int64_t *name_off, const int32_t *name_len, int64_t n,
               int64_t *d_order, int64_t *d_ndup, hipStream_t s, const char **where);

}  // namespace fx
