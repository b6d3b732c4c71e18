// fx_sort.hip -- order of the record names for the UNIQUE INDEX of a .fxi.
//
// The reference leaves this to SQLite: `CREATE UNIQUE INDEX readidx ON read (name)` (fastq.c:152, index.c:363) runs
// an external merge sort over every (name, rowid) on one CPU thread after the inserts -- for the 10^8 reads of a
// sequencing run that is minutes.  Here the names are already in HBM (they are slices of the resident stream), so
// the order is an LSD radix sort over them in place: the names are cut into 8-byte big-endian chunks (zero padded),
// a permutation is stable-sorted by name length first and then by chunk  ceil(maxlen/8)-1, ..., 1, 0; after the last
// pass it is ordered by (chunk 0, chunk 1, ..., length) = memcmp order with the shorter name first on a tie, which is
// SQLite's BINARY collation.  Each pass gathers one 8-byte key per name through the current permutation (a random
// 8-byte read per name) and runs rocPRIM's stable radix sort on (key, index) pairs; a final pass compares neighbours
// to count duplicate names.  The permutation goes to fx_fxi_bulk_index, which writes the index b-tree from it.
#include <cstring>                     // before hip_runtime.h: rocPRIM's texture iterator uses memset unqualified
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "fx_sort.hpp"

namespace fx {

static constexpr int SB = 256;
typedef uint64_t __attribute__((aligned(1))) u64_unal;

__global__ __launch_bounds__(SB) void k_sort_init(const int32_t *__restrict__ name_len, int64_t n, uint64_t *__restrict__ keys,
                                                  uint32_t *__restrict__ vals, unsigned *__restrict__ max_len) {
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    int len = 0;
    if (i < n) {
        len = name_len[i] > 0 ? name_len[i] : 0;
        keys[i] = (uint64_t)len;
        vals[i] = (uint32_t)i;
    }
    for (int o = 32; o; o >>= 1) len = max(len, __shfl_xor(len, o));
    if ((threadIdx.x & 63) == 0 && len) atomicMax(max_len, (unsigned)len);
}

// key of name vals[i] for chunk c: its bytes [8c, 8c+8) as a big-endian number, zero padded past the end
__global__ __launch_bounds__(SB) void k_sort_keys(const uint8_t *__restrict__ data, int64_t gbase, const int64_t *__restrict__ name_off,
                                                  const int32_t *__restrict__ name_len, const uint32_t *__restrict__ vals, int64_t n,
                                                  int c, uint64_t *__restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = vals[i];
    const int rest = name_len[r] - 8 * c;
    uint64_t k = 0;
    if (rest > 0) {
        const uint8_t *p = data + (name_off[r] - gbase) + 8 * c;
        if (rest >= 8) k = __builtin_bswap64(*reinterpret_cast<const u64_unal *>(p));
        else for (int b = 0; b < rest; ++b) k |= (uint64_t)p[b] << (56 - 8 * b);
    }
    keys[i] = k;
}

__global__ __launch_bounds__(SB) void k_sort_finish(const uint8_t *__restrict__ data, int64_t gbase, const int64_t *__restrict__ name_off,
                                                    const int32_t *__restrict__ name_len, const uint32_t *__restrict__ vals, int64_t n,
                                                    int64_t *__restrict__ order, int64_t *__restrict__ ndup) {
    const int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = vals[i];
    order[i] = (int64_t)a;
    if (i + 1 >= n) return;
    const uint32_t b = vals[i + 1];
    const int la = name_len[a] > 0 ? name_len[a] : 0, lb = name_len[b] > 0 ? name_len[b] : 0;
    if (la != lb) return;
    const uint8_t *pa = data + (name_off[a] - gbase), *pb = data + (name_off[b] - gbase);
    int j = 0;
    for (; j + 8 <= la; j += 8)
        if (*reinterpret_cast<const u64_unal *>(pa + j) != *reinterpret_cast<const u64_unal *>(pb + j)) return;
    for (; j < la; ++j) if (pa[j] != pb[j]) return;
    atomicAdd(reinterpret_cast<unsigned long long *>(ndup), 1ull);
}

#define SORTCHK(expr, what)                      \
    do {                                         \
        hipError_t e__ = (expr);                 \
        if (e__ != hipSuccess) { *where = what; cleanup(); return (int)e__; } \
    } while (0)

int sort_names(const uint8_t *data, int64_t gbase, const int64_t *name_off, const int32_t *name_len, int64_t n,
               int64_t *d_order, int64_t *d_ndup, hipStream_t s, const char **where) {
    uint64_t *keys[2] = {nullptr, nullptr};
    uint32_t *vals[2] = {nullptr, nullptr};
    void *tmp = nullptr;
    unsigned *d_max = nullptr;
    auto cleanup = [&]() {
        for (int k = 0; k < 2; ++k) { if (keys[k]) (void)hipFree(keys[k]); if (vals[k]) (void)hipFree(vals[k]); }
        if (tmp) (void)hipFree(tmp);
        if (d_max) (void)hipFree(d_max);
    };
    *where = "";
    SORTCHK(hipMemsetAsync(d_ndup, 0, 8, s), "memset");
    if (n <= 0) return 0;
    const size_t N = (size_t)n;
    for (int k = 0; k < 2; ++k) {
        SORTCHK(hipMalloc((void **)&keys[k], N * 8), "hipMalloc(sort keys)");
        SORTCHK(hipMalloc((void **)&vals[k], N * 4), "hipMalloc(sort values)");
    }
    SORTCHK(hipMalloc((void **)&d_max, 4), "hipMalloc");
    size_t tmp_bytes = 0;
    SORTCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys[0], keys[1], vals[0], vals[1], N, 0u, 64u, s), "radix_sort_pairs(size)");
    SORTCHK(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 8), "hipMalloc(sort scratch)");
    SORTCHK(hipMemsetAsync(d_max, 0, 4, s), "memset");
    const unsigned nb = (unsigned)((n + SB - 1) / SB);
    hipLaunchKernelGGL(k_sort_init, dim3(nb), dim3(SB), 0, s, name_len, n, keys[0], vals[0], d_max);
    unsigned max_len = 0;
    SORTCHK(hipMemcpyAsync(&max_len, d_max, 4, hipMemcpyDeviceToHost, s), "memcpy");
    SORTCHK(hipStreamSynchronize(s), "k_sort_init");
    int cur = 0;
    unsigned len_bits = 1;
    while (len_bits < 32 && (max_len >> len_bits)) ++len_bits;
    SORTCHK(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys[cur], keys[cur ^ 1], vals[cur], vals[cur ^ 1], N, 0u, len_bits, s), "radix_sort_pairs(length)");
    cur ^= 1;
    for (int c = (int)((max_len + 7) / 8) - 1; c >= 0; --c) {
        hipLaunchKernelGGL(k_sort_keys, dim3(nb), dim3(SB), 0, s, data, gbase, name_off, name_len, vals[cur], n, c, keys[cur]);
        SORTCHK(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys[cur], keys[cur ^ 1], vals[cur], vals[cur ^ 1], N, 0u, 64u, s), "radix_sort_pairs(chunk)");
        cur ^= 1;
    }
    hipLaunchKernelGGL(k_sort_finish, dim3(nb), dim3(SB), 0, s, data, gbase, name_off, name_len, vals[cur], n, d_order, d_ndup);
    SORTCHK(hipGetLastError(), "k_sort_finish");
    SORTCHK(hipStreamSynchronize(s), "sort");
    cleanup();
    return 0;
}

}  // namespace fx
