// fx_fxi_dev.hpp -- the leaf pages of a `.fxi`'s two big b-trees formatted ON THE DEVICE (round 5).
//
// The reference fills `read` / `seq` with one sqlite3_step(INSERT) per record and lets CREATE UNIQUE INDEX sort the
// names (fastq.c:29-60, 136-171; index.c:178-207, 239-251, 363).  fx_fxi.hpp writes the same b-trees as pages from
// HOST arrays: the read table (44 B per read), the packed names and the sort order cross PCIe, and sixteen host threads
// format ~100 bytes per read -- for 10^8 reads that was 5 of the 6 seconds of Fastq(path), around 14 ms of kernels.
// The pages are a pure function of what already sits in HBM -- the record table (SoA), the names where they are in the
// resident stream, the sorted order of fx_sort.hip -- so they are made here, and only finished pages go to the host:
//
//   k_fxi_cell_sizes / k_fxi_entry_sizes   bytes of every table cell / index entry (+ its 2-byte cell pointer)
//   k_fxi_fill<IDX, EMIT>                  which rows go on which leaf: rows are cut into chunks of FXI_R, a chunk starts a
//                                          fresh page and is filled greedily by ONE lane walking the chunk's prefix sums
//                                          in LDS (the greedy fill of a whole table is a chain as long as the table; a
//                                          chain per chunk costs a half-empty page per 2048 rows, 1 % of the file); run
//                                          twice: pages per chunk, and -- after their prefix sum -- the first row of every page
//   k_fxi_table_leaves / k_fxi_index_leaves one wave per 4 KiB page: the page is built in LDS (one lane per cell: varints,
//                                          serial types, the name bytes from the stream, big-endian integers; cell
//                                          pointers; page header) and leaves as four coalesced 1 KiB stores
//   k_fxi_dividers                         the (name, rowid) entries that move up into the interior pages of the index
//
// Interior pages (1 % of the tree) are still written by the host from the first-row list (fx_fxi.hpp).  File format:
// https://www.sqlite.org/fileformat2.html 1.6 (b-tree pages), 2.1 (record format).  Only 4 KiB pages without reserved
// bytes (what SQLite creates by default) are written here; anything else keeps the host formatter.
#pragma once
#include "fx_kernels.hpp"

namespace fx {

constexpr int FXI_PAGE = 4096;
constexpr int FXI_R = 2048;                     // rows (entries) per fill chunk
constexpr int FXI_MAXCOL = 8;
constexpr int FXI_INDEX_MAX_LOCAL = ((FXI_PAGE - 12) * 64 / 255) - 23;     // larger index payloads spill to overflow pages

struct FxiCols {
    const void *p[FXI_MAXCOL];                  // the INTEGER columns behind the TEXT column, in schema order
    int w[FXI_MAXCOL];                          // bytes per element: 4 or 8
    int ncols;
    const int64_t *name_off;                    // name of row i = stream[name_off[i] + name_add - gbase, + name_len[i])
    int64_t name_add, gbase;
    const int32_t *name_len;
    int64_t row_base;                           // rowid of row i = row_base + i + 1 (a part of a table that several handles write)
    // the index only (round 6): offset (name_add included) and length of the e-th smallest name, written by the sort beside
    // its order -- entry sizes, leaves and dividers then read them in order and gather nothing but the name (before:
    // order[e] -> name_len[r], name_off[r] -> the name, three dependent gathers, 440 bytes fetched per 42-byte entry)
    const int64_t *s_off;
    const int32_t *s_len;
};

__device__ __forceinline__ int64_t fxi_col(const FxiCols &c, int k, int64_t i) {
    return c.w[k] == 8 ? reinterpret_cast<const int64_t *>(c.p[k])[i] : (int64_t) reinterpret_cast<const int32_t *>(c.p[k])[i];
}
// bytes of the SQLite varint of v (v < 2^56 everywhere here: payload sizes, rowids, serial types of short strings)
__device__ __forceinline__ int fxi_varint_len(uint64_t v) {
    const int bits = 64 - __clzll((long long)(v | 1));
    const int n = (bits + 6) / 7;
    return n > 9 ? 9 : n;
}
// serial type of an INTEGER (schema format 4) and the bytes of its body
__device__ __forceinline__ int fxi_int_serial(int64_t v, int *nb) {
    if (v == 0) { *nb = 0; return 8; }
    if (v == 1) { *nb = 0; return 9; }
    const uint64_t u = v < 0 ? ~(uint64_t)v : (uint64_t)v;
    if (u <= 127) { *nb = 1; return 1; }
    if (u <= 32767) { *nb = 2; return 2; }
    if (u <= 8388607) { *nb = 3; return 3; }
    if (u <= 2147483647ull) { *nb = 4; return 4; }
    if (u <= 140737488355327ull) { *nb = 6; return 5; }
    *nb = 8; return 6;
}
__device__ __forceinline__ uint8_t *fxi_put_varint(uint8_t *q, uint64_t v) {
    const int n = fxi_varint_len(v);                        // <= 8 here
    for (int k = 0; k < n; ++k) q[k] = (uint8_t)(((v >> (7 * (n - 1 - k))) & 0x7F) | (k < n - 1 ? 0x80 : 0));
    return q + n;
}
__device__ __forceinline__ uint8_t *fxi_put_be(uint8_t *q, uint64_t v, int n) {
    for (int k = 0; k < n; ++k) q[k] = (uint8_t)(v >> (8 * (n - 1 - k)));
    return q + n;
}
// the bytes of a name from the stream into a page under construction
__device__ __forceinline__ uint8_t *fxi_put_name(uint8_t *q, const uint8_t *__restrict__ src, int L) {
    int k = 0;
    for (; k + 16 <= L; k += 16) {
        const uint4 v = *reinterpret_cast<const uint4_u *>(src + k);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 16; ++j) q[k + j] = (uint8_t)(w[j >> 2] >> ((j & 3) * 8));
    }
    if (k < L && L >= 16) {                                  // the last 1..15 bytes: the 16 bytes that END with the name, once more
        const uint4 v = *reinterpret_cast<const uint4_u *>(src + L - 16);      // (one load; a loop of byte loads is a chain of
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};                             // up to fifteen round trips per lane)
#pragma unroll
        for (int j = 0; j < 16; ++j) q[L - 16 + j] = (uint8_t)(w[j >> 2] >> ((j & 3) * 8));
        return q + L;
    }
    for (; k < L; ++k) q[k] = src[k];
    return q + L;
}

// ---------------------------------------------------------------------------------------------- sizes
// table leaf cell of row i: varint(payload) varint(rowid) | header: size, NULL (the rowid alias), TEXT, k integers | name | integers
__global__ __launch_bounds__(BLOCK) void k_fxi_cell_sizes(FxiCols c, int64_t n, uint16_t *__restrict__ sz, int *__restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const int L = c.name_len[i] > 0 ? c.name_len[i] : 0;
    const int hdr = 2 + fxi_varint_len((uint64_t)(13 + 2 * (int64_t)L)) + c.ncols;
    int body = L;
#pragma unroll
    for (int k = 0; k < FXI_MAXCOL; ++k)
        if (k < c.ncols) { int nb; (void)fxi_int_serial(fxi_col(c, k, i), &nb); body += nb; }
    const int payload = hdr + body;
    if (hdr > 127 || payload > FXI_PAGE - 35 || L > 3900) { atomicOr(bad, 1); sz[i] = 64; return; }
    sz[i] = (uint16_t)(fxi_varint_len((uint64_t)payload) + fxi_varint_len((uint64_t)(c.row_base + i + 1)) + payload + 2);
}
// index leaf cell of the e-th smallest name: varint(payload) | header: size, TEXT, integer | name | rowid
__global__ __launch_bounds__(BLOCK) void k_fxi_entry_sizes(FxiCols c, const int64_t *__restrict__ order, int64_t n, uint16_t *__restrict__ sz,
                                                          int *__restrict__ bad) {
    const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (e >= n) return;
    const int64_t r = order[e];
    const int L = c.s_len ? c.s_len[e] : (c.name_len[r] > 0 ? c.name_len[r] : 0);
    int nb;
    (void)fxi_int_serial(c.row_base + r + 1, &nb);
    const int payload = 1 + fxi_varint_len((uint64_t)(13 + 2 * (int64_t)L)) + 1 + L + nb;
    if (payload > FXI_INDEX_MAX_LOCAL) { atomicOr(bad, 1); sz[e] = 64; return; }
    sz[e] = (uint16_t)(fxi_varint_len((uint64_t)payload) + payload + 2);
}

// ---------------------------------------------------------------------------------------------- which row on which page
// One wave per chunk of FXI_R items with sizes sz[] (cell + pointer).  Table (IDX = false): pages are filled one after the
// other.  Index (IDX = true): page, divider, page, ..., page -- the entry behind a full page moves up into the parent
// (fx_fxi.hpp: fill_level) -- and the chunk's LAST entry is the divider between this chunk and the next, except in the
// last chunk, which ends with a page; an entry is never left alone behind a divider at the chunk's end (the page before
// gives one up: a page that is full holds at least four entries, FXI_INDEX_MAX_LOCAL).
// EMIT = false: pages[c] = number of pages of chunk c.  EMIT = true: first[pbase[c] + k] = first item of the chunk's k-th
// page; the last chunk also writes first[total pages] = n.
template <bool IDX, bool EMIT>
__global__ __launch_bounds__(64) void k_fxi_fill(const uint16_t *__restrict__ sz, int64_t n, int room, int32_t *__restrict__ pages,
                                                 const int64_t *__restrict__ pbase, int64_t *__restrict__ first) {
    __shared__ uint32_t P[FXI_R + 1];                       // P[j] = bytes of the chunk's first j items
    const int lane = threadIdx.x;
    const int64_t c = blockIdx.x, i0 = c * FXI_R;
    const int m = (int)(n - i0 < FXI_R ? n - i0 : FXI_R);
    const bool last = i0 + m >= n;
    constexpr int PER = FXI_R / 64;                          // 32 consecutive items per lane
    uint32_t v[PER], s = 0;
    if (m == FXI_R) {
        const uint4 *src = reinterpret_cast<const uint4 *>(sz + i0 + lane * PER);
#pragma unroll
        for (int q = 0; q < PER / 8; ++q) {
            const uint4 x = src[q];
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[q * 8 + 2 * k] = w[k] & 0xFFFFu; v[q * 8 + 2 * k + 1] = w[k] >> 16; }
        }
    } else {
#pragma unroll
        for (int k = 0; k < PER; ++k) { const int j = lane * PER + k; v[k] = j < m ? sz[i0 + j] : 0u; }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) s += v[k];
    uint32_t run = wave_incl_scan(s) - s;
#pragma unroll
    for (int k = 0; k < PER; ++k) { P[lane * PER + k] = run; run += v[k]; }
    if (lane == 63) P[FXI_R] = run;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane != 0) return;
    const int mm = IDX ? (last ? m : m - 1) : m;            // items that go on pages or between them inside the chunk
    const uint32_t avg = P[m] / (uint32_t)m;
    const int est = (int)((uint32_t)room / (avg ? avg : 1u));
    const int64_t ob = EMIT ? pbase[c] : 0;
    int pos = 0, np = 0;
    while (pos < mm) {
        if (EMIT) first[ob + np] = i0 + pos;
        ++np;
        const uint32_t lim = P[pos] + (uint32_t)room;
        int j = pos + est;                                   // largest j <= mm with P[j] - P[pos] <= room; sizes are nearly uniform
        if (j > mm) j = mm;
        if (j <= pos) j = pos + 1;
        while (j < mm && P[j + 1] <= lim) ++j;
        while (P[j] > lim) --j;
        if (IDX) {
            if (j >= mm) pos = mm;
            else { if (j == mm - 1) --j; pos = j + 1; }      // entry j is the divider
        } else
            pos = j;
    }
    if (!EMIT) pages[c] = np;
    else if (last) first[ob + np] = n;
}

// ---------------------------------------------------------------------------------------------- pages
__device__ __forceinline__ void fxi_zero_page(uint8_t *pg, int lane) {
    const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < FXI_PAGE / 1024; ++j) reinterpret_cast<uint4 *>(pg)[lane + 64 * j] = z;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void fxi_store_page(const uint8_t *pg, uint8_t *__restrict__ dst, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    typedef uint32_t fxi_v4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < FXI_PAGE / 1024; ++j) {              // four coalesced 1 KiB stores, past the caches (the page is not read again here)
        const fxi_v4 v = reinterpret_cast<const fxi_v4 *>(pg)[lane + 64 * j];
        __builtin_nontemporal_store(v, reinterpret_cast<fxi_v4 *>(dst) + lane + 64 * j);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// table leaves [k0, k1) of the tree -> out + (k - k0) * FXI_PAGE; leaf k holds rows [first[k], first[k + 1])
__global__ __launch_bounds__(BLOCK) void k_fxi_table_leaves(FxiCols c, const uint8_t *__restrict__ data, const int64_t *__restrict__ first,
                                                           int64_t k0, int64_t k1, uint8_t *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[BLOCK / 64][FXI_PAGE];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    uint8_t *const pg = lds[w];
    const int64_t nw = (int64_t)gridDim.x * (BLOCK / 64);
    for (int64_t k = k0 + (int64_t)blockIdx.x * (BLOCK / 64) + w; k < k1; k += nw) {
        fxi_zero_page(pg, lane);
        const int64_t a = first[k], b = first[k + 1];
        uint32_t top = FXI_PAGE;
        for (int64_t r0 = a; r0 < b; r0 += 64) {
            const int64_t i = r0 + lane;
            const bool valid = i < b;
            int L = 0, tl = 0, hdr = 0, body = 0, nb[FXI_MAXCOL], st[FXI_MAXCOL];
            int64_t val[FXI_MAXCOL];
            uint32_t len = 0;
            if (valid) {
                L = c.name_len[i] > 0 ? c.name_len[i] : 0;
                tl = fxi_varint_len((uint64_t)(13 + 2 * L));
                hdr = 2 + tl + c.ncols;
                body = L;
#pragma unroll
                for (int q = 0; q < FXI_MAXCOL; ++q)
                    if (q < c.ncols) { val[q] = fxi_col(c, q, i); st[q] = fxi_int_serial(val[q], &nb[q]); body += nb[q]; }
                len = (uint32_t)(fxi_varint_len((uint64_t)(hdr + body)) + fxi_varint_len((uint64_t)(c.row_base + i + 1)) + hdr + body);
            }
            const uint32_t incl = wave_incl_scan(len);
            if (valid) {
                const uint32_t at = top - incl;
                uint8_t *q = pg + at;
                q = fxi_put_varint(q, (uint64_t)(hdr + body));
                q = fxi_put_varint(q, (uint64_t)(c.row_base + i + 1));
                *q++ = (uint8_t)hdr;
                *q++ = 0;                                    // INTEGER PRIMARY KEY: NULL, the rowid is the value
                q = fxi_put_varint(q, (uint64_t)(13 + 2 * L));
#pragma unroll
                for (int x = 0; x < FXI_MAXCOL; ++x) if (x < c.ncols) *q++ = (uint8_t)st[x];
                q = fxi_put_name(q, data + (c.name_off[i] + c.name_add - c.gbase), L);
#pragma unroll
                for (int x = 0; x < FXI_MAXCOL; ++x) if (x < c.ncols) q = fxi_put_be(q, (uint64_t)val[x], nb[x]);
                const uint32_t slot = 8 + 2 * (uint32_t)(i - a);
                pg[slot] = (uint8_t)(at >> 8); pg[slot + 1] = (uint8_t)at;
            }
            top -= (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (lane == 0) {
            const uint32_t cnt = (uint32_t)(b - a);
            pg[0] = 0x0D;
            pg[3] = (uint8_t)(cnt >> 8); pg[4] = (uint8_t)cnt;
            pg[5] = (uint8_t)(top >> 8); pg[6] = (uint8_t)top;
        }
        fxi_store_page(pg, out + (k - k0) * FXI_PAGE, lane);
    }
}

// index leaves [k0, k1): leaf k holds the entries [first[k], first[k + 1] - 1) of the sorted order -- the entry in front of
// the next leaf's first is the divider that went up -- and the last leaf of the tree [first[k], n)
__global__ __launch_bounds__(BLOCK) void k_fxi_index_leaves(FxiCols c, const uint8_t *__restrict__ data, const int64_t *__restrict__ order,
                                                           const int64_t *__restrict__ first, int64_t nleaf, int64_t n, int64_t k0, int64_t k1,
                                                           uint8_t *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[BLOCK / 64][FXI_PAGE];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    uint8_t *const pg = lds[w];
    const int64_t nw = (int64_t)gridDim.x * (BLOCK / 64);
    for (int64_t k = k0 + (int64_t)blockIdx.x * (BLOCK / 64) + w; k < k1; k += nw) {
        fxi_zero_page(pg, lane);
        const int64_t a = first[k], b = k + 1 < nleaf ? first[k + 1] - 1 : n;
        uint32_t top = FXI_PAGE;
        for (int64_t e0 = a; e0 < b; e0 += 64) {
            const int64_t e = e0 + lane;
            const bool valid = e < b;
            int64_t r = 0;
            int L = 0, tl = 0, nb = 0, st = 0, payload = 0;
            uint32_t len = 0;
            if (valid) {
                r = order[e];
                L = c.s_len ? c.s_len[e] : (c.name_len[r] > 0 ? c.name_len[r] : 0);
                tl = fxi_varint_len((uint64_t)(13 + 2 * L));
                st = fxi_int_serial(c.row_base + r + 1, &nb);
                payload = 1 + tl + 1 + L + nb;
                len = (uint32_t)(fxi_varint_len((uint64_t)payload) + payload);
            }
            const uint32_t incl = wave_incl_scan(len);
            if (valid) {
                const uint32_t at = top - incl;
                uint8_t *q = pg + at;
                q = fxi_put_varint(q, (uint64_t)payload);
                *q++ = (uint8_t)(1 + tl + 1);
                q = fxi_put_varint(q, (uint64_t)(13 + 2 * L));
                *q++ = (uint8_t)st;
                q = fxi_put_name(q, data + ((c.s_off ? c.s_off[e] : c.name_off[r] + c.name_add) - c.gbase), L);
                q = fxi_put_be(q, (uint64_t)(c.row_base + r + 1), nb);
                const uint32_t slot = 8 + 2 * (uint32_t)(e - a);
                pg[slot] = (uint8_t)(at >> 8); pg[slot + 1] = (uint8_t)at;
            }
            top -= (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (lane == 0) {
            const uint32_t cnt = (uint32_t)(b - a);
            pg[0] = 0x0A;
            pg[3] = (uint8_t)(cnt >> 8); pg[4] = (uint8_t)cnt;
            pg[5] = (uint8_t)(top >> 8); pg[6] = (uint8_t)top;
        }
        fxi_store_page(pg, out + (k - k0) * FXI_PAGE, lane);
    }
}

// the entries that went up: divider d sits between leaf d and leaf d + 1 -> its row and the length of its name (two passes:
// lengths, then -- with their prefix sum -- the bytes)
__global__ __launch_bounds__(BLOCK) void k_fxi_divider_rows(FxiCols c, const int64_t *__restrict__ order, const int64_t *__restrict__ first,
                                                           int64_t nd, int64_t *__restrict__ row, int32_t *__restrict__ len) {
    const int64_t d = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (d >= nd) return;
    const int64_t e = first[d + 1] - 1, r = order[e];
    row[d] = r;
    len[d] = c.s_len ? c.s_len[e] : (c.name_len[r] > 0 ? c.name_len[r] : 0);
}
__global__ __launch_bounds__(BLOCK) void k_fxi_divider_names(FxiCols c, const uint8_t *__restrict__ data, const int64_t *__restrict__ row,
                                                            const int64_t *__restrict__ off, int64_t nd, uint8_t *__restrict__ out) {
    const int64_t d = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (d >= nd) return;
    const int64_t r = row[d];
    const uint8_t *src = data + (c.name_off[r] + c.name_add - c.gbase);
    const int64_t o = off[d], L = off[d + 1] - o;
    for (int64_t k = 0; k < L; ++k) out[o + k] = src[k];
}

}  // namespace fx
