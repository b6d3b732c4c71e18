// fx_bgzf_walk.hpp -- the member table of a BGZF file made ON THE DEVICE (gfx950, wave64).
//
// The reference reads a bgzip'd file through gzread and never needs to know where its members are (index.c:15-98).  The
// GPU inflate does: one wave per member.  Round 1-3 walked the member headers on the host -- BSIZE of one header says where
// the next one is: 47 k dependent reads of a mapped 1 GB file, 13-16 ms for C4, then 1 GB of mapping to give back, together
// as long as the decode kernel -- and only then began to stage.  Here the compressed bytes are staged first (they need no
// table) and the table is found in HBM:
//
//   k_bgzf_sig_count / _emit   every byte position is tested for the 16 bytes bgzip / htslib put in front of every member
//                              (1f 8b 08 04, MTIME 0, XFL 0, OS ff, XLEN 6, 'B' 'C' 2 0): one wave per 4 KiB, 64 positions
//                              per lane from registers (two 16-byte loads + funnel shifts), hits per granule, then in order
//   k_bgzf_member_rows         one thread per hit: BSIZE must lead exactly to the next hit (to the end of the file for the
//                              last one), ISIZE <= 65536 -- else the flag goes up and the HOST walk decides (a file with
//                              another header layout, or 16 bytes of payload that look like a header: the chain breaks)
//   + the scan of ISIZE (k_cnt_*) = where every member's bytes go
//
// Nothing is taken on trust: a table that does not tile the file exactly is not used, and every member is checked against
// the CRC-32 of its trailer after the inflate as before.
#pragma once
#include "fx_kernels.hpp"

namespace fx {

constexpr uint32_t BGZF_SIG0 = 0x04088b1fu, BGZF_SIG1 = 0x00000000u, BGZF_SIG2 = 0x0006ff00u, BGZF_SIG3 = 0x00024342u;
constexpr int BGZF_HDR = 18;                     // bytes of that header, BSIZE included

// bit k of the result: the signature begins at byte p0 + k (k < 16); w[0..7]: the 32 bytes at p0 (zeros past the end)
__device__ __forceinline__ uint32_t bgzf_sig_hits16(const uint32_t (&w)[8]) {
    uint32_t hits = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int j = k >> 2, sh = (k & 3) * 8;
        const uint32_t a = sh ? __funnelshift_r(w[j], w[j + 1], sh) : w[j];
        if (a != BGZF_SIG0) continue;
        const uint32_t b = sh ? __funnelshift_r(w[j + 1], w[j + 2], sh) : w[j + 1];
        const uint32_t c = sh ? __funnelshift_r(w[j + 2], w[j + 3], sh) : w[j + 2];
        const uint32_t d = sh ? __funnelshift_r(w[j + 3], w[j + 4], sh) : w[j + 3];
        if (b == BGZF_SIG1 && c == BGZF_SIG2 && d == BGZF_SIG3) hits |= 1u << k;
    }
    return hits;
}

// hits of the lane's 64 positions [g * 4096 + lane * 64, + 64); the buffer is readable (zeros) 48 bytes past n
__device__ __forceinline__ unsigned long long bgzf_lane_hits(const uint8_t *__restrict__ c, int64_t n, int64_t p0) {
    unsigned long long m = 0;
    if (p0 >= n) return 0;
    uint4 v[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) v[i] = (p0 + 16 * i < n + 32) ? *reinterpret_cast<const uint4 *>(c + p0 + 16 * i) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t w[8] = {v[i].x, v[i].y, v[i].z, v[i].w, v[i + 1].x, v[i + 1].y, v[i + 1].z, v[i + 1].w};
        m |= (unsigned long long)bgzf_sig_hits16(w) << (16 * i);
    }
    // a header must lie inside the file with its BSIZE field
    const int64_t room = n - BGZF_HDR - p0;                          // last position that can hold a header, relative to p0
    if (room < 63) m &= room < 0 ? 0ull : ((2ull << room) - 1ull);
    return m;
}

// one wave per 4 KiB granule: how many members begin in it.  The granules [g0, g0 + ngran) of the file (a file whose bytes are
// still arriving is searched group by group, bgzf_open_pipelined); cnt / off are indexed from 0
__global__ __launch_bounds__(BLOCK) void k_bgzf_sig_count(const uint8_t *__restrict__ c, int64_t n, int64_t g0, int64_t ngran, int32_t *__restrict__ cnt) {
    const int64_t g = (int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (g >= ngran) return;
    const unsigned long long m = bgzf_lane_hits(c, n, (g0 + g) * 4096 + (int64_t)lane_id() * 64);
    const uint32_t tot = wave_sum((uint32_t)__popcll(m));
    if (lane_id() == 0) cnt[g] = (int32_t)tot;
}

// ... and where, in file order: off[g] = members before granule g
__global__ __launch_bounds__(BLOCK) void k_bgzf_sig_emit(const uint8_t *__restrict__ c, int64_t n, int64_t g0, int64_t ngran, const int64_t *__restrict__ off,
                                                        int64_t *__restrict__ mstart) {
    const int64_t g = (int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (g >= ngran) return;
    if (off[g + 1] == off[g]) return;                                // (wave-uniform: most granules hold no header)
    const int64_t p0 = (g0 + g) * 4096 + (int64_t)lane_id() * 64;
    unsigned long long m = bgzf_lane_hits(c, n, p0);
    const uint32_t k = (uint32_t)__popcll(m);
    int64_t at = off[g] + (wave_incl_scan(k) - k);
    while (m) {
        const int b = __ffsll((long long)m) - 1;
        m &= m - 1;
        mstart[at++] = p0 + b;
    }
}

// one thread per member: BSIZE must lead to the next member exactly; the row of the member table.  mstart holds nstarts
// positions (>= nmem: a group of a file that is still arriving leaves its last member to the next group, which knows where
// it ends); first: where the first of them must begin (0, or where the chain of the group before led)
__global__ __launch_bounds__(BLOCK) void k_bgzf_member_rows(const uint8_t *__restrict__ c, int64_t n, const int64_t *__restrict__ mstart, int64_t nmem,
                                                           int64_t nstarts, int64_t first,
                                                           int64_t *__restrict__ coff, int32_t *__restrict__ clen, int32_t *__restrict__ isize,
                                                           int *__restrict__ bad, int32_t *__restrict__ clen_max) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nmem) return;
    const int64_t s = mstart[i], next = i + 1 < nstarts ? mstart[i + 1] : n;
    const int64_t msize = (int64_t)(c[s + 16] | (c[s + 17] << 8)) + 1;
    bool ok = s + msize == next && msize >= BGZF_HDR + 8 && (i > 0 || s == first);
    uint32_t isz = 0;
    if (ok) {
        const uint8_t *tr = c + next - 4;
        isz = tr[0] | (tr[1] << 8) | (tr[2] << 16) | ((uint32_t)tr[3] << 24);
        ok = isz <= 65536u;
    }
    coff[i] = s + BGZF_HDR;
    clen[i] = ok ? (int32_t)(msize - BGZF_HDR - 8) : 0;
    isize[i] = ok ? (int32_t)isz : 0;
    if (!ok) atomicOr(bad, 1);
    else atomicMax(clen_max, (int32_t)(msize - BGZF_HDR - 8));
}

// off[i] += base (the offsets of a group's members in the inflated stream: the scan of their ISIZE began at 0)
__global__ __launch_bounds__(BLOCK) void k_add_base(int64_t *__restrict__ off, int64_t n, int64_t base) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) off[i] += base;
}

}  // namespace fx
