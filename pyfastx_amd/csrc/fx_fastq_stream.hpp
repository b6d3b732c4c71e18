// fx_fastq_stream.hpp -- FASTQ composition as a STREAM over the bytes (gfx950, round 4).
//
// pyfastx_fastq_calc_composition (fastq.c:663-795) is a `line_num % 4` loop over the lines of the file: the bases of every
// second line of four are counted (A C G T upper case; everything else but '\r' is N), the bytes of every fourth line give the
// smallest and largest quality.  k_fastq_comp (fx_fastq.hpp) goes at it from the READ TABLE: ten lanes per read, 16 bytes per
// lane at whatever address the table says -- and is bound by exactly that: unaligned 16-byte gathers cost 40 L1 tag look-ups
// per instruction (7.2 ms for C3, 4.8 TB/s of a stream that is read whole anyway).
//
// This kernel reads the stream the way the scans do -- a wave per run of granules, 64 lanes x 16 contiguous bytes per load,
// non-temporal -- and decides per BYTE which line of four it belongs to from what the index build left behind: nl_prefix[g] is
// the global number of the first newline of granule g, so the line a byte belongs to is nl_prefix[g] + (newlines of the
// granule in front of it), a wave scan of the popcounts of the newline masks.  A 16-byte chunk then has ONE part that counts
// (its bytes in front of its newline, or behind it) of ONE kind (bases or qualities), picked by a mask looked up in LDS:
//   bases      x & 7 is distinct for ' ' A \n C T \r N G, so one v_perm_b32 makes a one-hot class byte of every byte and a second
//              one the byte that class stands for (x ^ expected != 0: a byte that is none of them -- lower case, IUPAC codes --
//              fixed up byte by byte: N for the reference); the one-hot words go into bit planes (fx_comp.hpp);
//   qualities  packed 16-bit min / max on the bytes and on the bytes shifted by one (no unpacking: the high byte of a half
//              decides), bytes that do not count filled with 0x00 / 0xFF.
// Chunks with three newlines or more (lines shorter than ~7 bytes) are walked byte by byte.  A quality byte below '!' or above
// 127 -- a '\r', which the reference skips with its line.l quirk (fastq.c:733-737), or bytes it reads as negative chars -- only
// raises a flag: the host then runs k_fastq_comp, which knows those cases, instead.  Whole streams only (a shard counts the
// reads it OWNS: k_fastq_comp, from its table).
#pragma once
#include "fx_fastq.hpp"
#include "fx_comp.hpp"

namespace fx {

// by x & 7:                     0 (' ')  1 A     2 \n    3 C   |  4 T     5 \r    6 N     7 G
constexpr uint32_t FS_OH_LO = 0x02000100u, FS_OH_HI = 0x04100008u;     // one-hot: A 1, C 2, G 4, T 8, N 16
constexpr uint32_t FS_EX_LO = 0x430A4180u, FS_EX_HI = 0x474E0D54u;     // the byte the code stands for (0x80: none)
constexpr int FS_GPW = 8;                                              // granules per wave
// masks of a chunk's bytes, s_mask[(r * 17 + k1) * 17 + k2] with k1 <= k2 the positions of its first two newlines (16: none):
// r = 0 the bytes in front of k1, r = 1 those between k1 and k2, r = 2 those behind k2, r = 3 none
constexpr int FS_NMASK = 4 * 17 * 17;

__device__ __forceinline__ uint32_t fs_orn(uint32_t a, uint32_t b) { return a | ~b; }

struct FsAcc { CompPlanes pl; uint32_t mn, mx; uint32_t extra[5]; bool qodd, pend; };   // mn / mx: high bytes of the 16-bit halves; pend: lane 63 holds a '\r' as the granule's last byte

// the bytes of `x` (one word of a chunk) that are bases (ms) into the one-hot word h; -> x ^ expected where a base is none of
// A C G T N \r (0: none).  Qualities (mq): min and max are not computed but TESTED -- is any quality byte above the largest or
// below the smallest the wave has met (qf: cmin in every byte, k1 = 0x7F - cmax, k2 = 0x80 - cmin per byte)?  A byte above
// cmax sets bit 7 of x + k1, one of cmin or more sets bit 7 of x + k2, one of 128 or more has it set itself; bytes that are no
// qualities are replaced by cmin, which passes both.  The bounds settle within the first granules of a wave; a row that fails
// the test is measured exactly (fs_row_minmax) and the bounds move.
__device__ __forceinline__ uint32_t fs_word(uint32_t x, uint32_t ms, uint32_t mq, uint32_t &h, uint32_t qf, uint32_t k1, uint32_t k2,
                                            uint32_t &over, uint32_t &notunder) {
    const uint32_t xs = (uint32_t)__builtin_amdgcn_bitop3_b32(ms, x, 0x0D0D0D0Du, 0xCA);     // ms ? x : '\r'
    const uint32_t code = xs & 0x07070707u;
    h = __builtin_amdgcn_perm(FS_OH_HI, FS_OH_LO, code);
    const uint32_t d = xs ^ __builtin_amdgcn_perm(FS_EX_HI, FS_EX_LO, code);
    const uint32_t xq = (uint32_t)__builtin_amdgcn_bitop3_b32(mq, x, qf, 0xCA);              // mq ? x : cmin
    over = (uint32_t)__builtin_amdgcn_bitop3_b32(over, xq + k1, xq, 0xFE);                   // over | (xq + k1) | xq
    notunder &= xq + k2;
    return d;
}
// smallest / largest quality byte of a row's chunk (mask mq), exactly: packed 16-bit min / max on the bytes and on the bytes
// shifted by one (the high byte of a half decides); bytes that are no qualities count as 0xFF / 0x00
__device__ __forceinline__ void fs_minmax(const uint4 &v, const uint4 &mq, uint32_t &mn, uint32_t &mx) {
    const uint32_t xw[4] = {v.x, v.y, v.z, v.w}, mw[4] = {mq.x, mq.y, mq.z, mq.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t hi = xw[i] & mw[i], lo = fs_orn(xw[i], mw[i]);
        mx = pk_max_u16(pk_max_u16(mx, hi), hi << 8);
        mn = pk_min_u16(pk_min_u16(mn, lo), (lo << 8) | 0xFFu);
    }
}

// byte b (0..15, any value at run time) of a chunk held in four words: selects, not an indexed array (which would live in scratch)
__device__ __forceinline__ uint32_t fs_byte(const uint4 &v, int b) {
    const uint32_t lo = (b & 4) ? v.y : v.x, hi = (b & 4) ? v.w : v.z;
    return (((b & 8) ? hi : lo) >> ((b & 3) * 8)) & 0xFFu;
}
// one byte of line-of-four p as the reference's loop takes it (fastq.c:720-745)
__device__ __forceinline__ void fs_one(uint32_t c, uint32_t p, FsAcc &a) {
    if (p == 1u) {
        if (c == 13u) return;
        const int cls = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;
#pragma unroll
        for (int k = 0; k < 5; ++k) a.extra[k] += cls == k ? 1u : 0u;
    } else if (p == 3u) {
        if (c < 33u || c > 127u) a.qodd = true;
        else { a.mn = pk_min_u16(a.mn, (c << 8) | 0xFFFF00FFu); a.mx = pk_max_u16(a.mx, c << 8); }
    }
}

// ---- '\r' in front of a line's '\n' (CRLF files, round 5).  The reference's loop over a quality line (fastq.c:731-745) meets
// the '\r' as the LAST byte of line.s: `--line.l; continue;` ends the loop -- the byte is skipped, nothing else happens.  A '\r'
// anywhere else in a quality line makes that loop stop short of the line's end (k_fastq_qual_walk knows how); such a file is
// left to the table kernels.  So: a '\r' that is followed by '\n' is taken out of the quality mask of its chunk, and a '\r'
// that is NOT followed by '\n' -- anywhere, which is more than the rule asks for and costs nothing -- raises the run's odd
// flag.  Rows without a '\r' (every row of an LF file: the test is wave-uniform and was made by fq_cr_mask already) pay one
// scalar branch.  crb: 0xFF in the bytes of v that are '\r'.
__device__ __forceinline__ uint4 fs_cr_bytes(const uint4 &v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const uint32_t z = zero_bytes(w[i] ^ 0x0D0D0D0Du) >> 7; o[i] = (z << 8) - z; }     // 0x01 -> 0xFF per byte
    return make_uint4(o[0], o[1], o[2], o[3]);
}
// the exact map of the '\r' bytes of a row as fq_cr_mask gives it, and whether the row holds one at all (wave-uniform)
__device__ __forceinline__ uint16_t fq_cr_mask2(const uint4 &v, bool &row_has) {
    const uint32_t y0 = v.x ^ 0x0D0D0D0Du, y1 = v.y ^ 0x0D0D0D0Du, y2 = v.z ^ 0x0D0D0D0Du, y3 = v.w ^ 0x0D0D0D0Du;
    uint32_t any_cr = (y0 - 0x01010101u) & ~y0;
    any_cr = (uint32_t)__builtin_amdgcn_bitop3_b32(any_cr, y1 - 0x01010101u, y1, 0xF4);      // a | (b & ~c)
    any_cr = (uint32_t)__builtin_amdgcn_bitop3_b32(any_cr, y2 - 0x01010101u, y2, 0xF4);
    any_cr = (uint32_t)__builtin_amdgcn_bitop3_b32(any_cr, y3 - 0x01010101u, y3, 0xF4);
    row_has = __ballot((any_cr & 0x80808080u) != 0) != 0ull;
    return row_has ? (uint16_t)eq_mask16(v, 0x0D0D0D0Du) : (uint16_t)0;
}

struct FsQ { uint32_t qf, qk1, qk2; int cmin, cmax; };       // the wave's quality bounds so far and the constants of the test made from them
__device__ __forceinline__ void fsq_init(FsQ &q) {
    q.cmin = 104; q.cmax = 33;                               // fastq.c:667-668
    q.qf = 33u * 0x01010101u; q.qk1 = (0x7Fu - 33u) * 0x01010101u; q.qk2 = (0x80u - 104u) * 0x01010101u;   // fill, "above cmax", "at least cmin"
}

// One row (1 KiB: a 16-byte chunk per lane) of a granule through the counters.  pA: which line of four the chunk's first byte
// belongs to; nlm / crm: the chunk's newline and '\r' maps; row_cr: some lane of the row holds a '\r'; next_first: 1 / 0 = the
// chunk behind lane 63's begins / does not begin with '\n', 2 = not known here (the caller settles a.pend).
// CRLF = false: the form for streams without '\r\n' line ends (what the build's sample of the stream says: nearly every file) --
// no '\r' logic at all, a '\r' among the quality bytes is below '!' and raises the odd flag like any such byte (the three
// instructions per row and the registers the other form costs were 3 % of k_fastq_lines_comp: 10.74 -> 11.10 ms for C3).
template <int J, bool CRLF>
__device__ __forceinline__ void fs_row(const uint4 &vj, uint32_t nlmj, uint32_t crmj, bool row_cr, uint32_t nlm_next, uint32_t pA,
                                       const uint4 *s_mask, FsAcc &a, CompCarry &cy, FsQ &qs, int *fix, int lane) {
    // Up to two line ends in a chunk ('+' lines are two bytes long: nearly every record has such a chunk): three stretches --
    // in front of the first newline (line pA), between the two (pA + 1), behind the second (pA + 2) -- of which at most one
    // holds bases (its line is 1 of four: stretch (1 - pA) & 3) and at most one qualities (line 3: stretch (3 - pA) & 3);
    // stretch 3 is none.  ONE look-up each, the address from pA and the two positions.
    const uint32_t cnt = (uint32_t)__popc(nlmj);
    const uint32_t m1 = nlmj & (nlmj - 1u);
    const int k1 = cnt ? __ffs(nlmj) - 1 : 16, k2 = m1 ? __ffs(m1) - 1 : 16;
    const bool slow = cnt >= 3u;                                       // three line ends in 16 bytes: byte by byte below
    const int kx = k1 * 17 + k2;
    const uint32_t rs = slow ? 3u : (1u - pA) & 3u, rq = slow ? 3u : (3u - pA) & 3u;
    const uint4 ms = s_mask[rs * 289u + kx];
    uint4 mq = s_mask[rq * 289u + kx];
    if (CRLF && row_cr) {
        const uint4 crb = fs_cr_bytes(vj);
        mq.x &= ~crb.x; mq.y &= ~crb.y; mq.z &= ~crb.z; mq.w &= ~crb.w;
        uint32_t lone = crmj & ~(nlmj >> 1) & 0x7FFFu;                 // a '\r' with something else than '\n' behind it
        const uint32_t nb = (uint32_t)__shfl_down((int)(nlmj & 1u), 1, 64);
        // the chunk behind lane 63's: lane 0 of the next row (nlm_next), or -- last row, J == 3 -- not known here (the caller settles a.pend)
        const uint32_t next_first = J < 3 ? (uint32_t)__builtin_amdgcn_readlane((int)nlm_next, 0) & 1u : 2u;
        if (crmj & 0x8000u) {                                          // the chunk's last byte: the next chunk's first byte decides
            const uint32_t nf = lane < 63 ? nb : next_first;
            if (nf == 0u) lone |= 0x8000u;
            else if (nf == 2u) a.pend = true;
        }
        if (lone) a.qodd = true;
    }
    uint32_t h0, h1, h2, h3, over = 0, notunder = 0xFFFFFFFFu;
    const uint32_t d0 = fs_word(vj.x, ms.x, mq.x, h0, qs.qf, qs.qk1, qs.qk2, over, notunder), d1 = fs_word(vj.y, ms.y, mq.y, h1, qs.qf, qs.qk1, qs.qk2, over, notunder);
    const uint32_t d2 = fs_word(vj.z, ms.z, mq.z, h2, qs.qf, qs.qk1, qs.qk2, over, notunder), d3 = fs_word(vj.w, ms.w, mq.w, h3, qs.qf, qs.qk1, qs.qk2, over, notunder);
    planes_add4(a.pl, cy, J, h0, h1, h2, h3);
    if (__builtin_expect(__ballot((((over | ~notunder) & 0x80808080u) != 0)) != 0ull, 0)) {
        // a quality outside the bounds: this row exactly, the wave's bounds move (wave-uniform again)
        uint32_t mn = 0xFFFFFFFFu, mx = 0u;
        fs_minmax(vj, mq, mn, mx);
        int lo = (int)min(mn >> 24, (mn >> 8) & 0xFFu), hi = (int)max(mx >> 24, (mx >> 8) & 0xFFu);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const int x = __shfl_xor(lo, d, 64), y = __shfl_xor(hi, d, 64);
            lo = x < lo ? x : lo; hi = y > hi ? y : hi;
        }
        if (lo <= hi) {                                                // the row held qualities
            qs.cmin = lo < qs.cmin ? lo : qs.cmin; qs.cmax = hi > qs.cmax ? hi : qs.cmax;
            if (lo < 33 || hi > 127) a.qodd = true;
            const int fl = qs.cmin <= qs.cmax ? qs.cmin : qs.cmax;     // (no quality met yet: anything fails, the row above is exact anyway)
            qs.qf = (uint32_t)fl * 0x01010101u;
            qs.qk1 = (uint32_t)(0x7F - (qs.cmax < 127 ? qs.cmax : 127)) * 0x01010101u;
            qs.qk2 = (uint32_t)(0x80 - (qs.cmin > 0 ? qs.cmin : 0)) * 0x01010101u;
        }
    }
    if (__builtin_expect((d0 | d1 | d2 | d3) != 0, 0)) {   // a base that is none of A C G T N \r: N for the reference; take the aliased class back out
        // (word by word, the bytes of a word by shifting it: a byte picked by a run-time index makes the compiler keep the
        // chunk in scratch memory -- 64 bytes per lane stored for every granule, on the straight path)
        const uint32_t dsw[4] = {d0, d1, d2, d3}, hsw[4] = {h0, h1, h2, h3};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t dw = dsw[q], hw = hsw[q];
#pragma unroll 1
            for (int k = 0; k < 4 && dw; ++k, dw >>= 8, hw >>= 8) {
                if (!(dw & 0xFFu)) continue;
                const uint32_t hk = hw & 0xFFu;
                if (hk) atomicSub(&fix[__ffs(hk) - 1], 1);
                atomicAdd(&fix[4], 1);
            }
        }
    }
    if (__builtin_expect(slow, 0)) {
        uint32_t p = pA;
        bool cr_open = false;                                          // a '\r' of a quality line whose next byte has not been seen
        const uint32_t vsw[4] = {vj.x, vj.y, vj.z, vj.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t w = vsw[q];
#pragma unroll 1
            for (int k = 0; k < 4; ++k, w >>= 8) {
                const uint32_t cc = w & 0xFFu;
                if (cr_open && cc != 10u) a.qodd = true;
                cr_open = false;
                if (cc == 10u) { p = (p + 1u) & 3u; continue; }
                if (cc == 13u && p == 3u) { cr_open = true; continue; }
                fs_one(cc, p, a);
            }
        }
        if (cr_open) a.qodd = true;                                    // (the chunk ends on it: left to the table kernels)
    }
}
// the ragged end of the stream and other byte-by-byte walks: one byte of line-of-four ph with the byte behind it (10 at the end of the stream)
__device__ __forceinline__ void fs_one_cr(uint32_t b, uint32_t next, uint32_t ph, FsAcc &a) {
    if (b == 13u && ph == 3u) { if (next != 10u) a.qodd = true; return; }
    fs_one(b, ph, a);
}

// ngran_full granules that lie entirely inside the stream go through the fast path; the ragged end of the stream ([tail0, n),
// less than a granule) is walked by the last wave, 64 bytes per lane.  With a LIST (round 5: the runs of k_fastq_lines_comp whose
// guess was wrong or missing, FQLC_G granules each) only those runs are counted -- the line-of-four of every chunk from the
// prefixes, which exist by then -- and the ragged end is somebody else's.
__global__ __launch_bounds__(BLOCK) void k_fastq_comp_stream(const uint8_t *__restrict__ data, int64_t n, int64_t nfull,
                                                            const int64_t *__restrict__ nl_prefix, int64_t line0, FastqAcc *acc,
                                                            const uint32_t *__restrict__ list, int64_t nlist, int list_gpr) {
    __shared__ uint4 s_mask[FS_NMASK];
    __shared__ int s_fix[BLOCK / 64][8];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < FS_NMASK; e += BLOCK) {
        const int r = e / 289, k1 = (e % 289) / 17, k2 = e % 17;
        const int lo = r == 0 ? -1 : r == 1 ? k1 : r == 2 ? k2 : 16, hi = r == 0 ? k1 : r == 1 ? k2 : 16;     // the bytes lo < p < hi
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t m = 0;
            for (int b = 0; b < 4; ++b) { const int p = 4 * i + b; if (p > lo && p < hi) m |= 0xFFu << (8 * b); }
            w[i] = m;
        }
        s_mask[e] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (lane < 8) s_fix[wv][lane] = 0;
    __syncthreads();
    int *fix = s_fix[wv];
    FsAcc a;
#pragma unroll
    for (int k = 0; k < COMP_NPL; ++k) a.pl.p[k] = 0;
    a.mn = 0xFFFFFFFFu; a.mx = 0u; a.qodd = false; a.pend = false;
    FsQ qs;
    fsq_init(qs);                                            // what the wave has met so far (wave-uniform)
#pragma unroll
    for (int c = 0; c < 5; ++c) a.extra[c] = 0;
    // A grid of as many waves as the device holds at once; wave w takes the runs w, w + nwaves, ... (one set of atomics per
    // WAVE at the end: a million waves adding to the same seven words would take longer than the stream takes to read).
    const int64_t wave = (int64_t)blockIdx.x * (BLOCK / 64) + wv, nwaves = (int64_t)gridDim.x * (BLOCK / 64);
    unsigned long long tot[5] = {0, 0, 0, 0, 0};
    int runs_in_planes = 0;
    uint4 v[GR_ROWS], nx[GR_ROWS];
    const int per = list ? list_gpr / FS_GPW : 1;            // pieces of FS_GPW granules per listed run
    const int64_t npieces = list ? nlist * per : (nfull + FS_GPW - 1) / FS_GPW;
  for (int64_t pc = wave; pc < npieces; pc += nwaves) {
    const int64_t g0 = list ? (int64_t)list[pc / per] * list_gpr + (pc % per) * FS_GPW : pc * FS_GPW;
    if (g0 >= nfull) continue;
    granule_load<true>(v, data, n, 0, g0);
#pragma unroll 1
    for (int kk = 0; kk < FS_GPW; ++kk) {
        const int64_t g = g0 + kk;
        if (g >= nfull) break;
        const bool more = kk + 1 < FS_GPW && g + 1 < nfull;
        if (more) granule_load<true>(nx, data, n, 0, g + 1);
        // ---- the line every chunk begins in
        uint32_t nlm[GR_ROWS], ex[GR_ROWS], crm[GR_ROWS];
        bool rcr[GR_ROWS];
        uint32_t run = 0;
#pragma unroll
        for (int j = 0; j < GR_ROWS; ++j) {
            nlm[j] = eq_mask16(v[j], 0x0A0A0A0Au);
            crm[j] = fq_cr_mask2(v[j], rcr[j]);
            const uint32_t c = (uint32_t)__popc(nlm[j]);
            const uint32_t inc = wave_incl_scan(c);
            ex[j] = run + inc - c;
            run += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        const uint32_t L0 = (uint32_t)((line0 + nl_prefix[g]) & 3);
        CompCarry cy;
        fs_row<0, true>(v[0], nlm[0], crm[0], rcr[0], nlm[1], (L0 + ex[0]) & 3u, s_mask, a, cy, qs, fix, lane);
        fs_row<1, true>(v[1], nlm[1], crm[1], rcr[1], nlm[2], (L0 + ex[1]) & 3u, s_mask, a, cy, qs, fix, lane);
        fs_row<2, true>(v[2], nlm[2], crm[2], rcr[2], nlm[3], (L0 + ex[2]) & 3u, s_mask, a, cy, qs, fix, lane);
        fs_row<3, true>(v[3], nlm[3], crm[3], rcr[3], 0u, (L0 + ex[3]) & 3u, s_mask, a, cy, qs, fix, lane);
        static_assert(GR_ROWS == 4, "four rows per granule");
        planes_finish16(a.pl, cy);
        if (__ballot(a.pend)) {                              // a '\r' as the granule's last byte: the byte behind it (the end of the stream passes)
            const int64_t q = (g + 1) * (int64_t)GRAN;
            if (a.pend && q < n && data[q] != 10) a.qodd = true;
            a.pend = false;
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < GR_ROWS; ++j) v[j] = nx[j];
        }
    }
    if (++runs_in_planes == 6) {                           // 6 x 8 granules x 16 words: the planes count to 1023
#pragma unroll
        for (int c = 0; c < 5; ++c) tot[c] += planes_count(a.pl, c);
#pragma unroll
        for (int k = 0; k < COMP_NPL; ++k) a.pl.p[k] = 0;
        runs_in_planes = 0;
    }
  }
    // ---- the ragged end of the stream: wave 0 walks its bytes, 64 per lane
    const int64_t tail0 = nfull * (int64_t)GRAN;
    if (!list && tail0 < n && wave == 0) {
        const int64_t lo = tail0 + (int64_t)lane * 64, hi = lo + 64 < n ? lo + 64 : n;
        uint32_t c = 0;
        for (int64_t p = lo; p < hi; ++p) c += data[p] == 10;
        const uint32_t inc = wave_incl_scan(c);
        uint32_t ph = (uint32_t)((line0 + nl_prefix[nfull] + (int64_t)(inc - c)) & 3);
        for (int64_t p = lo; p < hi; ++p) {
            const uint32_t b = data[p];
            if (b == 10u) { ph = (ph + 1u) & 3u; continue; }
            fs_one_cr(b, p + 1 < n ? data[p + 1] : 10u, ph, a);
        }
    }
    // ---- the lane's counts, the wave's, the accumulators
#pragma unroll
    for (int c = 0; c < 5; ++c) tot[c] = (unsigned long long)wave_sum64((long long)(tot[c] + planes_count(a.pl, c) + a.extra[c]));
    // the bounds of the tested rows (wave-uniform) and what the byte-by-byte walks met (per lane)
    int qmin = (int)min(a.mn >> 24, (a.mn >> 8) & 0xFFu), qmax = (int)max(a.mx >> 24, (a.mx >> 8) & 0xFFu);
    if (qs.cmin <= qs.cmax) { qmin = qs.cmin < qmin ? qs.cmin : qmin; qmax = qs.cmax > qmax ? qs.cmax : qmax; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const int x = __shfl_xor(qmin, d, 64), y = __shfl_xor(qmax, d, 64);
        qmin = x < qmin ? x : qmin; qmax = y > qmax ? y : qmax;
    }
    const bool have = qmin < 255 || qmax > 0;             // a quality byte was seen
    const bool odd = __ballot(a.qodd) != 0ull || (have && (qmin < 33 || qmax > 127));
    if (lane == 0) {
        const long long ta = (long long)tot[0] + fix[0], tc = (long long)tot[1] + fix[1], tg = (long long)tot[2] + fix[2],
                        tt = (long long)tot[3] + fix[3], tn = (long long)tot[4] + fix[4];
        if (ta) atomicAdd(&acc->a, (unsigned long long)ta);
        if (tc) atomicAdd(&acc->c, (unsigned long long)tc);
        if (tg) atomicAdd(&acc->g, (unsigned long long)tg);
        if (tt) atomicAdd(&acc->t, (unsigned long long)tt);
        if (tn) atomicAdd(&acc->n, (unsigned long long)tn);
        if (have) { atomicMin(&acc->minqs, qmin); atomicMax(&acc->maxqs, qmax); }
        if (odd) atomicAdd(&acc->qfix, 1);
    }
}

// =================================================================================== index AND composition in ONE read
// k_fastq_lines_comp = k_fastq_lines (the count pass of the one-read build: granule summaries + line records) with the
// composition of the same granules counted from the same registers -- fastq.c:8-182 and fastq.c:663-795 in one pass over the
// stream, where the reference makes two (and this engine made two: 7 + 7 ms for C3).
//
// Which line of four a byte belongs to is not known while the stream is read -- the newline prefixes come later in the
// build.  A wave takes FQLC_G consecutive granules (16: 64 KiB per guess, table look-up set-up and run record -- 11.9 ms for C3
// with 4, 11.2 with 8, 11.0 with 16; the plain count pass is fastest with 4); for the first one it GUESSES: a line of exactly one byte is the '+' line
// (two newlines two bytes apart; every candidate of the granule must agree), which fixes the number-of-four of the granule's
// first line, and the granules behind it follow by counting.  The counts of a run -- A C G T N, smallest / largest quality,
// the guess -- go to an 8-word record; k_fastq_comp_reduce, after the prefixes, compares every guess with the truth
// ((line offset + nl_prefix[first granule]) & 3) and adds the records up; the runs whose guess was wrong or missing (a granule
// without a '+' line in it: reads longer than a granule; '+' lines that repeat the name; 1-base reads) are counted again by
// k_fastq_comp_stream, from the prefixes, when they are few -- else fx_fastq_comp counts from the read table as before.  A run
// without a one-byte line makes a SECOND guess from the first bytes of its lines ('@' ... two lines on '+': files whose '+' lines
// repeat the name).
#ifndef FX_FQLC_G
#define FX_FQLC_G 16
#endif
constexpr int FQLC_G = FX_FQLC_G;                          // granules per run of k_fastq_lines_comp
static_assert(FQLC_G % FQR_G == 0, "the count pass fills one record slot per FQR_G granules");
struct FqRun { uint32_t cnt[5]; uint32_t q; uint32_t guess; uint32_t pad; };   // q: min | max << 8 | have << 16 | odd << 17; guess: 0..3, 0xFF none
static_assert(sizeof(FqRun) == 32, "one run record is 32 bytes");

template <bool CRLF>
__global__ __launch_bounds__(BLOCK) void k_fastq_lines_comp(const uint8_t *__restrict__ data, int64_t n, int prev_byte, int64_t g_end,
                                                           GranPk *__restrict__ out, uint32_t *__restrict__ recs, GranList ovl,
                                                           FqRun *__restrict__ runs, int64_t nruns) {
    __shared__ __attribute__((aligned(16))) uint16_t s_sp[BLOCK / 64][GRAN / CHUNK], s_cr[BLOCK / 64][GRAN / CHUNK];   // bit k of word c <-> byte 16 c + k
    __shared__ uint16_t s_pos[BLOCK / 64][FQL_CAP];
    __shared__ uint4 s_mask[FS_NMASK];
    __shared__ int s_fix[BLOCK / 64][8];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < FS_NMASK; e += BLOCK) {
        const int r = e / 289, k1 = (e % 289) / 17, k2 = e % 17;
        const int lo = r == 0 ? -1 : r == 1 ? k1 : r == 2 ? k2 : 16, hi = r == 0 ? k1 : r == 1 ? k2 : 16;
        uint32_t m4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t m = 0;
            for (int b = 0; b < 4; ++b) { const int p = 4 * i + b; if (p > lo && p < hi) m |= 0xFFu << (8 * b); }
            m4[i] = m;
        }
        s_mask[e] = make_uint4(m4[0], m4[1], m4[2], m4[3]);
    }
    if (lane < 8) s_fix[w][lane] = 0;
    __syncthreads();
    int *fix = s_fix[w];
    // a workgroup makes its mask table once and then takes run after run (a table per 64 KiB of stream was an eighth of the work)
    for (int64_t run = (int64_t)blockIdx.x * (BLOCK / 64) + w; run < nruns; run += (int64_t)gridDim.x * (BLOCK / 64)) {
    const int64_t gw = run * FQLC_G;
    FsAcc a;
#pragma unroll
    for (int k = 0; k < COMP_NPL; ++k) a.pl.p[k] = 0;
    a.mn = 0xFFFFFFFFu; a.mx = 0u; a.qodd = false; a.pend = false;
#pragma unroll
    for (int c = 0; c < 5; ++c) a.extra[c] = 0;
    FsQ qs;
    fsq_init(qs);
    uint32_t written = 0;                                  // line records of the run so far (they stand one after the other in the run's slot)
    uint32_t guess = 0xFFu, lines_before = 0;              // number-of-four of the run's first line (0xFF: not known), newlines of the run so far
    uint4 v[GR_ROWS];
    if (gw < g_end) granule_load<true>(v, data, n, 0, gw);
    for (int kk = 0; kk < FQLC_G; ++kk) {
        const int64_t g = gw + kk;
        if (g >= g_end) break;
        const int64_t sbase = g * (int64_t)GRAN;
        if (kk % FQR_G == 0) written = 0;                      // (a record slot per FQR_G granules: what k_fastq_rows reads in one request)
        uint32_t nlm[GR_ROWS], ex[GR_ROWS], crm[GR_ROWS], run_n = 0;
        bool rcr[GR_ROWS];
#pragma unroll
        for (int j = 0; j < GR_ROWS; ++j) {
            nlm[j] = eq_mask16(v[j], 0x0A0A0A0Au);
            s_sp[w][j * 64 + lane] = (uint16_t)eq_mask16(v[j], 0x20202020u);
            if (CRLF) { crm[j] = fq_cr_mask2(v[j], rcr[j]); s_cr[w][j * 64 + lane] = (uint16_t)crm[j]; }
            else { crm[j] = 0; rcr[j] = false; s_cr[w][j * 64 + lane] = fq_cr_mask(v[j]); }
            const uint32_t cj = (uint32_t)__popc(nlm[j]);
            const uint32_t inc = wave_incl_scan(cj);
            ex[j] = run_n + inc - cj;                      // newlines of the granule in front of this chunk
            run_n += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        const uint32_t M = run_n;
        if (CRLF && __ballot(a.pend)) {                        // the granule before ended on a '\r': this one must begin with '\n'
            if (a.pend && !(__builtin_amdgcn_readlane((int)nlm[0], 0) & 1)) a.qodd = true;
            a.pend = false;
        }
        // ---- the guess, once per run: the '+' line.  One byte long in an LF file -- it ends at a newline that has another one two
        // bytes in front of it --, '+' and '\r' in a CRLF file: a newline with a '\r' in front of it and the newline before three
        // bytes back (round 5; the reference's own gzip fixture is such a file).  Which byte stands there is not looked at: every
        // such place of the run's first granule must name the same phase, and k_fastq_comp_reduce checks the guess anyway.
        if (kk == 0) {
            uint32_t mine = 0xFFu;
            bool clash = false;
#pragma unroll
            for (int j = 0; j < GR_ROWS; ++j) {
                uint32_t cand = nlm[j] & (nlm[j] << 2) & ~(nlm[j] << 1) & 0xFFFFu;     // second newline of a pair inside the chunk
                if (CRLF && rcr[j]) cand |= nlm[j] & (nlm[j] << 3) & ~(nlm[j] << 1) & ~(nlm[j] << 2) & (crm[j] << 1) & 0xFFFFu;
                if (cand) {
                    const uint32_t b = (uint32_t)__ffs(cand) - 1u;
                    const uint32_t i = ex[j] + (uint32_t)__popc(nlm[j] & ((1u << b) - 1u));   // that newline is the granule's i-th: line i is line 2 of four
                    const uint32_t p0 = (2u - i) & 3u;
                    if (mine == 0xFFu) mine = p0; else clash |= mine != p0;
                }
            }
            const unsigned long long hv = __ballot(mine != 0xFFu);
            if (hv) {
                guess = (uint32_t)__builtin_amdgcn_readlane((int)mine, __ffsll((long long)hv) - 1);
                if (__ballot(clash || (mine != 0xFFu && mine != guess))) guess = 0xFFu;
            }
        }
        // ---- no line of one byte in the run's first granule ('+' lines that repeat the name: what fastq-dump writes): the second
        // guess, on this rare path only -- a line that begins with '@' whose next line but one begins with '+' is a header line.  (A
        // quality line may begin with '@', but the line two behind it is then a sequence line; k_fastq_comp_reduce checks every
        // guess anyway.)  The first byte of the line behind every newline goes to LDS, one slot per newline; lanes then look at
        // slots i and i + 2.
        if (kk == 0 && guess == 0xFFu && M >= 3u && M <= (uint32_t)FQL_CAP) {
            uint16_t *cls = s_pos[w];                          // (free here: the compaction below fills it after the composition)
            // (maps of the '@' and '+' bytes, not bytes picked out of the chunk by a run-time index: that would put every chunk of the
            // straight path into scratch memory)
            uint32_t at16[GR_ROWS], pl16[GR_ROWS];
#pragma unroll
            for (int j = 0; j < GR_ROWS; ++j) { at16[j] = eq_mask16(v[j], 0x40404040u); pl16[j] = eq_mask16(v[j], 0x2B2B2B2Bu); }
#pragma unroll
            for (int j = 0; j < GR_ROWS; ++j) {
                // bit 16 of a chunk's maps: the first byte of the chunk behind it (the next lane's, the next row's first; none behind the granule)
                const uint32_t fa = at16[j] & 1u, fp = pl16[j] & 1u;
                const uint32_t na = (uint32_t)__shfl_down((int)fa, 1, 64), np = (uint32_t)__shfl_down((int)fp, 1, 64);
                const uint32_t ra = j + 1 < GR_ROWS ? (uint32_t)__builtin_amdgcn_readlane((int)at16[j + 1 < GR_ROWS ? j + 1 : j], 0) & 1u : 0u;
                const uint32_t rp = j + 1 < GR_ROWS ? (uint32_t)__builtin_amdgcn_readlane((int)pl16[j + 1 < GR_ROWS ? j + 1 : j], 0) & 1u : 0u;
                const uint32_t am = at16[j] | ((lane < 63 ? na : ra) << 16), pm = pl16[j] | ((lane < 63 ? np : rp) << 16);
                uint32_t m = nlm[j], r = ex[j];
                while (m) {
                    const int k = __ffs(m) - 1;
                    m &= m - 1;
                    cls[r++] = (uint16_t)(((am >> (k + 1)) & 1u) ? 1u : ((pm >> (k + 1)) & 1u) ? 2u : 0u);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            uint32_t mine = 0xFFu;
            bool clash = false;
            for (uint32_t i = lane; i + 2u < M; i += 64u)
                if (cls[i] == 1u && cls[i + 2u] == 2u) {       // the line behind newline i is a header line: it is line i + 1 of the granule
                    const uint32_t p0 = (3u - i) & 3u;
                    if (mine == 0xFFu) mine = p0; else clash |= mine != p0;
                }
            const unsigned long long hv = __ballot(mine != 0xFFu);
            if (hv) {
                guess = (uint32_t)__builtin_amdgcn_readlane((int)mine, __ffsll((long long)hv) - 1);
                if (__ballot(clash || (mine != 0xFFu && mine != guess))) guess = 0xFFu;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // ---- composition of this granule (only with a guess: without one the run's record says so and nothing is used)
        if (guess != 0xFFu) {
            const uint32_t L0 = (guess + lines_before) & 3u;
            CompCarry cy;
            fs_row<0, CRLF>(v[0], nlm[0], crm[0], rcr[0], nlm[1], (L0 + ex[0]) & 3u, s_mask, a, cy, qs, fix, lane);
            fs_row<1, CRLF>(v[1], nlm[1], crm[1], rcr[1], nlm[2], (L0 + ex[1]) & 3u, s_mask, a, cy, qs, fix, lane);
            fs_row<2, CRLF>(v[2], nlm[2], crm[2], rcr[2], nlm[3], (L0 + ex[2]) & 3u, s_mask, a, cy, qs, fix, lane);
            fs_row<3, CRLF>(v[3], nlm[3], crm[3], rcr[3], 0u, (L0 + ex[3]) & 3u, s_mask, a, cy, qs, fix, lane);
            planes_finish16(a.pl, cy);
        }
        lines_before += M;
        if (kk + 1 < FQLC_G && g + 1 < g_end) granule_load<true>(v, data, n, 0, g + 1);       // the next granule is on its way
        int first = GRAN, last = -1;
        const bool over = M > (uint32_t)FQL_CAP;              // more lines than a slot holds: k_fastq_emit reads the granule again
        if (over) {
            fq_first_last(nlm, lane, first, last);
            if (lane == 0) ovl.g[atomicAdd(ovl.count, 1u)] = (uint32_t)g;
        } else if (M) {
            // ---- compact the newline positions: the first and the last of them are the granule's
#pragma unroll
            for (int j = 0; j < GR_ROWS; ++j) {
                uint32_t m = nlm[j], r = ex[j];
                while (m) {
                    const int k = __ffs(m) - 1;
                    m &= m - 1;
                    s_pos[w][r++] = (uint16_t)(j * 1024 + lane * CHUNK + k);
                }
            }
            first = s_pos[w][0]; last = s_pos[w][M - 1];
        }
        if (lane == 0) {
            GranOut o;
            o.n = M; o.h = 0; o.first = (uint32_t)first; o.last = (uint32_t)last;
            o.v1 = o.c1 = o.v2 = o.c2 = o.ovf = 0;
            out[g] = gran_pack(o);
        }
        if (over || !M) continue;
        fq_line_records(data, sbase, prev_byte, s_pos[w], M, reinterpret_cast<const uint64_t *>(&s_sp[w][0]), &s_cr[w][0],
                        recs + (g / FQR_G) * (int64_t)(FQR_G * FQL_CAP) + written, lane);
        written += M;
    }
    // ---- the run's record
    {
        if (CRLF && __ballot(a.pend)) {                        // the run's last byte is a '\r': the byte behind it, from memory (the end of the stream passes)
            const int64_t q = (gw + FQLC_G < g_end ? gw + FQLC_G : g_end) * (int64_t)GRAN;
            if (a.pend && q < n && data[q] != 10) a.qodd = true;
        }
        uint32_t tot[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) tot[c] = wave_sum(planes_count(a.pl, c) + a.extra[c]);
        int qmin = (int)min(a.mn >> 24, (a.mn >> 8) & 0xFFu), qmax = (int)max(a.mx >> 24, (a.mx >> 8) & 0xFFu);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const int x = __shfl_xor(qmin, d, 64), y = __shfl_xor(qmax, d, 64);
            qmin = x < qmin ? x : qmin; qmax = y > qmax ? y : qmax;
        }
        if (qs.cmin <= qs.cmax) { qmin = qs.cmin < qmin ? qs.cmin : qmin; qmax = qs.cmax > qmax ? qs.cmax : qmax; }
        const bool have = qmin < 255 || qmax > 0;
        const bool odd = __ballot(a.qodd) != 0ull || (have && (qmin < 33 || qmax > 127));
        if (lane == 0) {
            FqRun r;
#pragma unroll
            for (int c = 0; c < 5; ++c) { r.cnt[c] = (uint32_t)((int)tot[c] + fix[c]); fix[c] = 0; }
            r.q = (uint32_t)(qmin & 0xFF) | ((uint32_t)(qmax & 0xFF) << 8) | (have ? 1u << 16 : 0u) | (odd ? 1u << 17 : 0u);
            r.guess = guess; r.pad = 0;
            runs[run] = r;
        }
    }
    }
}

// after the prefixes: every run's guess against the truth, the records added up; the ragged end of the stream (less than a
// granule, no run) walked by the first wave.  res: FastqAcc (a c g t n, minqs, maxqs; qfix = runs with bytes the stream form leaves
// to the table kernels).  A run whose guess was wrong or missing goes on the list `rej` (rej[0] = how many, then the runs):
// k_fastq_comp_stream counts those again with the line numbers the prefixes give (round 5; round 4 dropped the whole result).
__global__ __launch_bounds__(BLOCK) void k_fastq_comp_reduce(const FqRun *__restrict__ runs, int64_t nruns, const int64_t *__restrict__ nl_prefix,
                                                            int64_t line0, const uint8_t *__restrict__ data, int64_t n, int64_t nfull,
                                                            FastqAcc *res, uint32_t *__restrict__ rej) {
    unsigned long long tot[5] = {0, 0, 0, 0, 0};
    int qmin = 255, qmax = 0, bad = 0;
    for (int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; r < nruns; r += (int64_t)gridDim.x * BLOCK) {
        const FqRun x = runs[r];
        const uint32_t truth = (uint32_t)((line0 + nl_prefix[r * FQLC_G]) & 3);
        if (x.guess != truth) { rej[1u + atomicAdd(&rej[0], 1u)] = (uint32_t)r; continue; }
        if ((x.q >> 17) & 1u) { ++bad; continue; }
#pragma unroll
        for (int c = 0; c < 5; ++c) tot[c] += x.cnt[c];
        if ((x.q >> 16) & 1u) { const int lo = (int)(x.q & 0xFFu), hi = (int)((x.q >> 8) & 0xFFu); qmin = lo < qmin ? lo : qmin; qmax = hi > qmax ? hi : qmax; }
    }
    if (blockIdx.x == 0 && threadIdx.x < 64 && nfull * (int64_t)GRAN < n) {        // the ragged end
        FsAcc a;
        a.mn = 0xFFFFFFFFu; a.mx = 0u; a.qodd = false; a.pend = false;
#pragma unroll
        for (int c = 0; c < 5; ++c) a.extra[c] = 0;
        const int lane = lane_id();
        const int64_t tail0 = nfull * (int64_t)GRAN, lo = tail0 + (int64_t)lane * 64, hi = lo + 64 < n ? lo + 64 : n;
        uint32_t c = 0;
        for (int64_t p = lo; p < hi; ++p) c += data[p] == 10;
        const uint32_t inc = wave_incl_scan(c);
        uint32_t ph = (uint32_t)((line0 + nl_prefix[nfull] + (int64_t)(inc - c)) & 3);
        for (int64_t p = lo; p < hi; ++p) {
            const uint32_t b = data[p];
            if (b == 10u) { ph = (ph + 1u) & 3u; continue; }
            fs_one_cr(b, p + 1 < n ? data[p + 1] : 10u, ph, a);
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) tot[k] += a.extra[k];
        const int l2 = (int)min(a.mn >> 24, (a.mn >> 8) & 0xFFu), h2 = (int)max(a.mx >> 24, (a.mx >> 8) & 0xFFu);
        if (l2 < 255 || h2 > 0) { qmin = l2 < qmin ? l2 : qmin; qmax = h2 > qmax ? h2 : qmax; }
        if (a.qodd) ++bad;
    }
#pragma unroll
    for (int c = 0; c < 5; ++c) tot[c] = (unsigned long long)wave_sum64((long long)tot[c]);
    bad = (int)wave_sum((uint32_t)bad);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const int x = __shfl_xor(qmin, d, 64), y = __shfl_xor(qmax, d, 64);
        qmin = x < qmin ? x : qmin; qmax = y > qmax ? y : qmax;
    }
    if (lane_id() == 0) {
        if (tot[0]) atomicAdd(&res->a, tot[0]);
        if (tot[1]) atomicAdd(&res->c, tot[1]);
        if (tot[2]) atomicAdd(&res->g, tot[2]);
        if (tot[3]) atomicAdd(&res->t, tot[3]);
        if (tot[4]) atomicAdd(&res->n, tot[4]);
        if (qmin < 255 || qmax > 0) { atomicMin(&res->minqs, qmin); atomicMax(&res->maxqs, qmax); }
        if (bad) atomicAdd(&res->qfix, bad);
    }
}

}  // namespace fx
