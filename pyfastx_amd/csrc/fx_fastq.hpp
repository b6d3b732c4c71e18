// fx_fastq.hpp -- FASTQ index build for gfx950 (MI355X, wave64), on top of the granule scan.
//
// The reference runs a `line_num % 4` state machine over the lines of the file (fastq.c:89-149).
// A FASTQ record is "four lines", so the only global fact a byte range needs is the line number of
// its first newline.  Build = two streaming reads of the stream:
//
//   k_span_scan<1> + k_gran_reduce<1> + k_gran_prefix     newline count / first / last per 4 KiB
//        granule and their exclusive prefixes (the same kernels the FASTA build uses, count-only mode)
//   k_fastq_emit      one wave per granule: re-read its 4 KiB (kept in LDS), compact the newline
//        positions into LDS, then ONE LANE PER NEWLINE: global line index = loff + nl_prefix[g] + rank,
//        phase = index & 3, and the lane writes the field(s) of record index >> 2 that this newline
//        determines (header end: name_off / name_len / dlen / soff; sequence end: rlen; '+' line end:
//        qoff; quality end: qlen).  No line table, no record-level gathers from memory, no atomics.
//
// (A single-pass variant -- the emit kernel getting its line numbers by a decoupled look-back over
// per-granule counts published by the other waves -- was built and measured: correct, but 213 ms instead
// of 2.8 ms for 7 GB.  With 1.7 M four-KiB tiles in flight 8 k at a time, every wave walks back over
// thousands of "counted, prefix not yet known" words with device-scope loads; look-back needs tiles
// that are large against the number in flight, and large tiles would have to sit in LDS while they wait.
// Two streaming reads at 6.8 and 4.5 TB/s are the better trade.)
//
// Byte-range shards use the same two kernels: the count pass is fx_fastq_scan, one
// all-gather of two integers gives every rank `loff` / `prev_nl`, the emit pass is fx_fastq_build_ctx.
#pragma once
#include "fx_spanscan.hpp"

#ifndef FX_FQ_U
#define FX_FQ_U 2                       // records per 16-lane group and iteration in k_fastq_comp (10 M reads: 1.18 ms; 3: 1.22, 4: 1.21)
#endif

namespace fx {

struct FqTab { int64_t *name_off, *rlen, *soff, *qoff; int32_t *name_len, *dlen, *qlen; };
struct FqOwn {
    int64_t loff;            // global line index of the shard's first newline (newlines in earlier shards' cores)
    int64_t prev_nl;         // global offset of the last newline before the shard (-1: none)
    int64_t k_first, nrows;  // records owned by this shard: global ids [k_first, k_first + nrows)
};
struct FastqAcc {            // device accumulators
    unsigned long long size;
    unsigned long long a, c, g, t, n;
    long long maxlen, minlen;
    int minqs, maxqs;
    int pad0, pad1;
};

constexpr int FQ_POSCAP = 512;           // newline positions per wave per round (more only for lines < 8 bytes on average)

// first ' ' in bytes [nb, ne) of `base` (8 aligned bytes at a time), or ne  (fastq.c:112-117)
__device__ __forceinline__ int64_t first_space(const uint8_t *base, int64_t nb, int64_t ne) {
    for (int64_t p = nb & ~7ll; p < ne; p += 8) {
        const uint2 w = *reinterpret_cast<const uint2 *>(base + p);
        uint32_t m = flags4(zero_bytes(w.x ^ 0x20202020u)) | (flags4(zero_bytes(w.y ^ 0x20202020u)) << 4);
        if (p < nb) m &= 0xFFu << (nb - p);
        if (m) { const int64_t hit = p + __ffs(m) - 1; return hit < ne ? hit : ne; }
    }
    return ne;
}

__global__ __launch_bounds__(BLOCK) void k_fastq_emit(ScanCtx x, int prev_byte, int is_last, FqOwn own, FqTab t) {
    __shared__ uint4 s_data[BLOCK / 64][GRAN / 16];
    __shared__ uint16_t s_pos[BLOCK / 64][FQ_POSCAP];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const int64_t g = (int64_t)blockIdx.x * (BLOCK / 64) + w;
    if (g >= x.ngran) return;                              // waves are independent: no workgroup barrier below
    const int64_t sbase = g * (int64_t)GRAN;
    uint4 v[GR_ROWS];
    if (sbase + GRAN <= x.n) {
        const uint4 *q = reinterpret_cast<const uint4 *>(x.data + sbase + lane * CHUNK);
#pragma unroll
        for (int j = 0; j < GR_ROWS; ++j) {                // streamed once: non-temporal, like the scan
            v[j].x = __builtin_nontemporal_load(&q[j * 64].x); v[j].y = __builtin_nontemporal_load(&q[j * 64].y);
            v[j].z = __builtin_nontemporal_load(&q[j * 64].z); v[j].w = __builtin_nontemporal_load(&q[j * 64].w);
        }
    } else {
#pragma unroll
        for (int j = 0; j < GR_ROWS; ++j) v[j] = load16(x.data, sbase + j * 1024 + lane * CHUNK, x.n);
        if (is_last) {                                     // virtual end-of-stream newline (fastq.c:148)
#pragma unroll
            for (int j = 0; j < GR_ROWS; ++j) {
                const int64_t p = sbase + j * 1024 + lane * CHUNK;
                if (x.n >= p && x.n < p + CHUNK && x.n > 0 && x.data[x.n - 1] != '\n') {
                    const int k = (int)(x.n - p);
                    const uint32_t b = 0x0Au << ((k & 3) * 8);
                    if ((k >> 2) == 0) v[j].x |= b; else if ((k >> 2) == 1) v[j].y |= b; else if ((k >> 2) == 2) v[j].z |= b; else v[j].w |= b;
                }
            }
        }
    }
    uint32_t nlm[GR_ROWS], ex[GR_ROWS];                    // newline mask of the lane's chunk, rank of its first newline in the granule
    uint32_t M = 0;
#pragma unroll
    for (int j = 0; j < GR_ROWS; ++j) {
        s_data[w][j * 64 + lane] = v[j];
        nlm[j] = eq_mask16(v[j], 0x0A0A0A0Au);
        const uint32_t c = __popc(nlm[j]);
        const uint32_t inc = wave_incl_scan(c);
        ex[j] = M + inc - c;
        M += (uint32_t)__shfl((int)inc, 63, 64);
    }
    if (!M) return;
    const uint8_t *sb = reinterpret_cast<const uint8_t *>(&s_data[w][0]);
    const int64_t gs = x.gbase + sbase;                    // global offset of the granule
    const int64_t I0 = own.loff + x.nl_prefix[g];          // global line index of the granule's first newline
    int64_t q_carry = x.prevnl[g];                         // newline before the current round's first one
    if (q_carry < 0) q_carry = own.prev_nl;
    for (uint32_t lo = 0; lo < M; lo += FQ_POSCAP) {
        // ---- compact the newline positions of ranks [lo, lo + FQ_POSCAP) into LDS
#pragma unroll
        for (int j = 0; j < GR_ROWS; ++j) {
            uint32_t m = nlm[j], r = ex[j];
            while (m) {
                const int k = __ffs(m) - 1;
                m &= m - 1;
                if (r - lo < (uint32_t)FQ_POSCAP) s_pos[w][r - lo] = (uint16_t)(j * 1024 + lane * CHUNK + k);
                ++r;
            }
        }
        const uint32_t cnt = (M - lo < (uint32_t)FQ_POSCAP) ? M - lo : (uint32_t)FQ_POSCAP;
        // ---- one lane per newline
        for (uint32_t t0 = 0; t0 < cnt; t0 += 64) {
            const uint32_t i = t0 + lane;
            if (i >= cnt) continue;
            const int lp = s_pos[w][i];
            const int64_t p = gs + lp;
            const int64_t q = i ? gs + s_pos[w][i - 1] : q_carry;     // previous newline (-1: none)
            const int64_t idx = I0 + lo + i;
            const int64_t row = (idx >> 2) - own.k_first;
            if (row < 0 || row >= own.nrows) continue;
            const int64_t len = p - q - 1;                            // bytes of the line that ends at p
            // is the byte before the newline a '\r'?
            int cr = 0;
            if (len > 0) {
                const int b = lp ? sb[lp - 1] : (sbase ? x.data[sbase - 1] : prev_byte);
                cr = b == '\r';
            }
            switch ((int)(idx & 3)) {
            case 0: {                                                 // header line  (fastq.c:99-117)
                const int64_t s0 = q + 1;                             // '@'
                int64_t nlen = len - 1;
                if (nlen > 0 && cr) --nlen;                           // fastq.c:107-109
                const int64_t nb = s0 + 1;
                int64_t hit;
                if (nb >= gs) hit = gs + first_space(sb, nb - gs, nb - gs + nlen);            // the name lies in this granule
                else          hit = x.gbase + first_space(x.data, nb - x.gbase, nb - x.gbase + nlen);
                t.name_off[row] = nb; t.name_len[row] = (int32_t)(hit - nb); t.dlen[row] = (int32_t)len;   // fastq.c:103: '@' and '\r' included
                t.soff[row] = p + 1;                                  // fastq.c:122
                break;
            }
            case 1: t.rlen[row] = len - cr; break;                    // fastq.c:124-128
            case 2: t.qoff[row] = p + 1; break;                       // fastq.c:133
            default: t.qlen[row] = (int32_t)(len - cr); break;        // quality line, trailing CR dropped (fastq.c:734-737)
            }
        }
        q_carry = gs + s_pos[w][cnt - 1];
    }
}

// newlines of the shard at a local offset < cut: out[0] = count, out[1] = global offset of the last (-1: none)
__device__ __forceinline__ void count_below(const ScanCtx &x, int is_last, int64_t cut, int64_t *out) {
    const int lane = lane_id();
    int64_t g = cut / GRAN;
    if (g >= x.ngran) { if (lane == 0) { out[0] = x.nl_prefix[x.ngran]; out[1] = x.prevnl[x.ngran]; } return; }
    const int64_t sbase = g * (int64_t)GRAN;
    int64_t cnt = x.nl_prefix[g], last = x.prevnl[g];
    for (int j = 0; j < GR_ROWS; ++j) {
        const int64_t p = sbase + j * 1024 + lane * CHUNK;
        uint4 v = load16(x.data, p, x.n);
        if (is_last && x.n >= p && x.n < p + CHUNK && x.n > 0 && x.data[x.n - 1] != '\n') {
            const int k = (int)(x.n - p);
            const uint32_t b = 0x0Au << ((k & 3) * 8);
            if ((k >> 2) == 0) v.x |= b; else if ((k >> 2) == 1) v.y |= b; else if ((k >> 2) == 2) v.z |= b; else v.w |= b;
        }
        uint32_t m = eq_mask16(v, 0x0A0A0A0Au);
        const int64_t keep = cut - p;                      // bytes of this chunk below the cut
        m &= keep >= CHUNK ? 0xFFFFu : (keep <= 0 ? 0u : ((1u << keep) - 1u));
        cnt += wave_sum_small(__popc(m));
        const unsigned long long b = __ballot(m != 0);
        if (b) {
            const int l = 63 - __clzll(b);
            last = x.gbase + sbase + j * 1024 + l * CHUNK + (31 - __clz(rdlane(m, l)));
        }
    }
    if (lane == 0) { out[0] = cnt; out[1] = last; }
}
// out[0..1]: newlines below `cut` (count, last offset); out[2]: count below cut - 1
__global__ __launch_bounds__(64) void k_core_count(ScanCtx x, int is_last, int64_t cut, int64_t *out) {
    count_below(x, is_last, cut, out);
    __shared__ int64_t tmp[2];
    count_below(x, is_last, cut - 1, tmp);
    if (lane_id() == 0) out[2] = tmp[0];
}

// stat.size = sum of rlen over the rows that have a sequence line; meta.maxlen / minlen over complete rows
__global__ __launch_bounds__(BLOCK) void k_fastq_stats(FqTab t, int64_t n_seq_rows, int64_t n_rows, FastqAcc *acc) {
    __shared__ long long red[3][BLOCK / 64];
    long long s = 0, mx = 0, mn = 10000000000LL;
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_seq_rows; i += stride) {
        s += t.rlen[i];
        if (i < n_rows) { const long long ql = t.qlen[i]; mx = ql > mx ? ql : mx; mn = ql < mn ? ql : mn; }
    }
    s = wave_sum64(s);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const long long a = __shfl_xor(mx, d, 64), b = __shfl_xor(mn, d, 64);
        mx = a > mx ? a : mx; mn = b < mn ? b : mn;
    }
    const int w = threadIdx.x >> 6;
    if (lane_id() == 0) { red[0][w] = s; red[1][w] = mx; red[2][w] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long S = 0, X = 0, N = 10000000000LL;
        for (int i = 0; i < BLOCK / 64; ++i) { S += red[0][i]; X = red[1][i] > X ? red[1][i] : X; N = red[2][i] < N ? red[2][i] : N; }
        if (S) atomicAdd(&acc->size, (unsigned long long)S);
        atomicMax(&acc->maxlen, X); atomicMin(&acc->minlen, N);
    }
}

// FASTQ composition (fastq.c:715-753): A C G T (upper case only) over the sequence lines, every other byte but
// '\r' is N; min / max byte of the quality lines, '\r' ignored.  16 lanes per record, 4 records per wave; a lane
// takes 16 bytes of the line per step (unaligned 16-byte loads from the line start, so only the last piece of a
// line is partial; bytes past its end are replaced by '\r').
// Sequence: no compare per letter -- as in k_fasta_comp (fx_comp.hpp) the 3-bit code (b >> 1) & 7 picks a one-hot
// class byte (A C G T N '\r') through v_perm_b32 and the byte it stands for through a second one; the four words
// of a piece are added bit-plane-wise (three full adders), and only the carry-out word ("four more at this bit") is
// popcounted per class.  A piece with a byte outside A C G T N '\r' (lower case, IUPAC codes: N for the reference)
// is fixed up byte by byte through a per-wave LDS array: take the aliased class back out, add one N.
// Quality: the 16 bytes go through packed 16-bit min / max (v_pk_min_u16 / v_pk_max_u16 on the even and odd
// bytes); a piece whose smallest byte is below '!' or whose largest is above 127 -- a '\r', or bytes the reference
// reads as negative chars -- takes the exact per-byte loop instead.
constexpr uint32_t FQ_OH_LO = 0x04080201u, FQ_OH_HI = 0x10200000u;     // A 1, C 2, T 8, G 4 | -, -, \r 32, N 16
constexpr uint32_t FQ_EX_LO = 0x47544341u, FQ_EX_HI = 0x4E0D8080u;     // 'A' 'C' 'T' 'G' | none, none, '\r', 'N'

__device__ const uint4 fq_sixteen_zeros = {0u, 0u, 0u, 0u};
typedef unsigned short __attribute__((ext_vector_type(2))) fq_u16x2;
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    const fq_u16x2 r = __builtin_elementwise_min(__builtin_bit_cast(fq_u16x2, a), __builtin_bit_cast(fq_u16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    const fq_u16x2 r = __builtin_elementwise_max(__builtin_bit_cast(fq_u16x2, a), __builtin_bit_cast(fq_u16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}
// 16 bytes at p (any alignment); within 16 bytes of the end of the blob: byte by byte, zeros past the end
__device__ __forceinline__ uint4 fq_load16(const uint8_t *__restrict__ data, int64_t p, int64_t n_bytes) {
    if (p + 16 <= n_bytes) return *reinterpret_cast<const uint4_u *>(data + p);
    uint32_t w[4] = {0, 0, 0, 0};
    for (int k = 0; k < 16; ++k) if (p + k < n_bytes) w[k >> 2] |= (uint32_t)data[p + k] << ((k & 3) * 8);
    return make_uint4(w[0], w[1], w[2], w[3]);
}
// keep the first `keep` (1..15) bytes of v, the others become `fill` (a byte replicated over the word)
__device__ __forceinline__ void fq_keep_first(uint32_t (&x)[4], int keep, uint32_t fill) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = keep - 4 * i;
        const uint32_t m = k >= 4 ? 0xFFFFFFFFu : k <= 0 ? 0u : ((1u << (8 * k)) - 1u);
        x[i] = (x[i] & m) | (fill & ~m);
    }
}

__global__ __launch_bounds__(BLOCK) void k_fastq_comp(const uint8_t *__restrict__ data, int64_t gbase, int64_t n_bytes, FqTab t,
                                                     int64_t n_seq_rows, int64_t n_rows, FastqAcc *acc) {
    __shared__ int fix_all[BLOCK / 64][8];                 // per wave: signed corrections of the class counts (slot = class bit index)
    const int lane = lane_id(), sub = lane & 15, grp = lane >> 4, wv = threadIdx.x >> 6;
    int *fix = fix_all[wv];
    if (lane < 8) fix[lane] = 0;
    const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
    uint32_t ones = 0, twos = 0;                           // bit planes of the one-hot words: 1s and 2s per bit position
    uint32_t c4[6] = {0, 0, 0, 0, 0, 0};                   // per class: popcount of the carry-outs (each worth 4)
    int qmin = 104, qmax = 33;                             // fastq.c:667-668

    auto seq_piece = [&](const uint4 &v, int64_t left) {   // 16 bytes of a sequence line, `left` of them inside it
        uint32_t x[4] = {v.x, v.y, v.z, v.w};
        if (left < 16) fq_keep_first(x, (int)left, 0x0D0D0D0Du);
        uint32_t h[4], dacc = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t code = (x[k] >> 1) & 0x07070707u;
            h[k] = __builtin_amdgcn_perm(FQ_OH_HI, FQ_OH_LO, code);
            dacc |= x[k] ^ __builtin_amdgcn_perm(FQ_EX_HI, FQ_EX_LO, code);
        }
        if (dacc) {                                        // a byte that is none of A C G T N \r: N for the reference
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                const uint32_t code = (x[k] >> 1) & 0x07070707u;
                const uint32_t d = x[k] ^ __builtin_amdgcn_perm(FQ_EX_HI, FQ_EX_LO, code);
                if (!d) continue;
#pragma unroll 1
                for (int j = 0; j < 4; ++j) {
                    if (!((d >> (8 * j)) & 0xFFu)) continue;
                    const uint32_t hk = (h[k] >> (8 * j)) & 0xFFu;     // what the planes take this byte for
                    if (hk) atomicSub(&fix[__ffs(hk) - 1], 1);
                    atomicAdd(&fix[4], 1);
                }
            }
        }
        const uint32_t tA = __builtin_amdgcn_bitop3_b32(ones, h[0], h[1], 0xE8);
        ones = __builtin_amdgcn_bitop3_b32(ones, h[0], h[1], 0x96);
        const uint32_t tB = __builtin_amdgcn_bitop3_b32(ones, h[2], h[3], 0xE8);
        ones = __builtin_amdgcn_bitop3_b32(ones, h[2], h[3], 0x96);
        const uint32_t f = __builtin_amdgcn_bitop3_b32(twos, tA, tB, 0xE8);
        twos = __builtin_amdgcn_bitop3_b32(twos, tA, tB, 0x96);
#pragma unroll
        for (int c = 0; c < 5; ++c) c4[c] += __popc(f & (0x01010101u << c));
    };
    auto qual_piece = [&](const uint4 &v, int64_t left) {  // 16 bytes of a quality line
        uint32_t lo[4] = {v.x, v.y, v.z, v.w}, hi[4] = {v.x, v.y, v.z, v.w};
        const int keep = left < 16 ? (int)left : 16;
        if (keep < 16) { fq_keep_first(lo, keep, 0xFFFFFFFFu); fq_keep_first(hi, keep, 0u); }
        uint32_t mn = 0x00FF00FFu, mx = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mn = pk_min_u16(mn, pk_min_u16(lo[k] & 0x00FF00FFu, (lo[k] >> 8) & 0x00FF00FFu));
            mx = pk_max_u16(mx, pk_max_u16(hi[k] & 0x00FF00FFu, (hi[k] >> 8) & 0x00FF00FFu));
        }
        const int cmin = (int)min(mn & 0xFFFFu, mn >> 16), cmax = (int)max(mx & 0xFFFFu, mx >> 16);
        if (cmin >= 33 && cmax < 128) {
            qmin = cmin < qmin ? cmin : qmin; qmax = cmax > qmax ? cmax : qmax;
        } else {                                           // '\r' (skipped) or bytes outside the printable range: exactly as fastq.c:733-737
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            for (int j = 0; j < keep; ++j) {
                const int q = (int)(signed char)(w[j >> 2] >> ((j & 3) * 8));
                if (q == 13) continue;
                qmin = q < qmin ? q : qmin; qmax = q > qmax ? q : qmax;
            }
        }
    };

    // A wave takes FQ_RPW = 4 * FQ_U consecutive records per iteration (FQ_U per 16-lane group).  Table rows are read
    // one iteration ahead; the first piece of every sequence line and quality line of the iteration is requested
    // before any of them is counted (reads up to 256 bytes need no more than those loads): 2 * FQ_U loads in
    // flight per lane -- with one record per group and loads inside branches the kernel was bound by load round trips, 2.8 ms for 7 GB.
    constexpr int FQ_U = FX_FQ_U, FQ_RPW = 4 * FQ_U;
    const uint8_t *safe = n_bytes >= 16 ? data : reinterpret_cast<const uint8_t *>(&fq_sixteen_zeros);
    const int64_t stride = nwaves * FQ_RPW;
    int64_t i0 = wave * FQ_RPW + grp * FQ_U;               // first record of this group in this iteration
    // the next iteration's rows, raw: the loads are unconditional (index clamped) and nothing is computed from them
    // until the next pass -- anything else makes every one of them a round trip of its own
    int64_t r_soff[FQ_U], r_rlen[FQ_U], r_qoff[FQ_U];
    int32_t r_qlen[FQ_U];
    const int64_t last_seq = n_seq_rows - 1, last_row = n_rows > 0 ? n_rows - 1 : 0;
#pragma unroll
    for (int u = 0; u < FQ_U; ++u) {
        const int64_t is = i0 + u < last_seq ? i0 + u : last_seq, iq = i0 + u < last_row ? i0 + u : last_row;
        r_soff[u] = t.soff[is]; r_rlen[u] = t.rlen[is]; r_qoff[u] = t.qoff[iq]; r_qlen[u] = t.qlen[iq];
    }
    for (int64_t t0 = wave * FQ_RPW; t0 < n_seq_rows; t0 += stride) {
        int64_t ps[FQ_U], e[FQ_U], pq[FQ_U], qe[FQ_U];
        uint4 vs[FQ_U], vq[FQ_U];
#pragma unroll
        for (int u = 0; u < FQ_U; ++u) {
            const int64_t i = i0 + u;
            const int64_t s = r_soff[u] - gbase, q = r_qoff[u] - gbase;
            ps[u] = s + sub * 16; e[u] = i < n_seq_rows ? s + r_rlen[u] : 0;       // e <= ps: nothing to do
            pq[u] = q + sub * 16; qe[u] = i < n_rows ? q + r_qlen[u] : 0;
        }
        // the 2 * FQ_U loads of the iteration, branch-free (a lane with nothing to read, or within 16 bytes of the end
        // of the blob, reads 16 bytes that are always there instead): loads inside branches are waited for one by one
#pragma unroll
        for (int u = 0; u < FQ_U; ++u) {
            const bool ls = ps[u] < e[u] && ps[u] + 16 <= n_bytes, lq = pq[u] < qe[u] && pq[u] + 16 <= n_bytes;
            vs[u] = *reinterpret_cast<const uint4_u *>(ls ? data + ps[u] : safe);
            vq[u] = *reinterpret_cast<const uint4_u *>(lq ? data + pq[u] : safe);
        }
        i0 += stride;                                      // next iteration's rows: requested after the data, used after the counting
#pragma unroll
        for (int u = 0; u < FQ_U; ++u) {
            const int64_t is = i0 + u < last_seq ? i0 + u : last_seq, iq = i0 + u < last_row ? i0 + u : last_row;
            r_soff[u] = t.soff[is]; r_rlen[u] = t.rlen[is]; r_qoff[u] = t.qoff[iq]; r_qlen[u] = t.qlen[iq];
        }
#pragma unroll
        for (int u = 0; u < FQ_U; ++u) {                   // the last bytes of the blob: byte by byte
            if (ps[u] < e[u] && ps[u] + 16 > n_bytes) vs[u] = fq_load16(data, ps[u], n_bytes);
            if (pq[u] < qe[u] && pq[u] + 16 > n_bytes) vq[u] = fq_load16(data, pq[u], n_bytes);
        }
#pragma unroll
        for (int u = 0; u < FQ_U; ++u) {
            if (ps[u] < e[u]) {
                seq_piece(vs[u], e[u] - ps[u]);
                for (int64_t p = ps[u] + 256; p < e[u]; p += 256) seq_piece(fq_load16(data, p, n_bytes), e[u] - p);
            }
            if (pq[u] < qe[u]) {
                qual_piece(vq[u], qe[u] - pq[u]);
                for (int64_t p = pq[u] + 256; p < qe[u]; p += 256) qual_piece(fq_load16(data, p, n_bytes), qe[u] - p);
            }
        }
    }
    // class totals of the lane: 1 * ones + 2 * twos + 4 * carry-outs, then the wave, then the accumulators
    unsigned long long tot[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const uint32_t m = 0x01010101u << c;
        tot[c] = wave_sum64((long long)(__popc(ones & m) + 2 * __popc(twos & m) + 4 * c4[c]));
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        int a = __shfl_xor(qmin, d, 64), b = __shfl_xor(qmax, d, 64);
        qmin = a < qmin ? a : qmin; qmax = b > qmax ? b : qmax;
    }
    if (lane == 0) {
        // class bits: 0 A, 1 C, 2 G, 3 T, 4 N (bit 5, '\r', is not counted)
        const long long ta = (long long)tot[0] + fix[0], tc = (long long)tot[1] + fix[1], tg = (long long)tot[2] + fix[2],
                        tt = (long long)tot[3] + fix[3], tn = (long long)tot[4] + fix[4];
        if (ta) atomicAdd(&acc->a, (unsigned long long)ta); if (tc) atomicAdd(&acc->c, (unsigned long long)tc);
        if (tg) atomicAdd(&acc->g, (unsigned long long)tg); if (tt) atomicAdd(&acc->t, (unsigned long long)tt);
        if (tn) atomicAdd(&acc->n, (unsigned long long)tn);
        atomicMin(&acc->minqs, qmin); atomicMax(&acc->maxqs, qmax);
    }
}

}  // namespace fx
