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
        for (int j = 0; j < GR_ROWS; ++j) v[j] = q[j * 64];
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

// FASTQ composition (fastq.c:715-753).  16 lanes per record, 4 records per wave; a lane
// reads 16 aligned bytes per step (256-byte window per record).  Sequence line: SWAR
// compare+popcount for 'A','C','G','T' (upper case only) and '\r' (ignored); every other
// byte is N.  Quality line: min / max byte, '\r' ignored.
__device__ __forceinline__ uint32_t valid16(int64_t pp, int64_t lo, int64_t hi) {
    int64_t a0 = lo - pp, a1 = hi - pp;
    a0 = a0 < 0 ? 0 : (a0 > 16 ? 16 : a0);
    a1 = a1 < 0 ? 0 : (a1 > 16 ? 16 : a1);
    return a1 > a0 ? (((1u << a1) - 1u) & ~((1u << a0) - 1u)) : 0u;
}
__global__ __launch_bounds__(BLOCK) void k_fastq_comp(const uint8_t *__restrict__ data, int64_t gbase, FqTab t,
                                                     int64_t n_seq_rows, int64_t n_rows, FastqAcc *acc) {
    const int lane = lane_id(), sub = lane & 15, grp = lane >> 4;
    const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
    uint32_t ca = 0, cc = 0, cg = 0, ct = 0, cn = 0;       // per lane, flushed per record batch (no overflow: <= 16 per step)
    unsigned long long ta = 0, tc = 0, tg = 0, tt = 0, tn = 0;
    int qmin = 104, qmax = 33;                             // fastq.c:667-668
    for (int64_t t0 = wave * 4; t0 < n_seq_rows; t0 += nwaves * 4) {
        const int64_t i = t0 + grp;
        if (i < n_seq_rows) {                              // line_num % 4 == 2
            const int64_t s = t.soff[i] - gbase, e = s + t.rlen[i];
            for (int64_t p = (s & ~15ll) + sub * 16; p < e; p += 256) {
                const uint4 v = *reinterpret_cast<const uint4 *>(data + p);
                const uint32_t ok = valid16(p, s, e);
                const uint32_t ma = eq_mask16(v, 0x41414141u) & ok, mc = eq_mask16(v, 0x43434343u) & ok;
                const uint32_t mg = eq_mask16(v, 0x47474747u) & ok, mt = eq_mask16(v, 0x54545454u) & ok;
                const uint32_t mr = eq_mask16(v, 0x0D0D0D0Du) & ok;
                ca += __popc(ma); cc += __popc(mc); cg += __popc(mg); ct += __popc(mt);
                cn += __popc(ok & ~(ma | mc | mg | mt | mr));
            }
        }
        if (i < n_rows) {                                  // line_num % 4 == 0
            const int64_t s = t.qoff[i] - gbase, e = s + t.qlen[i];
            for (int64_t p = (s & ~15ll) + sub * 16; p < e; p += 256) {
                const uint4 v = *reinterpret_cast<const uint4 *>(data + p);
                uint32_t ok = valid16(p, s, e) & ~eq_mask16(v, 0x0D0D0D0Du);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                while (ok) {
                    const int j = __ffs(ok) - 1;
                    ok &= ok - 1;
                    const int q = (int)(signed char)(w[j >> 2] >> ((j & 3) * 8));
                    qmin = q < qmin ? q : qmin; qmax = q > qmax ? q : qmax;
                }
            }
        }
        ta += ca; tc += cc; tg += cg; tt += ct; tn += cn;
        ca = cc = cg = ct = cn = 0;
    }
    ta = wave_sum64(ta); tc = wave_sum64(tc); tg = wave_sum64(tg); tt = wave_sum64(tt); tn = wave_sum64(tn);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        int a = __shfl_xor(qmin, d, 64), b = __shfl_xor(qmax, d, 64);
        qmin = a < qmin ? a : qmin; qmax = b > qmax ? b : qmax;
    }
    if (lane == 0) {
        if (ta) atomicAdd(&acc->a, ta); if (tc) atomicAdd(&acc->c, tc); if (tg) atomicAdd(&acc->g, tg);
        if (tt) atomicAdd(&acc->t, tt); if (tn) atomicAdd(&acc->n, tn);
        atomicMin(&acc->minqs, qmin); atomicMax(&acc->maxqs, qmax);
    }
}

}  // namespace fx
