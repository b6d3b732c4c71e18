// fx_fastq.hpp -- FASTQ index build for gfx950 (MI355X, wave64), on top of the granule scan.
//
// The reference runs a `line_num % 4` state machine over the lines of the file (fastq.c:89-149).
// A FASTQ record is "four lines", so the only global fact a byte range needs is the line number of
// its first newline.  The build reads the stream ONCE when its lines are not very short:
//
//   k_fastq_lines     a wave per 4 KiB granule (four in a row, the next one requested early): newline / space / CR
//        masks of the granule (the last two into LDS), the newline positions compacted, then one lane per newline writes a 32-bit LINE RECORD -- position in the granule,
//        "a CR precedes the newline", first space of the line (from its second byte; what a header line's name ends
//        at) -- into the granule's slot of FQL_CAP records, and the granule's newline count / first / last exactly
//        as k_span_scan<1> would.  4 bytes per line instead of the line.
//   k_gran_reduce<1> + k_gran_prefix     exclusive prefixes of the granule counts (shared with the FASTA build)
//   k_fastq_rows      a wave per four granules, their line records side by side in LDS, one lane per ROW: global line
//        index = loff + nl_prefix[g] + rank, and the lane goes through the four lines of record index >> 2 -- header end:
//        name_off / name_len / dlen / soff; sequence end: rlen; '+' line end: qoff; quality end: qlen -- so every store
//        covers consecutive rows.  It touches the stream only for a header line that began in an earlier granule (its
//        name may end there).
//   k_fastq_emit      the same rows from the BYTES of a granule (re-read into LDS): for the granules on a list --
//        those with more than FQL_CAP lines and the partial last one -- or, when the sampled line density says most
//        granules would overflow (lines shorter than ~40 bytes), for all of them after a count-only k_span_scan<1>:
//        the two-read build.
//   No line table, no record-level gathers from memory; one atomic per workgroup that has an overflowing granule.
//
// 20 M reads of 150 bp (7 GB): 1.37 + 0.06 (prefixes) + 0.49 ms = 2.2 ms with k_fastq_stats; two reads: 1.05 + 0.06 + 1.55.
// k_fastq_lines reads the stream exactly once (PMC: 1.000 x) at 5.1 TB/s with the VALU ~60 % busy (506 instructions per
// granule, half of them the three exact byte masks); a copy of the granule in LDS and a byte scan of the lines that start
// with '@' instead of the space / CR maps was slower (1.49 ms).
//
// (Measured dead ends, both correct and both slower.  (1) One kernel, the line numbers by a decoupled look-back over
// per-granule counts published by the other waves: 213 ms -- 1.7 M four-KiB tiles, 8 k in flight, every wave walks
// back over thousands of "counted, prefix not yet known" words.  (2) The same with 64 KiB tiles -- a persistent grid
// of 16-wave workgroups, tile in LDS, one 64-bit look-back word per tile polled 256 tiles per round trip, next tile
// requested early: 4.1 ms at best against 2.6 ms for two reads.  Load, count, barrier, look-back, barrier, emit is
// ~10 us per tile and workgroup, and the registers of the emit stage leave room for one such workgroup per CU.)
//
// Byte-range shards use the same kernels: the count pass is fx_fastq_scan, one
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
    int qfix, pad1;          // qfix: waves of k_fastq_comp that met a '\r' inside a quality line (fastq.c:733-737: k_fastq_qual_walk then redoes that half)
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

// A wave's granule: its 4 KiB into registers (non-temporal: streamed once) with the virtual end-of-stream newline
// of an unterminated last line put in (fastq.c:148).
__device__ __forceinline__ void fq_load_granule(const ScanCtx &x, int is_last, int64_t sbase, uint4 (&v)[GR_ROWS]) {
    const int lane = lane_id();
    if (sbase + GRAN <= x.n) {
        const uint4 *q = reinterpret_cast<const uint4 *>(x.data + sbase + lane * CHUNK);
#pragma unroll
        for (int j = 0; j < GR_ROWS; ++j) {
            v[j].x = __builtin_nontemporal_load(&q[j * 64].x); v[j].y = __builtin_nontemporal_load(&q[j * 64].y);
            v[j].z = __builtin_nontemporal_load(&q[j * 64].z); v[j].w = __builtin_nontemporal_load(&q[j * 64].w);
        }
    } else {
#pragma unroll
        for (int j = 0; j < GR_ROWS; ++j) v[j] = load16(x.data, sbase + j * 1024 + lane * CHUNK, x.n);
        if (is_last) {
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
}

// granule -> LDS, newline masks of the lane's chunks; -> newlines in the granule
__device__ __forceinline__ uint32_t fq_masks(const uint4 (&v)[GR_ROWS], uint4 *sd, uint32_t (&nlm)[GR_ROWS]) {
    const int lane = lane_id();
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < GR_ROWS; ++j) {
        sd[j * 64 + lane] = v[j];
        nlm[j] = eq_mask16(v[j], 0x0A0A0A0Au);
        c += __popc(nlm[j]);
    }
    return wave_sum(c);
}

// ONE LANE PER NEWLINE of the wave's granule (already in LDS at sb): I0 = global line index of its first newline,
// q_carry = global offset of the newline before it (-1: none).
template <int POSCAP>
__device__ __forceinline__ void fq_emit_rows(const ScanCtx &x, int prev_byte, const FqOwn &own, const FqTab &t, const uint8_t *sb,
                                             uint16_t *spos, const uint32_t (&nlm)[GR_ROWS],
                                             uint32_t M, int64_t sbase, int64_t I0, int64_t q_carry) {
    const int lane = lane_id();
    uint32_t ex[GR_ROWS];                                  // rank of each chunk's first newline in the granule
    {
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < GR_ROWS; ++j) {
            const uint32_t c = __popc(nlm[j]);
            const uint32_t inc = wave_incl_scan(c);
            ex[j] = acc + inc - c;
            acc += (uint32_t)__shfl((int)inc, 63, 64);
        }
    }
    const int64_t gs = x.gbase + sbase;                    // global offset of the granule
    for (uint32_t lo = 0; lo < M; lo += POSCAP) {
        // ---- compact the newline positions of ranks [lo, lo + POSCAP) into LDS
#pragma unroll
        for (int j = 0; j < GR_ROWS; ++j) {
            uint32_t m = nlm[j], r = ex[j];
            while (m) {
                const int k = __ffs(m) - 1;
                m &= m - 1;
                if (r - lo < (uint32_t)POSCAP) spos[r - lo] = (uint16_t)(j * 1024 + lane * CHUNK + k);
                ++r;
            }
        }
        const uint32_t cnt = (M - lo < (uint32_t)POSCAP) ? M - lo : (uint32_t)POSCAP;
        // ---- one lane per newline
        for (uint32_t t0 = 0; t0 < cnt; t0 += 64) {
            const uint32_t i = t0 + lane;
            if (i >= cnt) continue;
            const int lp = spos[i];
            const int64_t p = gs + lp;
            const int64_t q = i ? gs + spos[i - 1] : q_carry;         // previous newline (-1: none)
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
        q_carry = gs + spos[cnt - 1];
    }
}

// list == null: every granule of the shard; else the granules list[0 .. nlist)
__global__ __launch_bounds__(BLOCK) void k_fastq_emit(ScanCtx x, int prev_byte, int is_last, FqOwn own, FqTab t,
                                                     const uint32_t *__restrict__ list, int64_t nlist) {
    __shared__ uint4 s_data[BLOCK / 64][GRAN / 16];
    __shared__ uint16_t s_pos[BLOCK / 64][FQ_POSCAP];
    const int w = threadIdx.x >> 6;
    int64_t g = (int64_t)blockIdx.x * (BLOCK / 64) + w;
    if (g >= (list ? nlist : x.ngran)) return;             // waves are independent: no workgroup barrier below
    if (list) g = list[g];
    const int64_t sbase = g * (int64_t)GRAN;
    uint4 v[GR_ROWS];
    fq_load_granule(x, is_last, sbase, v);
    uint32_t nlm[GR_ROWS];                                 // newline mask of the lane's chunks
    const uint32_t M = fq_masks(v, &s_data[w][0], nlm);
    if (!M) return;
    int64_t q_carry = x.prevnl[g];                         // newline before the granule's first one
    if (q_carry < 0) q_carry = own.prev_nl;
    fq_emit_rows<FQ_POSCAP>(x, prev_byte, own, t, reinterpret_cast<const uint8_t *>(&s_data[w][0]), &s_pos[w][0], nlm, M, sbase,
                            own.loff + x.nl_prefix[g], q_carry);
}

// ---- line records
constexpr int FQL_CAP = 128;                   // line records per granule slot (lines of 32 bytes on average fill it)
constexpr uint32_t FQL_CR = 1u << 12, FQL_HAS = 1u << 13;     // record: pos (12 bits) | CR | has-space | space pos << 14

// first set bit of a 4096-bit LDS mask at a position in [a, b), or -1  (a < b)
__device__ __forceinline__ int fq_first_bit(const uint64_t *m, int a, int b) {
    const int w1 = (b - 1) >> 6;
    for (int wd = a >> 6; wd <= w1; ++wd) {
        uint64_t v = m[wd];
        if (wd == (a >> 6)) v &= ~0ull << (a & 63);
        if (wd == w1 && (b & 63)) v &= (1ull << (b & 63)) - 1ull;
        if (v) return wd * 64 + __ffsll((long long)v) - 1;
    }
    return -1;
}

// the count pass of the one-read build: granules [0, g_end) (all but the partial last one).  A wave takes FQL_G
// consecutive granules and asks for the next one before it works on the one it has.
#ifndef FX_FQL_G
#define FX_FQL_G 4
#endif
constexpr int FQL_G = FX_FQL_G;
// ... and k_fastq_rows takes FQR_G of them per wave: the count pass fills ONE record slot per FQR_G granules
#ifndef FX_FQR_G
#define FX_FQR_G 4
#endif
constexpr int FQR_G = FX_FQR_G;
static_assert(FQR_G <= 8, "a staged line record has three bits for the granule of the wave it came from");
static_assert(FQL_G % FQR_G == 0, "the count pass fills one record slot per FQR_G granules: a wave of k_fastq_rows reads one of them");

// The line records of a granule whose newline positions stand compacted in spos[0, M): one lane per line -- the CR bit from the
// granule's map of '\r' bytes, the first space of the line by a word-wise bit search from the line's second byte.
// (Round 4 tried to READ both from the stream instead -- one byte in front of every newline, the head of every header line, which
// the one-read build can tell from its guess -- and drop the two maps from the path of every byte: 150 vector instructions less
// per granule and 3.6 ms MORE for C3's 34.8 GB; the loads sit at the end of a granule's work with nothing to hide them behind.)
__device__ __forceinline__ void fq_line_records(const uint8_t *__restrict__ data, int64_t sbase, int prev_byte, const uint16_t *spos, uint32_t M,
                                                const uint64_t *spm, const uint16_t *crm, uint32_t *__restrict__ slot, int lane) {
    for (uint32_t i = lane; i < M; i += 64) {
        const int lp = spos[i];
        const int ql = i ? (int)spos[i - 1] : -1;                    // previous newline of the granule
        int cr;
        if (lp) cr = (crm[(lp - 1) >> 4] >> ((lp - 1) & 15)) & 1;
        else    cr = (sbase ? data[sbase - 1] : prev_byte) == '\r';
        // a header's name ends at the first space from the line's SECOND byte on (fastq.c:112-117).  The first
        // line of the granule may have begun earlier: every byte of it that lies here is searched, k_fastq_rows sorts it out
        const int from = i ? ql + 2 : 0;
        const int sp = from < lp ? fq_first_bit(spm, from, lp) : -1;
        slot[i] = (uint32_t)lp | (cr ? FQL_CR : 0u) | (sp >= 0 ? FQL_HAS | ((uint32_t)sp << 14) : 0u);
    }
}

// the exact map of the '\r' bytes of a row -- most files have none at all: computed only for a row in which some lane holds one
// ((y - 0x01..) & ~y & 0x80.. is non-zero exactly when a byte of y is zero: 13 instructions instead of 27; the kernels are VALU-bound)
__device__ __forceinline__ uint16_t fq_cr_mask(const uint4 &v) {
    const uint32_t y0 = v.x ^ 0x0D0D0D0Du, y1 = v.y ^ 0x0D0D0D0Du, y2 = v.z ^ 0x0D0D0D0Du, y3 = v.w ^ 0x0D0D0D0Du;
    uint32_t any_cr = (y0 - 0x01010101u) & ~y0;
    any_cr = (uint32_t)__builtin_amdgcn_bitop3_b32(any_cr, y1 - 0x01010101u, y1, 0xF4);      // a | (b & ~c)
    any_cr = (uint32_t)__builtin_amdgcn_bitop3_b32(any_cr, y2 - 0x01010101u, y2, 0xF4);
    any_cr = (uint32_t)__builtin_amdgcn_bitop3_b32(any_cr, y3 - 0x01010101u, y3, 0xF4);
    return __ballot((any_cr & 0x80808080u) != 0) ? (uint16_t)eq_mask16(v, 0x0D0D0D0Du) : (uint16_t)0;
}

// first / last newline of a granule with too many lines for its slot (the others read them from the compacted positions)
__device__ __forceinline__ void fq_first_last(const uint32_t (&nlm)[GR_ROWS], int lane, int &first, int &last) {
    first = GRAN; last = -1;
#pragma unroll
    for (int j = 0; j < GR_ROWS; ++j)
        if (nlm[j]) {
            const int cb = j * 1024 + lane * CHUNK;
            if (first == GRAN) first = cb + __ffs(nlm[j]) - 1;
            last = cb + 31 - __clz(nlm[j]);
        }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const int f = __shfl_xor(first, d, 64), l = __shfl_xor(last, d, 64);
        first = f < first ? f : first; last = l > last ? l : last;
    }
}

__global__ __launch_bounds__(BLOCK) void k_fastq_lines(const uint8_t *__restrict__ data, int64_t n, int prev_byte, int64_t g_end,
                                                      GranPk *__restrict__ out, uint32_t *__restrict__ recs, GranList ovl) {
    __shared__ __attribute__((aligned(16))) uint16_t s_sp[BLOCK / 64][GRAN / CHUNK], s_cr[BLOCK / 64][GRAN / CHUNK];   // bit k of word c <-> byte 16 c + k
    __shared__ uint16_t s_pos[BLOCK / 64][FQL_CAP];
    __shared__ uint32_t ov_n, ov_done, ov_g[(BLOCK / 64) * FQL_G];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    if (threadIdx.x == 0) { ov_n = 0; ov_done = 0; }
    __syncthreads();
    const int64_t gw = ((int64_t)blockIdx.x * (BLOCK / 64) + w) * FQL_G;
    uint4 v[GR_ROWS];
    if (gw < g_end) granule_load<true>(v, data, n, 0, gw);
    uint32_t written = 0;                  // records of the slot so far (a slot per FQR_G granules, the records one after the other)
    for (int kk = 0; kk < FQL_G; ++kk) {
        const int64_t g = gw + kk;
        if (g >= g_end) break;
        if (kk % FQR_G == 0) written = 0;
        uint32_t *const rslot = recs + (g / FQR_G) * (int64_t)(FQR_G * FQL_CAP);
        const int64_t sbase = g * (int64_t)GRAN;
        uint32_t nlm[GR_ROWS], ex[GR_ROWS], M = 0;
#pragma unroll
        for (int j = 0; j < GR_ROWS; ++j) {
            nlm[j] = eq_mask16(v[j], 0x0A0A0A0Au);
            s_sp[w][j * 64 + lane] = (uint16_t)eq_mask16(v[j], 0x20202020u);
            s_cr[w][j * 64 + lane] = fq_cr_mask(v[j]);
            const uint32_t cj = (uint32_t)__popc(nlm[j]);
            const uint32_t inc = wave_incl_scan(cj);
            ex[j] = M + inc - cj;                              // newlines of the granule in front of this chunk
            M += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        if (kk + 1 < FQL_G && g + 1 < g_end) granule_load<true>(v, data, n, 0, g + 1);       // the next granule is on its way
        int first = GRAN, last = -1;
        const bool over = M > (uint32_t)FQL_CAP;              // more lines than a slot holds: k_fastq_emit reads the granule again
        if (over) {
            fq_first_last(nlm, lane, first, last);
            if (lane == 0) ov_g[atomicAdd(&ov_n, 1u)] = (uint32_t)g;
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
                        rslot + written, lane);
        written += M;
    }
    // ---- the overflowing granules of the workgroup: one append (the last wave to arrive does it)
    if (lane == 0) {
        __threadfence_block();
        if (atomicAdd(&ov_done, 1u) == (BLOCK / 64) - 1) {
            const uint32_t cnt = ov_n;
            if (cnt) {
                const uint32_t base = atomicAdd(ovl.count, cnt);
                for (uint32_t k = 0; k < cnt; ++k) ovl.g[base + k] = ov_g[k];
            }
        }
    }
}

// rows from the line records: granules [0, g_end) that did not overflow.  A wave takes FQR_G consecutive granules and
// asks for their summaries, then for their records, before it computes anything: one wave per granule spent its time
// waiting for two dependent round trips (0.73 ms for 1.7 M granules; the traffic is worth 0.25 ms).

// The fields the line record r of line idx determines (line i of its granule, which begins at global offset gs; q =
// global offset of the newline before it), into row idx >> 2 of the table when the shard owns it.
template <bool NT = false>
__device__ __forceinline__ void fq_row_of_line(const ScanCtx &x, const FqOwn &own, const FqTab &t, int64_t gs, bool first_of_granule, uint32_t r,
                                               int64_t q, int64_t idx) {
    const int64_t p = gs + (r & 0xFFFu);
    const int64_t row = (idx >> 2) - own.k_first;
    if (row < 0 || row >= own.nrows) return;
    const int64_t len = p - q - 1;                                        // bytes of the line that ends at p
    const int cr = (len > 0 && (r & FQL_CR)) ? 1 : 0;
    switch ((int)(idx & 3)) {
    case 0: {                                                             // header line  (fastq.c:99-117)
        int64_t nlen = len - 1;
        if (nlen > 0 && cr) --nlen;                                       // fastq.c:107-109
        const int64_t nb = q + 2, ne = nb + nlen;                         // the name's first byte, the end of the search
        int64_t sp = (r & FQL_HAS) ? gs + ((r >> 14) & 0xFFFu) : ne;      // first space among the line's bytes in this granule
        if (first_of_granule && nb < gs) {                                // the line began in an earlier granule: those bytes first
            const int64_t e = ne < gs ? ne : gs;
            const int64_t hit = x.gbase + first_space(x.data, nb - x.gbase, e - x.gbase);
            if (hit < e) sp = hit;
        } else if (first_of_granule && sp < nb) {                         // it begins exactly here and its FIRST byte is a space
            sp = x.gbase + first_space(x.data, nb - x.gbase, ne - x.gbase);
        }
        const int64_t hit = sp < ne ? sp : ne;
        if (NT) {                                                          // (the table is not read again by this kernel: past the caches)
            __builtin_nontemporal_store(nb, &t.name_off[row]); __builtin_nontemporal_store((int32_t)(hit - nb), &t.name_len[row]);
            __builtin_nontemporal_store((int32_t)len, &t.dlen[row]); __builtin_nontemporal_store(p + 1, &t.soff[row]);
            break;
        }
        t.name_off[row] = nb; t.name_len[row] = (int32_t)(hit - nb); t.dlen[row] = (int32_t)len;   // fastq.c:103: '@' and '\r' included
        t.soff[row] = p + 1;                                              // fastq.c:122
        break;
    }
    case 1: if (NT) __builtin_nontemporal_store(len - cr, &t.rlen[row]); else t.rlen[row] = len - cr; break;                                // fastq.c:124-128
    case 2: if (NT) __builtin_nontemporal_store(p + 1, &t.qoff[row]); else t.qoff[row] = p + 1; break;                                   // fastq.c:133
    default: if (NT) __builtin_nontemporal_store((int32_t)(len - cr), &t.qlen[row]); else t.qlen[row] = (int32_t)(len - cr); break;      // quality line, trailing CR dropped (fastq.c:734-737)
    }
}

// ONE LANE PER ROW.  The line records of the wave's granules are consecutive lines of the file (when none of the
// granules overflowed): they are put side by side in LDS (bits 26-28 of a staged record: which of the wave's granules),
// and lane k takes row (first row of the wave) + k -- its header, sequence, '+' and quality line one after the other,
// each from its record and the one before.  Every lane is in the same phase at the same time and every store covers
// consecutive rows.  (One lane per LINE: four phases side by side in every wave and every fourth lane storing,
// 0.54 ms for 20 M reads where the traffic is worth 0.2; the same with the fields put together in LDS first: 0.67 ms;
// this form 0.48-0.50 ms, eight granules per wave 0.77, two 0.55; the x-th eighth of the granules to XCD x, so that
// neighbouring rows go through one L2: no difference.)
__global__ __launch_bounds__(BLOCK) void k_fastq_rows(ScanCtx x, FqOwn own, FqTab t, const uint32_t *__restrict__ recs, int64_t g_end) {
    __shared__ uint32_t s_rec[BLOCK / 64][FQR_G * FQL_CAP];
    const int lane = lane_id();
    const int64_t g0 = ((int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) * FQR_G;
    if (g0 >= g_end) return;                                               // waves are independent: no workgroup barrier below
    uint32_t M[FQR_G];
    int64_t I0[FQR_G], q0[FQR_G];
    bool whole = true;                                                     // no granule of the wave left to k_fastq_emit
    // The line records of the wave's granules stand one after the other in the slot of their RUN (the count pass, whose waves take
    // the same runs of FQL_G granules, appends them there): for reads of 150 bases that is ~47 records -- ONE 256-byte request
    // (a slot per granule, round 3: four requests of 256 bytes at a stride of 512, a dozen records in each).  They are asked for
    // together with the summaries that say how many there are: one round trip instead of two.
    const uint32_t *rslot = recs + (g0 / FQR_G) * (int64_t)(FQR_G * FQL_CAP);
    const uint32_t rr = rslot[lane];
#pragma unroll
    for (int k = 0; k < FQR_G; ++k) {
        M[k] = 0; I0[k] = 0; q0[k] = -1;
        if (g0 + k < g_end) { M[k] = x.go[g0 + k].nh & 0xFFFFu; I0[k] = x.nl_prefix[g0 + k]; q0[k] = x.prevnl[g0 + k]; }
        if (M[k] > (uint32_t)FQL_CAP) { M[k] = 0; whole = false; }        // overflowed (no records of it in the slot): k_fastq_emit reads that granule again
    }
    uint32_t cum[FQR_G + 1];
    cum[0] = 0;
#pragma unroll
    for (int k = 0; k < FQR_G; ++k) cum[k + 1] = cum[k] + M[k];
    if (!whole) {                                                          // one lane per line, granule by granule
#pragma unroll
        for (int k = 0; k < FQR_G; ++k) {
            const uint32_t *slot = rslot + cum[k];
            const int64_t gs = x.gbase + (g0 + k) * (int64_t)GRAN;
            const int64_t qq = q0[k] < 0 ? own.prev_nl : q0[k];
            for (uint32_t i = lane; i < M[k]; i += 64)
                fq_row_of_line(x, own, t, gs, i == 0, slot[i], i ? gs + (slot[i - 1] & 0xFFFu) : qq, own.loff + I0[k] + i);
        }
        return;
    }
    uint32_t *sr = s_rec[threadIdx.x >> 6];
    for (uint32_t i = lane; i < cum[FQR_G]; i += 64) {                      // (bits 26-28 of a staged record: which of the wave's granules)
        const uint32_t r = i < 64u ? rr : rslot[i];
        uint32_t k = 0;
#pragma unroll
        for (int kk = 1; kk < FQR_G; ++kk) k += i >= cum[kk] ? 1u : 0u;
        sr[i] = r | (k << 26);
    }
    const uint32_t Mtot = cum[FQR_G];
    if (!Mtot) return;
    const int64_t L0 = own.loff + I0[0];                                   // global index of the wave's first line
    const int64_t gs0 = x.gbase + g0 * (int64_t)GRAN;
    const int64_t qq0 = q0[0] < 0 ? own.prev_nl : q0[0];                   // the newline before the wave's first line
    const int64_t rb = L0 >> 2, re = (L0 + Mtot - 1) >> 2;                 // rows the wave's lines touch
    for (int64_t row = rb + lane; row <= re; row += 64) {
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int64_t jj = 4 * row + ph - L0;
            if (jj < 0 || jj >= (int64_t)Mtot) continue;
            const uint32_t j = (uint32_t)jj;
            const uint32_t r = sr[j];
            const uint32_t k = (r >> 26) & 7u;
            const int64_t gs = gs0 + (int64_t)k * GRAN;
            int64_t q = qq0;
            if (j) { const uint32_t rp = sr[j - 1]; q = gs0 + (int64_t)((rp >> 26) & 7u) * GRAN + (rp & 0xFFFu); }
            bool first_of_granule = false;
#pragma unroll
            for (int kk = 0; kk < FQR_G; ++kk) first_of_granule |= k == (uint32_t)kk && j == cum[kk];
            fq_row_of_line(x, own, t, gs, first_of_granule, r & 0x03FFFFFFu, q, L0 + j);
        }
    }
}

// ONE LANE PER ROW, A WORKGROUP PER G GRANULES (round 5).  k_fastq_rows gives a wave FQR_G = 4 granules: ~47 lines, ~12 rows -- 12 of
// its 64 lanes work in the row loop and every column store is a 48-96-byte piece of a 128-byte line (2.3-2.5 ms for C3 where the
// traffic is worth 0.7).  Here the four waves of a workgroup stage the records of G / 4 runs side by side in LDS, then thread k
// takes row (first row of the workgroup) + k: ~190 of 256 lanes busy for G = 64 and reads of 150 bases, column stores of 1.5 KB.
// As in k_fastq_rows a wave asks for the first 64 records of its runs TOGETHER with the granule summaries that say how many of
// them count (every wave reads all G summaries -- lane = granule -- and scans them itself: no barrier in front of the staging).
// Bits 26-31 of a staged record: which of the workgroup's granules.  G is chosen by the launch from the stream's line density
// (FQW_CAP staged records); a workgroup with more lines than that, or with an overflowed granule (its lines are k_fastq_emit's),
// walks its granules one lane per line, as k_fastq_rows does.
#ifndef FX_FQW_CAP
#define FX_FQW_CAP 4096
#endif
constexpr int FQW_CAP = FX_FQW_CAP;
// Round 6, measured and left at ONE group: FX_FQW_GROUPS groups of G granules per workgroup, one after the other, the loads of ALL
// its groups (summaries, first 64 records of every run) issued before the first group is staged, so that the second group's round
// trip to memory runs under the first group's staging and rows; one barrier per group + one between groups (the
// staged records of a group are overwritten by the next).  C3, same box, k_fastq_rows: 1 group 1.52 ms, 2 groups 2.27, 3 groups
// 2.36 (tools/fq_build_bench.py 1e8): half as many workgroups, each living twice as long with a barrier more -- what the waves wait
// for is not the round trip of their loads (the persistent-grid form of round 4 had said the same).
#ifndef FX_FQW_GROUPS
#define FX_FQW_GROUPS 1
#endif
template <int G> struct FqwPre {                                           // what a wave asks for up front, per group
    uint32_t rr[G / FQR_G / (BLOCK / 64)];
    uint32_t M;
    bool over;
    int64_t L0, qp;
    int ng;
};
template <int G>
__device__ __forceinline__ void fqw_load(const ScanCtx &x, const FqOwn &own, const uint32_t *__restrict__ recs, int64_t g_end, int64_t g0, int lane, int w, FqwPre<G> &p) {
    constexpr int RPW = G / FQR_G / (BLOCK / 64);
    p.ng = g0 < g_end ? (int)(g_end - g0 < G ? g_end - g0 : G) : 0;
    p.M = 0; p.over = false; p.L0 = 0; p.qp = -1;
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int k0 = (w * RPW + q) * FQR_G;
        p.rr[q] = k0 < p.ng ? recs[((g0 + k0) / FQR_G) * (int64_t)(FQR_G * FQL_CAP) + lane] : 0u;
    }
    if (p.ng == 0) return;
    if (lane < p.ng) { p.M = x.go[g0 + lane].nh & 0xFFFFu; if (p.M > (uint32_t)FQL_CAP) { p.M = 0; p.over = true; } }
    p.L0 = own.loff + x.nl_prefix[g0];                                     // global index of the group's first line (asked for with the rest)
    p.qp = x.prevnl[g0];
}
template <int G, bool NT>
__device__ __forceinline__ void fqw_group(const ScanCtx &x, const FqOwn &own, const FqTab &t, const uint32_t *__restrict__ recs, int64_t g0,
                                          const FqwPre<G> &p, uint32_t *s_rec, uint32_t *s_cum, int tid, int lane, int w) {
    constexpr int RPW = G / FQR_G / (BLOCK / 64);
    const int ng = p.ng;
    if (ng == 0) return;                                                   // (the same for every thread of the workgroup, as every branch below that leaves early)
    const uint32_t M = p.M;
    const int64_t L0 = p.L0, qp = p.qp;
    const unsigned long long ob = __ballot(p.over);
    const uint32_t incl = wave_incl_scan(M), excl = incl - M;              // lines of the group's granules in front of granule `lane`
    const uint32_t Mtot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (ob || Mtot > (uint32_t)FQW_CAP) {                                  // one lane per line, granule by granule (the records of a run
        for (int k = w; k < ng; k += BLOCK / 64) {                          // stand one after the other in its slot, overflowed granules left out)
            const uint32_t Mk = (uint32_t)__shfl((int)M, k, 64);
            if (!Mk) continue;
            const int64_t g = g0 + k;
            const uint32_t in_run = (uint32_t)__shfl((int)excl, k, 64) - (uint32_t)__shfl((int)excl, k - k % FQR_G, 64);
            const uint32_t *slot = recs + (g / FQR_G) * (int64_t)(FQR_G * FQL_CAP) + in_run;
            const int64_t gs = x.gbase + g * (int64_t)GRAN;
            const int64_t q0 = x.prevnl[g], qq = q0 < 0 ? own.prev_nl : q0, I0 = x.nl_prefix[g];
            for (uint32_t i = lane; i < Mk; i += 64)
                fq_row_of_line(x, own, t, gs, i == 0, slot[i], i ? gs + (slot[i - 1] & 0xFFFu) : qq, own.loff + I0 + i);
        }
        return;
    }
    if (w == 0 && lane < G) s_cum[lane] = excl;
#pragma unroll
    for (int q = 0; q < RPW; ++q) {                                        // the records of a run: one after the other in the run's slot
        const int k0 = (w * RPW + q) * FQR_G;
        if (k0 >= ng) break;
        const int k1 = k0 + FQR_G < ng ? k0 + FQR_G : ng;
        const uint32_t base = (uint32_t)__shfl((int)excl, k0, 64), cnt = (uint32_t)__shfl((int)incl, k1 - 1, 64) - base;
        uint32_t c1 = (uint32_t)__shfl((int)excl, k0 + 1 < ng ? k0 + 1 : k0, 64), c2 = (uint32_t)__shfl((int)excl, k0 + 2 < ng ? k0 + 2 : k0, 64),
                 c3 = (uint32_t)__shfl((int)excl, k0 + 3 < ng ? k0 + 3 : k0, 64);
        if (k0 + 1 >= ng) c1 = ~0u;
        if (k0 + 2 >= ng) c2 = ~0u;
        if (k0 + 3 >= ng) c3 = ~0u;
        const uint32_t *rslot = recs + ((g0 + k0) / FQR_G) * (int64_t)(FQR_G * FQL_CAP);
        for (uint32_t i = lane; i < cnt; i += 64) {
            const uint32_t r = i < 64u ? p.rr[q] : rslot[i];
            const uint32_t at = base + i;
            const uint32_t k = (uint32_t)k0 + (at >= c1 ? 1u : 0u) + (at >= c2 ? 1u : 0u) + (at >= c3 ? 1u : 0u);
            s_rec[at] = (r & 0x03FFFFFFu) | (k << 26);
        }
    }
    static_assert(FQR_G == 4, "the staging above names the run's four granules");
    __syncthreads();
    if (!Mtot) return;
    const int64_t gs0 = x.gbase + g0 * (int64_t)GRAN;
    const int64_t qq0 = qp < 0 ? own.prev_nl : qp;                         // the newline before the group's first line
    const int64_t rb = L0 >> 2, re = (L0 + Mtot - 1) >> 2;                 // rows the group's lines touch
    for (int64_t row = rb + tid; row <= re; row += BLOCK) {
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int64_t jj = 4 * row + ph - L0;
            if (jj < 0 || jj >= (int64_t)Mtot) continue;
            const uint32_t j = (uint32_t)jj;
            const uint32_t r = s_rec[j];
            const uint32_t k = r >> 26;
            const int64_t gs = gs0 + (int64_t)k * GRAN;
            int64_t q = qq0;
            if (j) { const uint32_t rp = s_rec[j - 1]; q = gs0 + (int64_t)(rp >> 26) * GRAN + (rp & 0xFFFu); }
            fq_row_of_line<NT>(x, own, t, gs, j == s_cum[k], r & 0x03FFFFFFu, q, L0 + j);
        }
    }
}
template <int G, bool NT>
__global__ __launch_bounds__(BLOCK) void k_fastq_rows_wg(ScanCtx x, FqOwn own, FqTab t, const uint32_t *__restrict__ recs, int64_t g_end) {
    static_assert(G % (FQR_G * (BLOCK / 64)) == 0 && G <= 64, "whole runs per wave; six bits for the granule a staged record came from");
    __shared__ uint32_t s_rec[FQW_CAP];
    __shared__ uint32_t s_cum[G];
    const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    const int64_t g0 = (int64_t)blockIdx.x * (G * FX_FQW_GROUPS);
    if (g0 >= g_end) return;
    FqwPre<G> pre[FX_FQW_GROUPS];
#pragma unroll
    for (int grp = 0; grp < FX_FQW_GROUPS; ++grp) fqw_load<G>(x, own, recs, g_end, g0 + (int64_t)grp * G, lane, w, pre[grp]);
#pragma unroll
    for (int grp = 0; grp < FX_FQW_GROUPS; ++grp) {
        if (grp) __syncthreads();                                          // (the rows of the group before have read their staged records)
        fqw_group<G, NT>(x, own, t, recs, g0 + (int64_t)grp * G, pre[grp], s_rec, s_cum, tid, lane, w);
    }
}

// newlines in up to three 256 KiB windows of the stream (start, middle, end): the line density that decides
// between the two forms of the build (fastq_count).  out[0] += newlines, out[1] += bytes looked at, out[2] += "\r\n" pairs.
constexpr int64_t FQ_SAMPLE = 256 * 1024;
__global__ __launch_bounds__(BLOCK) void k_nl_sample(const uint8_t *__restrict__ data, int64_t n, unsigned long long *out) {
    const int64_t len = n < FQ_SAMPLE ? n : FQ_SAMPLE;
    int64_t lo = blockIdx.x == 0 ? 0 : (blockIdx.x == 1 ? (n / 2) & ~15ll : (n - len) & ~15ll);
    if (blockIdx.x > 0 && n <= (int64_t)(blockIdx.x + 1) * FQ_SAMPLE) return;      // short stream: the first window(s) cover it
    const int64_t hi = lo + len < n ? lo + len : n;
    uint32_t c = 0, crlf = 0;
    for (int64_t p = lo + (int64_t)threadIdx.x * CHUNK; p < hi; p += (int64_t)BLOCK * CHUNK) {
        const uint4 v = load16(data, p, n);
        uint32_t m = eq_mask16(v, 0x0A0A0A0Au);
        if (hi - p < CHUNK) m &= (1u << (hi - p)) - 1u;
        c += __popc(m);
        crlf += __popc((eq_mask16(v, 0x0D0D0D0Du) << 1) & m);     // "\r\n" inside a chunk: does the stream end its lines that way? (out[2])
    }
    c = wave_sum(c);
    crlf = wave_sum(crlf);
    if (lane_id() == 0 && crlf) atomicAdd(&out[2], (unsigned long long)crlf);
    if (lane_id() == 0) { atomicAdd(&out[0], (unsigned long long)c); }
    if (threadIdx.x == 0) atomicAdd(&out[1], (unsigned long long)(hi - lo));
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

// meta.maxlen / minlen again, after k_fastq_comp rewrote the qlen of rows with a '\r' inside their quality line
// (acc->maxlen / minlen reset by the host before the launch)
__global__ __launch_bounds__(BLOCK) void k_fastq_qlen_range(FqTab t, int64_t n_rows, FastqAcc *acc) {
    long long mx = 0, mn = 10000000000LL;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_rows; i += (int64_t)gridDim.x * BLOCK) {
        const long long ql = t.qlen[i];
        mx = ql > mx ? ql : mx; mn = ql < mn ? ql : mn;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const long long a = __shfl_xor(mx, d, 64), b = __shfl_xor(mn, d, 64);
        mx = a > mx ? a : mx; mn = b < mn ? b : mn;
    }
    if (lane_id() == 0) { atomicMax(&acc->maxlen, mx); atomicMin(&acc->minlen, mn); }
}

// FASTQ composition (fastq.c:715-753): A C G T (upper case only) over the sequence lines, every other byte but
// '\r' is N; min / max byte of the quality lines, '\r' ignored.  lpr lanes per record, 64 / lpr records per wave; a lane
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

// The straight path of an iteration has no branch and no partial piece: a lane whose piece would run past the end of its
// line reads the LAST 16 bytes of the line instead (so does a lane that has no piece at all, and a lane of a row past
// the end of the table reads the last row's).  Twice-read bytes do not change a minimum or a maximum; in the sequence
// line the first `ov` bytes of such a piece -- the ones an earlier lane has counted -- are turned into '\r' through a
// mask looked up in LDS.  It holds when 16 <= line length <= 16 * lpr for every record of the iteration; an iteration
// with any other record (shorter than a piece, longer than one step, no rows at all) takes the general path, piece by
// piece with its own loads.
// What bounds the kernel is the shape of its loads, not what it does with them: with the counting compiled out it takes
// the same time (20 M reads of 150 bases, 7 GB: 1.63 ms; with every address rounded down to 16 or to 64 bytes -- wrong
// answers, timing only -- 1.35 ms, which is also what k_fastq_lines needs to stream the whole file).  PMC: the texture
// addresser is busy 65 % of the time, 53 cycles per load instruction, and the L1 looks 40 tags up per instruction: 16
// bytes per lane at any alignment are handled lane by lane.  What helped was fewer of them: lanes per record from the
// mean read length (16 lanes for a 150-byte line: 2.26 ms; 10: 1.64), the rows of an iteration fetched once per wave
// instead of once per group (1.67 -> 1.53).  Occupancy (4 to 6 waves), two or three records per group and the depth of
// the software pipeline made no difference.
__global__ __launch_bounds__(BLOCK) void k_fastq_comp(const uint8_t *__restrict__ data, int64_t gbase, int64_t n_bytes, FqTab t,
                                                     int64_t n_seq_rows, int64_t n_rows, FastqAcc *acc, int lpr,
                                                     const uint8_t *__restrict__ safe) {
    __shared__ int fix_all[BLOCK / 64][8];                 // per wave: signed corrections of the class counts (slot = class bit index)
    __shared__ uint4 drop_tab[17];                         // [o]: 0x00 in the first o bytes of the piece, 0xFF in the others
    // lpr lanes per record (the host picks ceil(mean read length / 16): ten for 150-base reads, where sixteen would leave
    // four lanes in ten without a piece), 64 / lpr records side by side per wave, the lanes left over idle
    const int lane = lane_id(), grp = lane / lpr, sub = lane - grp * lpr, wv = threadIdx.x >> 6;
    const int ngrp = 64 / lpr, step = 16 * lpr;
    const bool live = grp < ngrp;
    int *fix = fix_all[wv];
    if (lane < 8) fix[lane] = 0;
    if (threadIdx.x < 17) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = (int)threadIdx.x - 4 * i;        // bytes of word i that are dropped
            w[i] = k >= 4 ? 0u : k <= 0 ? 0xFFFFFFFFu : ~((1u << (8 * k)) - 1u);
        }
        drop_tab[threadIdx.x] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();
    const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
    uint32_t ones = 0, twos = 0;                           // bit planes of the one-hot words: 1s and 2s per bit position
    uint32_t c4[6] = {0, 0, 0, 0, 0, 0};                   // per class: popcount of the carry-outs (each worth 4)
    int qmin = 104, qmax = 33;                             // fastq.c:667-668

    auto seq_count = [&](const uint32_t (&x)[4]) {         // 16 bytes of a sequence line ('\r' where there is nothing to count)
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
    auto seq_piece = [&](const uint4 &v, int64_t left) {   // general path: `left` bytes of the piece lie inside the line
        uint32_t x[4] = {v.x, v.y, v.z, v.w};
        if (left < 16) fq_keep_first(x, (int)left, 0x0D0D0D0Du);
        seq_count(x);
    };
    bool saw_cr = false;                                   // a '\r' INSIDE a quality line (the table's qlen leaves a trailing one out)
    auto qual_exact = [&](const uint4 &v, int keep) {      // fastq.c:733-737, byte by byte
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        for (int j = 0; j < keep; ++j) {
            const int q = (int)(signed char)(w[j >> 2] >> ((j & 3) * 8));
            if (q == 13) { saw_cr = true; continue; }
            qmin = q < qmin ? q : qmin; qmax = q > qmax ? q : qmax;
        }
    };
    // even / odd bytes of the four words, zero-extended to 16 bits (one v_perm each), into packed minima and maxima
    auto qual_minmax = [&](const uint32_t (&lo)[4], const uint32_t (&hi)[4], uint32_t &mn, uint32_t &mx) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mn = pk_min_u16(mn, pk_min_u16(__builtin_amdgcn_perm(0u, lo[k], 0x0C020C00u), __builtin_amdgcn_perm(0u, lo[k], 0x0C030C01u)));
            mx = pk_max_u16(mx, pk_max_u16(__builtin_amdgcn_perm(0u, hi[k], 0x0C020C00u), __builtin_amdgcn_perm(0u, hi[k], 0x0C030C01u)));
        }
    };
    auto qual_piece = [&](const uint4 &v, int64_t left) {  // general path
        uint32_t lo[4] = {v.x, v.y, v.z, v.w}, hi[4] = {v.x, v.y, v.z, v.w};
        const int keep = left < 16 ? (int)left : 16;
        if (keep < 16) { fq_keep_first(lo, keep, 0xFFFFFFFFu); fq_keep_first(hi, keep, 0u); }
        uint32_t mn = 0x00FF00FFu, mx = 0u;
        qual_minmax(lo, hi, mn, mx);
        const int cmin = (int)min(mn & 0xFFFFu, mn >> 16), cmax = (int)max(mx & 0xFFFFu, mx >> 16);
        if (cmin >= 33 && cmax < 128) { qmin = cmin < qmin ? cmin : qmin; qmax = cmax > qmax ? cmax : qmax; }
        else qual_exact(v, keep);                          // '\r' (skipped) or bytes outside the printable range
    };

    // A wave takes FQ_RPW = ngrp * FQ_U consecutive records per iteration (FQ_U per group of lpr lanes) and runs three
    // iterations deep: while iteration k is counted, the pieces of iteration k + 1 (2 * FQ_U loads per lane) and the
    // table rows of iteration k + 2 are on their way.  `safe` is a kernel argument: a pointer picked from a variable of
    // the module makes every one of these a flat load.
    constexpr int FQ_U = FX_FQ_U;
    const int FQ_RPW = ngrp * FQ_U;
    const int64_t stride = nwaves * FQ_RPW;
    int64_t i0 = wave * FQ_RPW + grp * FQ_U;               // first record of this group in the iteration being planned
    int64_t ic = i0;                                       // ... in the iteration being counted
    // table rows: lane j of the wave asks for row (first row of the iteration) + j, j < FQ_RPW -- four loads per
    // iteration, one or two cache lines each -- and the lanes of a group pick theirs up through ds_bpermute when they
    // plan.  (Every lane asking for its own group's row: 4 * FQ_U loads with 64 addresses each.)  The loads are
    // unconditional (index clamped) and nothing is computed from them until the next pass -- anything else makes every
    // one of them a round trip of its own.
    int64_t R_soff = 0, R_rlen = 0, R_qoff = 0;
    int32_t R_qlen = 0;
    const int64_t last_seq = n_seq_rows - 1, last_row = n_rows > 0 ? n_rows - 1 : 0;
    auto rows_request = [&]() {
        const int64_t ib = i0 - grp * FQ_U + lane;
        if (lane < FQ_RPW) {
            const int64_t is = ib < last_seq ? ib : last_seq, iq = ib < last_row ? ib : last_row;
            R_soff = t.soff[is]; R_rlen = t.rlen[is]; R_qoff = t.qoff[iq]; R_qlen = t.qlen[iq];
        }
    };
    struct Plan { int64_t as, aq; int ov; };               // where this lane's two pieces start; ov: bytes of the sequence piece to drop (bit 8: general path)
    auto plan = [&](Plan (&P)[FQ_U]) {                     // from the rows that have arrived
#pragma unroll
        for (int u = 0; u < FQ_U; ++u) {
            const int64_t i = i0 + u;
            const int src = live ? grp * FQ_U + u : 0;
            const int64_t r_soff_u = shfl64(R_soff, src), r_rlen_u = shfl64(R_rlen, src), r_qoff_u = shfl64(R_qoff, src);
            const int32_t r_qlen_u = __shfl(R_qlen, src, 64);
            const bool sv = live && i < n_seq_rows, qv = live && i < n_rows;
            const int rl = (int)(r_rlen_u < 4096 ? r_rlen_u : 4096), ql = r_qlen_u < 4096 ? r_qlen_u : 4096;
            const bool slow = (sv && (rl < 16 || rl > step)) || (qv && (ql < 16 || ql > step)) || n_rows <= 0 ||
                              rl < 16 || ql < 16;          // (a clamped row shorter than a piece: nothing valid to read in its place)
            const int os = sub * 16 + 16 - rl, oq = sub * 16 + 16 - ql;          // how far the piece would run past the end of the line
            const int ds = os > 0 ? os : 0, dq = oq > 0 ? oq : 0;
            P[u].as = r_soff_u - gbase + (sub * 16 - ds);
            P[u].aq = r_qoff_u - gbase + (sub * 16 - dq);
            P[u].ov = (sv ? (ds < 16 ? ds : 16) : 16) | (slow ? 256 : 0);
        }
    };
    auto load = [&](const Plan (&P)[FQ_U], uint4 (&S)[FQ_U], uint4 (&Q)[FQ_U]) {
#pragma unroll
        for (int u = 0; u < FQ_U; ++u) {
            const bool bad = (P[u].ov & 256) != 0;
            S[u] = *reinterpret_cast<const uint4_u *>(bad ? safe : data + P[u].as);
            Q[u] = *reinterpret_cast<const uint4_u *>(bad ? safe : data + P[u].aq);
        }
    };
    Plan cur[FQ_U], nxt[FQ_U];
    uint4 vs[FQ_U], vq[FQ_U], ns[FQ_U], nq[FQ_U];
    // Loads come back in the order they were asked for, so the order is: rows of k + 2, THEN pieces of k + 1 -- waiting
    // for those rows at the top of the next pass leaves the pieces in flight; the other way round every pass begins by
    // draining everything.
    rows_request();
    plan(cur);
    i0 += stride;
    rows_request();
    __builtin_amdgcn_sched_barrier(0);
    load(cur, vs, vq);
    for (int64_t t0 = wave * FQ_RPW; t0 < n_seq_rows; t0 += stride) {
        plan(nxt);
        i0 += stride;
        __builtin_amdgcn_sched_barrier(0);
        rows_request();
        __builtin_amdgcn_sched_barrier(0);
        load(nxt, ns, nq);
        __builtin_amdgcn_sched_barrier(0);
        bool slow = false;
#pragma unroll
        for (int u = 0; u < FQ_U; ++u) slow |= (cur[u].ov & 256) != 0;
        if (__builtin_expect(__ballot(slow) == 0ull, 1)) {
            uint32_t mn = 0x00FF00FFu, mx = 0u;
#pragma unroll
            for (int u = 0; u < FQ_U; ++u) {
                const uint4 m = drop_tab[cur[u].ov];
                const uint32_t x[4] = {(uint32_t)__builtin_amdgcn_bitop3_b32(m.x, vs[u].x, 0x0D0D0D0Du, 0xCA),      // m ? v : '\r'
                                       (uint32_t)__builtin_amdgcn_bitop3_b32(m.y, vs[u].y, 0x0D0D0D0Du, 0xCA),
                                       (uint32_t)__builtin_amdgcn_bitop3_b32(m.z, vs[u].z, 0x0D0D0D0Du, 0xCA),
                                       (uint32_t)__builtin_amdgcn_bitop3_b32(m.w, vs[u].w, 0x0D0D0D0Du, 0xCA)};
                seq_count(x);
                const uint32_t q[4] = {vq[u].x, vq[u].y, vq[u].z, vq[u].w};
                qual_minmax(q, q, mn, mx);
            }
            const int cmin = (int)min(mn & 0xFFFFu, mn >> 16), cmax = (int)max(mx & 0xFFFFu, mx >> 16);
            if (cmin >= 33 && cmax < 128) { qmin = cmin < qmin ? cmin : qmin; qmax = cmax > qmax ? cmax : qmax; }
            else {
#pragma unroll
                for (int u = 0; u < FQ_U; ++u) qual_exact(vq[u], 16);
            }
        } else {
            // the general path: the rows again, then piece by piece
#pragma unroll 1
            for (int u = 0; u < FQ_U; ++u) {
                const int64_t i = ic + u;
                if (live && i < n_seq_rows) {
                    const int64_t s_ = t.soff[i] - gbase, e = s_ + t.rlen[i];
                    for (int64_t p = s_ + sub * 16; p < e; p += step) seq_piece(fq_load16(data, p, n_bytes), e - p);
                }
                if (live && i < n_rows) {
                    const int64_t q_ = t.qoff[i] - gbase, e = q_ + t.qlen[i];
                    for (int64_t p = q_ + sub * 16; p < e; p += step) qual_piece(fq_load16(data, p, n_bytes), e - p);
                }
            }
        }
        ic += stride;
#pragma unroll
        for (int u = 0; u < FQ_U; ++u) { cur[u] = nxt[u]; vs[u] = ns[u]; vq[u] = nq[u]; }
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
    // With a '\r' inside a quality line the reference's loop shrinks line.l as it goes (fastq.c:733-737): the last bytes of
    // the line are never looked at and meta.minlen / maxlen see the shrunken length.  That does not belong in this loop
    // (it costs registers and a wave-wide test per iteration for something no real file has): the kernel only says that it
    // met one, and the host then lets k_fastq_qual_walk redo the quality half exactly.
    if (__ballot(saw_cr) && lane == 0) atomicAdd(&acc->qfix, 1);
}

// The quality half of the composition for a file with a '\r' inside some quality line, exactly as the reference's loop
// runs (fastq.c:733-745): one lane per read walks its quality line as it stands in the stream, from qoff to the newline --
// line.l shrinks by one at every '\r' met, bytes at or past the bound are not examined -- and leaves what is left of line.l
// in the row's qlen (k_fastq_qlen_range then gives the reference's minlen / maxlen).  Slow (byte loads, one lane per line)
// and only ever launched for such files; acc->minqs / maxqs reset by the host before.
__global__ __launch_bounds__(BLOCK) void k_fastq_qual_walk(const uint8_t *__restrict__ data, int64_t gbase, int64_t n_bytes, FqTab t,
                                                          int64_t n_rows, FastqAcc *acc) {
    int qmin = 104, qmax = 33;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_rows; i += (int64_t)gridDim.x * BLOCK) {
        const int64_t q0 = t.qoff[i] - gbase;
        int64_t l = 0;
        while (q0 + l < n_bytes && data[q0 + l] != 10) ++l;
        for (int64_t k = 0; k < l; ++k) {
            const int q = (int)(signed char)data[q0 + k];
            if (q == 13) { --l; continue; }
            qmin = q < qmin ? q : qmin; qmax = q > qmax ? q : qmax;
        }
        t.qlen[i] = (int32_t)l;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const int a = __shfl_xor(qmin, d, 64), b = __shfl_xor(qmax, d, 64);
        qmin = a < qmin ? a : qmin; qmax = b > qmax ? b : qmax;
    }
    if (lane_id() == 0) { atomicMin(&acc->minqs, qmin); atomicMax(&acc->maxqs, qmax); }
}

}  // namespace fx
