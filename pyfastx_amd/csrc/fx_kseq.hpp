// fx_kseq.hpp -- Fastx: kseq_read (kseq.c:138-179, ks_getuntil2 kseq.c:59-109) over the resident stream.
//
// pyfastx.Fastx (fastx.c:124-130) hands out whatever kseq_read finds: FASTA and FASTQ records in one file, sequence
// and quality strings over any number of lines, junk between records, white space kept inside the lines.  That parser
// is a byte-at-a-time state machine whose quality part depends on the LENGTH of the sequence part, so its records
// cannot be found from local evidence the way the index builders' records can (fx_spanscan.hpp, fx_fastq.hpp).  Here:
//
//   k_kq_count / k_kq_lines   one read of the stream: the offsets of all line ends (a line table)
//   k_kq_desc                 per line: start, length, first and last byte
//   k_kq_walk                 ONE workgroup walks the line table: three waves stream the descriptors into an LDS ring, one
//                             wave runs the state machine over them -- 64 lines per step where the lines allow it (16
//                             four-line FASTQ records, or a run of FASTA header / sequence lines: ballots and one wave
//                             scan), one line per step otherwise.  It writes the record table and, per line, where its
//                             bytes go in the concatenated sequence / quality strings.
//   k_kq_gather               records [first, first + count) -> their strings, one 16-lane group per line
//
// The walk is sequential in the number of 64-line steps, not in bytes: 10^8 lines take a few tenths of a second.
// tools/kseq_line_model.py is the executable model k_kq_walk transliterates (fuzzed against oracle/fx_oracle.c: fxo_kseq).
#pragma once
#include "fx_kernels.hpp"
#include "fx_spanscan.hpp"

namespace fx {

#define KQ_LD(x) __hip_atomic_load(&(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define KQ_ST(x, v) __hip_atomic_store(&(x), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)

constexpr int KQ_TILE = 4096;                 // bytes per wave step of the count / lines kernels
constexpr int KQ_RING = 4096;                 // line descriptors in the LDS ring (64 KiB)
constexpr int KQ_BATCH = 512;                 // lines per producer batch
constexpr int KQ_SLOTS = KQ_RING / KQ_BATCH;
constexpr int KQ_PRODUCERS = 3;              // (7 made the walk slower: the walker wave shares its SIMD with them)
constexpr int KQ_WALK_BLOCK = 64 * (1 + KQ_PRODUCERS);
constexpr uint32_t KQ_BIG = 1u << 25;         // a window with a line this long takes the one-line steps (32-bit wave scans)
constexpr int64_t KQ_POS = (1ll << 62) - 1;   // ldst: position in the low 62 bits, class above
constexpr int KQ_LONG = 1 << 16;              // lines longer than this are copied by k_kq_gather_long

enum { KQ_SEEK = 0, KQ_HDR = 1, KQ_SEQ = 2, KQ_QUAL = 3 };
enum { KQ_C_SEQ = 1, KQ_C_QUAL = 2 };
enum { KQ_F_FASTQ = 1, KQ_F_UNTOUCHED = 2, KQ_F_HDR_UNTERM = 4 };

struct alignas(16) KqRec {                    // one kseq_read that returned >= 0
    int64_t hdr_off;                          // first byte after the '>' / '@'
    int64_t hdr_line;                         // line that holds it; the sequence lines follow
    int64_t seq_len;                          // seq.l (= qual.l of a FASTQ record)
    int64_t seq_cum;                          // sum of seq_len over the records before this one
    uint32_t hdr_len;                         // to the end of the line ('\n' excluded, a '\r' included)
    uint32_t s_n;                             // lines between the header line and the one that ended the sequence
    uint32_t q_n;                             // quality lines read (FASTQ records)
    uint32_t flags;                           // KQ_F_*
};
static_assert(sizeof(KqRec) == 48, "KqRec layout");

// ------------------------------------------------------------------ line table
__global__ __launch_bounds__(BLOCK) void k_kq_count(const uint8_t *__restrict__ data, int64_t n, int64_t ntiles, int32_t *__restrict__ cnt,
                                                   unsigned long long *__restrict__ hdrchars) {
    const int lane = lane_id();
    const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
    uint32_t hc = 0;
    for (int64_t t = wave; t < ntiles; t += nwaves) {
        uint32_t c = 0;
#pragma unroll
        for (int r = 0; r < KQ_TILE / 1024; ++r) {
            const uint4 v = load16(data, t * KQ_TILE + r * 1024 + lane * CHUNK, n);
            c += __popc(eq_mask16(v, 0x0A0A0A0Au));
            hc += __popc(eq_mask16(v, 0x3E3E3E3Eu)) + __popc(eq_mask16(v, 0x40404040u));
        }
        c = wave_sum(c);
        if (lane == 0) cnt[t] = (int32_t)c;
    }
    hc = wave_sum(hc);
    if (lane == 0 && hc) atomicAdd(hdrchars, (unsigned long long)hc);
}

// nl[k] = offset of the k-th '\n'; the caller appends the end of the stream when the last line has none
__global__ __launch_bounds__(BLOCK) void k_kq_lines(const uint8_t *__restrict__ data, int64_t n, int64_t ntiles, const int64_t *__restrict__ off,
                                                   int64_t *__restrict__ nl, int64_t virt_at) {
    const int lane = lane_id();
    const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
    if (virt_at >= 0 && blockIdx.x == 0 && threadIdx.x == 0) nl[virt_at] = n;
    for (int64_t t = wave; t < ntiles; t += nwaves) {
        int64_t base = off[t];
#pragma unroll
        for (int r = 0; r < KQ_TILE / 1024; ++r) {
            const int64_t p = t * KQ_TILE + r * 1024 + lane * CHUNK;
            uint32_t m = eq_mask16(load16(data, p, n), 0x0A0A0A0Au);
            const uint32_t c = __popc(m), inc = wave_incl_scan(c);
            int64_t o = base + inc - c;
            while (m) {
                const int k = __ffs(m) - 1;
                m &= m - 1;
                nl[o++] = p + k;
            }
            base += (int64_t)__shfl((int)inc, 63, 64);
        }
    }
}

// desc[i] = { start (64 bit), length, first byte | last byte << 8 | (no '\n' behind it) << 16 }
__global__ __launch_bounds__(BLOCK) void k_kq_desc(const uint8_t *__restrict__ data, int64_t n, const int64_t *__restrict__ nl, int64_t L,
                                                  uint4 *__restrict__ desc, uint32_t *__restrict__ err) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < L; i += (int64_t)gridDim.x * BLOCK) {
        const int64_t start = i ? nl[i - 1] + 1 : 0, end = nl[i], len = end - start;
        if (len > 0xFFFFFFF0ll) atomicOr(err, 1u);
        uint32_t info = end == n ? 1u << 16 : 0u;
        if (len > 0) info |= (uint32_t)data[start] | ((uint32_t)data[end - 1] << 8);
        desc[i] = make_uint4((uint32_t)(start & 0xFFFFFFFFll), (uint32_t)(start >> 32), (uint32_t)len, info);
    }
}

// ------------------------------------------------------------------ the regular prefix, in parallel
// A file of four-line FASTQ records, or of FASTA records whose lines hold nothing kseq treats specially, needs no state
// machine: every line can be judged by itself, the records and string positions are prefix sums.  These passes take the
// file up to the first line that breaks the pattern (usually: all of it); k_kq_walk starts there with the state they
// leave behind (KqInit) -- at the end of the stream it only closes the last record.
constexpr uint32_t KQ_REG_MAX = 1u << 20;     // longer lines end the regular prefix (32-bit partial sums of the scans)
struct KqInit { long long j0, nrec, S, cur_off, cur_line, acc; uint32_t st, cur_len, cur_flags, pad; };

__device__ __forceinline__ uint32_t kq_len(const uint4 &d) { return d.z; }
__device__ __forceinline__ uint32_t kq_first(const uint4 &d) { return d.w & 0xFF; }
__device__ __forceinline__ uint32_t kq_last(const uint4 &d) { return (d.w >> 8) & 0xFF; }
__device__ __forceinline__ bool kq_unterm(const uint4 &d) { return (d.w >> 16) & 1; }
__device__ __forceinline__ bool kq_is_hdr(const uint4 &d) { return d.z >= 1 && (kq_first(d) == '>' || kq_first(d) == '@'); }
// what a line adds to an EMPTY string (ks_getuntil2, kseq.c:106: a trailing CR goes from a string longer than one byte)
__device__ __forceinline__ uint32_t kq_con0(const uint4 &d) { return d.z - ((kq_last(d) == 13 && d.z > 1) ? 1u : 0u); }

// first[0] = first line that is not what its place in a four-line record asks for; first[1] = first line a FASTA run
// stops at ('+' in front, a lone CR, no '\n' behind it).  Both preset to L by the caller.
__global__ __launch_bounds__(BLOCK) void k_kq_classify(const uint4 *__restrict__ desc, int64_t L, unsigned long long *__restrict__ first) {
    unsigned long long mq = ~0ull, ma = ~0ull;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < L; i += (int64_t)gridDim.x * BLOCK) {
        const uint4 d = desc[i];
        const uint32_t f = kq_first(d), len = d.z;
        const int role = (int)(i & 3);
        bool ok;
        if (role == 0) ok = kq_is_hdr(d);
        else if (role == 1) ok = len >= 1 && f != '>' && f != '@' && f != '+';
        else if (role == 2) ok = len >= 1 && f == '+' && !kq_unterm(d);
        else ok = kq_con0(d) == kq_con0(desc[i - 2]);
        if (len >= KQ_REG_MAX) ok = false;
        if (!ok && (unsigned long long)i < mq) mq = (unsigned long long)i;
        const bool stopper = (len >= 1 && f == '+') || (len == 1 && f == 13) || kq_unterm(d) || len >= KQ_REG_MAX;
        if (stopper && (unsigned long long)i < ma) ma = (unsigned long long)i;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const unsigned long long oq = (unsigned long long)shfl64((long long)mq, lane_id() ^ d), oa = (unsigned long long)shfl64((long long)ma, lane_id() ^ d);
        mq = oq < mq ? oq : mq; ma = oa < ma ? oa : ma;
    }
    if (lane_id() == 0) {
        if (mq != ~0ull) atomicMin(&first[0], mq);
        if (ma != ~0ull) atomicMin(&first[1], ma);
    }
}

// ---- FASTQ: records 0 .. R - 1 are lines 4r .. 4r + 3
__global__ __launch_bounds__(BLOCK) void k_kq_fq_cnt(const uint4 *__restrict__ desc, int64_t R, int32_t *__restrict__ cnt) {
    for (int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; r < R; r += (int64_t)gridDim.x * BLOCK) cnt[r] = (int32_t)kq_con0(desc[4 * r + 1]);
}
__global__ __launch_bounds__(BLOCK) void k_kq_fq_emit(const uint4 *__restrict__ desc, int64_t R, const int64_t *__restrict__ off, KqRec *__restrict__ recs,
                                                     int64_t *__restrict__ ldst, uint32_t *__restrict__ lcon, KqInit *__restrict__ init) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        KqInit t{4 * R, R, off[R], 0, 0, 0, (uint32_t)KQ_SEEK, 0, 0, 0};
        *init = t;
    }
    for (int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; r < R; r += (int64_t)gridDim.x * BLOCK) {
        const uint4 h = desc[4 * r];
        const uint32_t con = kq_con0(desc[4 * r + 1]);
        const int64_t cum = off[r];
        KqRec t;
        t.hdr_off = (int64_t)(((uint64_t)h.y << 32) | h.x) + 1; t.hdr_line = 4 * r; t.seq_len = con; t.seq_cum = cum;
        t.hdr_len = h.z - 1; t.s_n = 1; t.q_n = 1; t.flags = KQ_F_FASTQ;
        recs[r] = t;
        ldst[4 * r + 1] = cum | ((int64_t)KQ_C_SEQ << 62); lcon[4 * r + 1] = con;
        ldst[4 * r + 3] = cum | ((int64_t)KQ_C_QUAL << 62); lcon[4 * r + 3] = con;
    }
}

// ---- FASTA: lines 0 .. n - 1, line 0 a header; a record = a header line and the lines up to the next one
__global__ __launch_bounds__(BLOCK) void k_kq_fa_cnt(const uint4 *__restrict__ desc, int64_t n, int32_t *__restrict__ hflag, int32_t *__restrict__ con) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const uint4 d = desc[i];
        const bool H = kq_is_hdr(d);
        hflag[i] = H ? 1 : 0;
        con[i] = (H || d.z == 0) ? 0 : (int32_t)(d.z - (kq_last(d) == 13 ? 1u : 0u));
    }
}
__global__ __launch_bounds__(BLOCK) void k_kq_fa_lines(const uint4 *__restrict__ desc, int64_t n, const int64_t *__restrict__ hoff, const int64_t *__restrict__ coff,
                                                      const int32_t *__restrict__ con, int64_t *__restrict__ hpos, int64_t *__restrict__ ldst,
                                                      uint32_t *__restrict__ lcon) {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
        const uint4 d = desc[i];
        if (kq_is_hdr(d)) hpos[hoff[i]] = i;
        else if (d.z > 0) { ldst[i] = coff[i] | ((int64_t)KQ_C_SEQ << 62); lcon[i] = (uint32_t)con[i]; }
    }
}
__global__ __launch_bounds__(BLOCK) void k_kq_fa_recs(const uint4 *__restrict__ desc, int64_t n, int64_t nh, const int64_t *__restrict__ hpos,
                                                     const int64_t *__restrict__ coff, KqRec *__restrict__ recs, KqInit *__restrict__ init) {
    for (int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; r < nh; r += (int64_t)gridDim.x * BLOCK) {
        const int64_t i = hpos[r];
        const uint4 d = desc[i];
        const int64_t off = (int64_t)(((uint64_t)d.y << 32) | d.x) + 1;
        if (r + 1 < nh) {
            const int64_t i2 = hpos[r + 1];
            KqRec t;
            t.hdr_off = off; t.hdr_line = i; t.seq_len = coff[i2] - coff[i]; t.seq_cum = coff[i];
            t.hdr_len = d.z - 1; t.s_n = (uint32_t)(i2 - i - 1); t.q_n = 0; t.flags = 0;
            recs[r] = t;
        } else {                                            // the last header's record is still open: the walk closes it
            KqInit t{n, nh - 1, coff[i], off, i, coff[n] - coff[i], (uint32_t)KQ_SEQ, d.z - 1, 0, 0};
            *init = t;
        }
    }
}

// ------------------------------------------------------------------ the walk
struct KqOut { long long n_rec, code, seq_bytes, pad; };

__device__ __forceinline__ void kq_rec_store(KqRec *__restrict__ r, int64_t hdr_off, int64_t hdr_line, int64_t seq_len, int64_t seq_cum,
                                             uint32_t hdr_len, uint32_t s_n, uint32_t q_n, uint32_t flags) {
    KqRec t;
    t.hdr_off = hdr_off; t.hdr_line = hdr_line; t.seq_len = seq_len; t.seq_cum = seq_cum;
    t.hdr_len = hdr_len; t.s_n = s_n; t.q_n = q_n; t.flags = flags;
    *r = t;
}
// wave-uniform values the compiler cannot prove uniform: through readfirstlane they live in scalar registers, and the
// walker's bookkeeping and branches run on the scalar unit
__device__ __forceinline__ int kq_u32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t kq_u64(int64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(v & 0xFFFFFFFFll)), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int64_t kq_bcast64(int64_t v, int src) {
    const int lo = __shfl((int)(v & 0xFFFFFFFFll), src, 64), hi = __shfl((int)(v >> 32), src, 64);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}

__global__ __launch_bounds__(KQ_WALK_BLOCK) void k_kq_walk(const uint8_t *__restrict__ data, int64_t n, const uint4 *__restrict__ desc, int64_t L,
                                                  KqRec *__restrict__ recs, int64_t cap, int64_t *__restrict__ ldst,
                                                  uint32_t *__restrict__ lcon, const KqInit *__restrict__ init, KqOut *__restrict__ out) {
    __shared__ uint4 ring[KQ_RING];
    // flags between the waves: relaxed workgroup-scope atomics, which stay LDS instructions (volatile accesses become flat
    // ones, and those also wait for the wave's outstanding global stores)
    __shared__ uint32_t ready[KQ_SLOTS];                  // batch number + 1 that a slot holds
    __shared__ uint32_t cons_batch, stop;
    const int lane = lane_id(), w = threadIdx.x >> 6;
    if (threadIdx.x < KQ_SLOTS) ready[threadIdx.x] = 0;
    const KqInit in = *init;                               // where the parallel prefix passes left off (all zero: the start)
    const int64_t b0 = in.j0 / KQ_BATCH;
    if (threadIdx.x == 0) { cons_batch = (uint32_t)b0; stop = 0; }
    __syncthreads();
    const int64_t nb = (L + KQ_BATCH - 1) / KQ_BATCH;

    if (w > 0) {
        // ---------------- producers: batch b -> slot b % KQ_SLOTS, once the walker has left batch b - KQ_SLOTS
        for (int64_t b = b0 + w - 1; b < nb; b += KQ_PRODUCERS) {
            uint4 v[KQ_BATCH / 64];
#pragma unroll
            for (int r = 0; r < KQ_BATCH / 64; ++r) {
                const int64_t i = b * KQ_BATCH + r * 64 + lane;
                v[r] = i < L ? desc[i] : make_uint4(0, 0, 0, 0);
            }
            if (b >= b0 + KQ_SLOTS)
                while ((int64_t)KQ_LD(cons_batch) < b - (KQ_SLOTS - 1) && !KQ_LD(stop)) __builtin_amdgcn_s_sleep(2);
            if (KQ_LD(stop)) return;
            const int slot = (int)(b % KQ_SLOTS);
#pragma unroll
            for (int r = 0; r < KQ_BATCH / 64; ++r) ring[slot * KQ_BATCH + r * 64 + lane] = v[r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // the batch is in LDS before its flag
            if (lane == 0) KQ_ST(ready[slot], (uint32_t)(b + 1));
        }
        return;
    }

    // ---------------- the walker (wave 0).  Everything but the per-lane window values is wave-uniform.
    int st = (int)in.st, code = 0;
    int64_t j = in.j0, S = in.S, nrec = in.nrec;
    int64_t cur_off = in.cur_off, cur_line = in.cur_line, acc = in.acc, qacc = 0, tcr = 0, lastc = 0, qfirst = 0;
    uint32_t cur_len = in.cur_len, cur_flags = in.cur_flags, qn = 0, sn = 0;
    int64_t have = b0 - 1;                                  // batches [b0, have] are known to be in the ring
    uint32_t pub = (uint32_t)b0;
    const uint64_t lt = (1ull << lane) - 1ull;

    while (j < L && !code) {
        st = kq_u32(st); j = kq_u64(j); S = kq_u64(S); nrec = kq_u64(nrec); acc = kq_u64(acc); have = kq_u64(have); pub = (uint32_t)kq_u32((int)pub);
        cur_off = kq_u64(cur_off); cur_line = kq_u64(cur_line); cur_len = (uint32_t)kq_u32((int)cur_len); cur_flags = (uint32_t)kq_u32((int)cur_flags);
        // the window: lines j .. j + 63
        const int64_t lastline = j + 63 < L ? j + 63 : L - 1, bneed = lastline / KQ_BATCH;
        while (have < bneed) {
            const int64_t b = have + 1;
            while (KQ_LD(ready[b % KQ_SLOTS]) != (uint32_t)(b + 1)) __builtin_amdgcn_s_sleep(1);
            have = b;
        }
        // The ring is read behind the flag in program order and LDS serves a wave in order: only the compiler has to be kept
        // from moving the reads up.  (A workgroup fence here would also wait for this wave's global stores of the step
        // before -- once per 64 lines, it was most of the walk's time.)
        asm volatile("" ::: "memory");
        const uint32_t jb = (uint32_t)(j / KQ_BATCH);
        if (jb != pub) { pub = jb; if (lane == 0) KQ_ST(cons_batch, jb); }
        const bool ex = j + lane < L;
        const uint4 d = ex ? ring[(j + lane) & (KQ_RING - 1)] : make_uint4(0, 0, 0, 0);
        const int64_t start = (int64_t)(((uint64_t)d.y << 32) | d.x);
        const uint32_t len = d.z, first = d.w & 0xFF, last = (d.w >> 8) & 0xFF;
        const bool un = (d.w >> 16) & 1;
        const bool ishdr = ex && len >= 1 && (first == '>' || first == '@');
        const bool big = __ballot(ex && len >= KQ_BIG) != 0;

        if (st == KQ_SEEK && !big) {
            // ---- 16 four-line records at once (the lines of a well-formed FASTQ file)
            const int role = lane & 3;
            const uint32_t con = len - ((last == 13 && len > 1) ? 1u : 0u);
            const uint32_t con2 = (uint32_t)__shfl((int)con, (lane + 62) & 63, 64);      // of the line two above
            bool ok;
            if (role == 0) ok = ishdr;
            else if (role == 1) ok = ex && len >= 1 && first != '>' && first != '@' && first != '+';
            else if (role == 2) ok = ex && len >= 1 && first == '+' && !un;
            else ok = ex && con == con2;
            const uint64_t m = __ballot(ok);
            const uint64_t g = m & (m >> 1) & (m >> 2) & (m >> 3) & 0x1111111111111111ull;
            const uint64_t bad = ~g & 0x1111111111111111ull;
            const int R = kq_u32(bad ? (__builtin_ctzll(bad) >> 2) : 16);
            if (R > 0) {
                const bool mine = (lane >> 2) < R;
                const uint32_t pin = wave_incl_scan((role == 1 && mine) ? con : 0u);
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)pin, 63);
                const uint32_t seqlen = (uint32_t)__shfl((int)con, (lane + 1) & 63, 64);
                if (mine) {
                    if (role == 0) {
                        const int64_t ri = nrec + (lane >> 2);
                        if (ri < cap) kq_rec_store(&recs[ri], start + 1, j + lane, seqlen, S + pin, len - 1, 1, 1, KQ_F_FASTQ);
                    } else if (role != 2) {
                        ldst[j + lane] = (S + pin - con) | ((int64_t)(role == 1 ? KQ_C_SEQ : KQ_C_QUAL) << 62);
                        lcon[j + lane] = con;
                    }
                }
                nrec += R; S += total; j += 4 * R;
                continue;
            }
        }
        // lane 0 holds line j
        const int64_t s0 = kq_bcast64(start, 0);
        const uint32_t len0 = (uint32_t)__shfl((int)len, 0, 64), info0 = (uint32_t)__shfl((int)d.w, 0, 64);
        const uint32_t f0 = info0 & 0xFF, la0 = (info0 >> 8) & 0xFF;
        const bool un0 = (info0 >> 16) & 1;

        if (st == KQ_SEEK) {
            if (len0 >= 1 && (f0 == '>' || f0 == '@')) { st = KQ_HDR; continue; }
            // kseq.c:142-146: the next '>' or '@' wherever it stands
            int64_t p = -1;
            for (int64_t q = 0; q < (int64_t)len0 && p < 0; q += 64) {
                const bool hit = q + lane < (int64_t)len0 && (data[s0 + q + lane] == '>' || data[s0 + q + lane] == '@');
                const uint64_t hm = __ballot(hit);
                if (hm) p = s0 + q + __builtin_ctzll(hm);
            }
            if (p >= 0) {
                if (p + 1 >= n) { code = -1; break; }
                cur_off = p + 1; cur_len = (uint32_t)(s0 + len0 - (p + 1)); cur_line = j; cur_flags = un0 ? KQ_F_HDR_UNTERM : 0;
                st = KQ_SEQ; acc = 0;
            }
            ++j;
            continue;
        }
        if ((st == KQ_HDR || st == KQ_SEQ) && !big) {
            // ---- a run of header / sequence lines at once, up to the first line that needs a closer look
            const bool stopper = !ex || (len >= 1 && first == '+') || (len == 1 && first == 13) || un;
            const uint64_t sm = __ballot(stopper);
            const int m = kq_u32(sm ? __builtin_ctzll(sm) : 64);
            if (m > 0) {
                const bool inr = lane < m;
                const bool H = inr && ishdr;
                const uint32_t con = (inr && !H && len > 0) ? len - (last == 13 ? 1u : 0u) : 0u;
                const uint32_t pin = wave_incl_scan(con), pex = pin - con;
                const uint64_t hm = __ballot(H);
                const bool carried = st == KQ_SEQ;
                const int64_t carry = carried ? acc : 0;
                if (inr && !H && len > 0) {
                    ldst[j + lane] = (S + carry + pex) | ((int64_t)KQ_C_SEQ << 62);
                    lcon[j + lane] = con;
                }
                const uint32_t pin_end = (uint32_t)__shfl((int)pin, m - 1, 64);
                if (hm) {
                    const int h0 = __builtin_ctzll(hm), hl = 63 - __builtin_clzll(hm), nh = __popcll(hm);
                    const uint32_t pex_h0 = (uint32_t)__shfl((int)pex, h0, 64);
                    if (carried) {
                        if (lane == 0 && nrec < cap)
                            kq_rec_store(&recs[nrec], cur_off, cur_line, acc + pex_h0, S, cur_len, (uint32_t)(j + h0 - cur_line - 1), 0, cur_flags);
                        ++nrec;
                    }
                    const uint64_t above = lane < 63 ? hm >> (lane + 1) : 0ull;
                    const int h2 = above ? lane + 1 + __builtin_ctzll(above) : lane;
                    const uint32_t pex_h2 = (uint32_t)__shfl((int)pex, h2, 64);
                    if (H && above) {
                        const int64_t ri = nrec + __popcll(hm & lt);
                        if (ri < cap) kq_rec_store(&recs[ri], start + 1, j + lane, pex_h2 - pin, S + carry + pin, len - 1, (uint32_t)(h2 - lane - 1), 0, 0);
                    }
                    nrec += nh - 1;
                    const uint32_t pin_hl = (uint32_t)__shfl((int)pin, hl, 64);
                    cur_off = kq_bcast64(start, hl) + 1; cur_len = (uint32_t)__shfl((int)len, hl, 64) - 1; cur_line = j + hl; cur_flags = 0;
                    S += carry + pin_hl;
                    acc = pin_end - pin_hl;
                } else {
                    acc += pin_end;
                }
                st = KQ_SEQ;
                j += m;
                continue;
            }
        }
        // ---- one line
        if (st == KQ_HDR) {
            if (un0 && len0 == 1) { code = -1; break; }
            cur_off = s0 + 1; cur_len = len0 - 1; cur_line = j; cur_flags = un0 ? KQ_F_HDR_UNTERM : 0;
            st = KQ_SEQ; acc = 0;
            ++j;
        } else if (st == KQ_SEQ) {
            if (len0 == 0) {
                ++j;
            } else if (f0 == '>' || f0 == '@') {
                if (lane == 0 && nrec < cap) kq_rec_store(&recs[nrec], cur_off, cur_line, acc, S, cur_len, (uint32_t)(j - cur_line - 1), 0, cur_flags);
                ++nrec; S += acc;
                st = KQ_HDR;
            } else if (f0 == '+') {
                if (un0) { code = -2; break; }
                sn = (uint32_t)(j - cur_line - 1);
                st = KQ_QUAL; qacc = 0; qn = 0; tcr = 0; lastc = j; qfirst = j;
                ++j;
            } else {
                // the first byte goes in by itself (kseq.c:156); the strip belongs to the call for the rest of the line,
                // which returns early when nothing at all is left: a lone CR as the last byte of the stream stays
                const uint32_t con = len0 - ((la0 == 13 && acc + len0 > 1 && !(un0 && len0 == 1)) ? 1u : 0u);
                if (lane == 0) { ldst[j] = (S + acc) | ((int64_t)KQ_C_SEQ << 62); lcon[j] = con; }
                acc += con;
                ++j;
            }
        } else {
            // KQ_QUAL.  ks_getuntil2 strips ONE trailing CR per call from a string longer than one byte (kseq.c:106), also
            // in a call that appends nothing: tcr = the run of CRs the quality string ends with, lastc = the last line that
            // holds bytes of it.
            int64_t tr = 0;
            if (la0 == 13) while (tr < (int64_t)len0 && data[s0 + len0 - 1 - tr] == 13) ++tr;
            uint32_t con = len0;
            qacc += len0;
            tcr = tr == (int64_t)len0 ? tcr + len0 : tr;
            if (tcr >= 1 && qacc > 1) {
                --qacc; --tcr;
                if (len0) --con;
                else if (lane == 0) {                     // the CR belongs to an earlier line: the last one that still holds bytes
                    while (lastc > qfirst && lcon[lastc] == 0) --lastc;       // (there is one: the lines' counts add up to qacc)
                    if (lcon[lastc]) lcon[lastc] -= 1;
                }
            }
            lastc = kq_u64(lastc);                         // lane 0's
            if (lane == 0) { ldst[j] = (S + qacc - con) | ((int64_t)KQ_C_QUAL << 62); lcon[j] = con; }
            if (con) lastc = j;
            ++qn; ++j;
            if (qacc >= acc) {
                if (qacc != acc) { code = -2; break; }
                if (lane == 0 && nrec < cap) kq_rec_store(&recs[nrec], cur_off, cur_line, acc, S, cur_len, sn, qn, cur_flags | KQ_F_FASTQ);
                ++nrec; S += acc;
                st = KQ_SEEK;
            }
        }
    }
    if (!code) {
        if (st == KQ_SEQ) {
            if (lane == 0 && nrec < cap) kq_rec_store(&recs[nrec], cur_off, cur_line, acc, S, cur_len, (uint32_t)(L - cur_line - 1), 0, cur_flags);
            ++nrec; S += acc;
            code = -1;
        } else if (st == KQ_QUAL) {
            if (qn == 0 && acc == 0) {
                if (lane == 0 && nrec < cap) kq_rec_store(&recs[nrec], cur_off, cur_line, 0, S, cur_len, sn, 0, cur_flags | KQ_F_FASTQ | KQ_F_UNTOUCHED);
                ++nrec;
                code = -1;
            } else code = -2;
        } else code = -1;
    }
    if (lane == 0) {
        KQ_ST(stop, 1u);
        out->n_rec = nrec; out->code = code; out->seq_bytes = S; out->pad = nrec > cap ? 1 : 0;
    }
}

// ------------------------------------------------------------------ gather
// Lines [l0, l1] of the table -> seq_dst / qual_dst at (position - cum0): one 16-lane group per line, 16 bytes per lane
// and step.  Lines longer than KQ_LONG go on a list (k_kq_gather_long: the whole grid on each of them).
struct KqLong { int64_t src, dst; uint32_t len, cls; };
__device__ __forceinline__ void kq_copy16(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, int64_t o, int64_t len, bool upper) {
    if (o + 16 <= len) {
        uint4 v = *reinterpret_cast<const uint4_u *>(src + o);
        if (upper) { v.x = upper4(v.x); v.y = upper4(v.y); v.z = upper4(v.z); v.w = upper4(v.w); }
        *reinterpret_cast<uint4_u *>(dst + o) = v;
    } else {
        for (int64_t k = o; k < len; ++k) {
            uint8_t c = src[k];
            if (upper && c >= 'a' && c <= 'z') c -= 32;
            dst[k] = c;
        }
    }
}
__global__ __launch_bounds__(BLOCK) void k_kq_gather(const uint8_t *__restrict__ data, const int64_t *__restrict__ nl, const int64_t *__restrict__ ldst,
                                                    const uint32_t *__restrict__ lcon, int64_t l0, int64_t l1, int64_t cum0, uint8_t *__restrict__ seq_dst,
                                                    uint8_t *__restrict__ qual_dst, int upper, KqLong *__restrict__ longs, uint32_t *__restrict__ n_long,
                                                    uint32_t long_cap) {
    const int gl = threadIdx.x & 15;
    const int64_t grp = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 4, ngrp = ((int64_t)gridDim.x * BLOCK) >> 4;
    for (int64_t i = l0 + grp; i <= l1; i += ngrp) {
        const int64_t e = ldst[i];
        const int cls = (int)((uint64_t)e >> 62);
        if (!cls) continue;
        uint8_t *base = cls == KQ_C_SEQ ? seq_dst : qual_dst;
        const uint32_t con = lcon[i];
        if (!base || !con) continue;
        const int64_t src = i ? nl[i - 1] + 1 : 0, dpos = (e & KQ_POS) - cum0;
        if (con > (uint32_t)KQ_LONG) {
            if (gl == 0) {
                const uint32_t k = atomicAdd(n_long, 1u);
                if (k < long_cap) longs[k] = KqLong{src, dpos, con, (uint32_t)cls};
            }
            continue;
        }
        const bool up = upper && cls == KQ_C_SEQ;
        for (int64_t o = gl * 16; o < (int64_t)con; o += 256) kq_copy16(base + dpos, data + src, o, con, up);
    }
}
__global__ __launch_bounds__(BLOCK) void k_kq_gather_long(const uint8_t *__restrict__ data, const KqLong *__restrict__ longs, uint32_t n_long,
                                                         uint8_t *__restrict__ seq_dst, uint8_t *__restrict__ qual_dst, int upper) {
    for (uint32_t e = 0; e < n_long; ++e) {
        const KqLong q = longs[e];
        uint8_t *base = q.cls == KQ_C_SEQ ? seq_dst : qual_dst;
        const bool up = upper && q.cls == KQ_C_SEQ;
        for (int64_t o = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * 16; o < (int64_t)q.len; o += (int64_t)gridDim.x * BLOCK * 16)
            kq_copy16(base + q.dst, data + q.src, o, q.len, up);
    }
}

}  // namespace fx
