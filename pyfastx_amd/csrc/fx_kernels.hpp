// fx_kernels.hpp -- hand-written HIP kernels for gfx950 (MI355X, wave64).
//
// The reference (lmdu/pyfastx) walks the file line by line on one CPU thread
// (kseq.c:59-109 feeding index.c:230-339 / fastq.c:89-149).  None of that
// structure survives here: the stream is resident in HBM and is processed as
//
//   FASTA index   fx_spanscan.hpp: one read of the stream, per-4-KiB summaries, no line table
//   FASTQ index   fx_fastq.hpp: count pass + emit pass (one lane per newline), no line table
//   fetch         k_fetch / k_fastq_fetch: gather, despace, upper, revcomp, phred  (index.c:683-707, util.c:157-269, read.c)
//   composition   k_fasta_comp (fx_comp.hpp), k_fastq_comp (fx_fastq.hpp)
//
// Integer/byte work only: no MFMA anywhere; the roofline is HBM bandwidth.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fx {

constexpr int BLOCK = 256;                      // 4 waves of 64
constexpr int UNROLL = 8;                       // 16-byte loads in flight per lane
constexpr int CHUNK = 16;                       // bytes per lane per load (global_load_dwordx4)
constexpr int TILE = BLOCK * CHUNK * UNROLL;    // 32 KiB of file per workgroup

// ---------------------------------------------------------------- SWAR bytes
// 0x80 in every byte of x that is zero (exact, no borrow false positives).
__device__ __forceinline__ uint32_t zero_bytes(uint32_t x) {
    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}
// gather the four 0x80 flags of a word into bits 0..3: v_dot4_u32_u8 with byte weights 1,2,4,8 gives 128 * mask
__device__ __forceinline__ uint32_t flags4(uint32_t t) { return __builtin_amdgcn_udot4(t, 0x08040201u, 0u, false) >> 7; }
// 16-bit mask of the bytes of v equal to the byte replicated in pat (bit k <-> byte k): two dot4 accumulations
// per 8 bytes (weights 1..8 and 16..128), 128 * mask8 each
__device__ __forceinline__ uint32_t eq_mask16(const uint4 &v, uint32_t pat) {
    uint32_t lo = __builtin_amdgcn_udot4(zero_bytes(v.x ^ pat), 0x08040201u, 0u, false);
    lo = __builtin_amdgcn_udot4(zero_bytes(v.y ^ pat), 0x80402010u, lo, false);
    uint32_t hi = __builtin_amdgcn_udot4(zero_bytes(v.z ^ pat), 0x08040201u, 0u, false);
    hi = __builtin_amdgcn_udot4(zero_bytes(v.w ^ pat), 0x80402010u, hi, false);
    return (lo >> 7) | (hi << 1);
}

// 16 bytes at data[p..p+16); bytes at or beyond n read as 0 (never '\n' or '>').
__device__ __forceinline__ uint4 load16(const uint8_t *__restrict__ data, int64_t p, int64_t n) {
    if (p + CHUNK <= n) return *reinterpret_cast<const uint4 *>(data + p);
    uint32_t w[4] = {0, 0, 0, 0};
    for (int k = 0; k < CHUNK; ++k)
        if (p + k < n) w[k >> 2] |= (uint32_t)data[p + k] << ((k & 3) * 8);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// ------------------------------------------------------------- wave / block
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// inclusive prefix sum over the 64 lanes with DPP row shifts / broadcasts: six v_add_u32_dpp instead of six
// ds_bpermute round trips (row_shr:1,2,4,8 inside each row of 16, row_bcast:15 into rows 1 and 3, row_bcast:31
// into rows 2 and 3 -- the GFX9 scan sequence)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ int64_t wave_sum64(int64_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t *lds4) {
    const int w = threadIdx.x >> 6, l = lane_id();
    v = wave_sum(v);
    __syncthreads();
    if (l == 0) lds4[w] = v;
    __syncthreads();
    return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

// first index i in [0,n) with a[i] > key  (n if none)
__device__ __forceinline__ int64_t upper_bound(const int64_t *__restrict__ a, int64_t n, int64_t key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (a[mid] <= key) lo = mid + 1; else hi = mid; }
    return lo;
}

// FASTA record table columns (SoA, one entry per header line).  Column semantics follow
// index.c:234-339 exactly, including the quirks (elen taken from the header line only,
// index.c:266-269; blen/boff in "position" units that over-count by one when the stream lacks a
// final '\n', index.c:231 -- the virtual end-of-stream newline reproduces that).  The kernels
// that fill it live in fx_spanscan.hpp.
struct FastaCols {
    int64_t *hoff, *boff, *blen, *slen, *llen, *hdr_line;
    int32_t *elen, *dlen, *name_len, *norm;
    uint32_t *bad;
    int32_t *reg;                    // line-regular (line_regular below): may slices use the line arithmetic?
};

// `norm` (index.c:342) says "at most one line of another length": true of a record whose LAST line is the short one,
// and of a record with one odd line anywhere else -- where the line arithmetic of sequence.c:498-510 reads the wrong
// bytes (the reference returns them from a cold cache and the true slice from a warm one).  A record is LINE-REGULAR
// when every sequence line but the last holds exactly llen - elen bases and the last one 1 .. llen - elen: then base i
// sits at boff + i + elen * (i / bpl) for every i.  Given norm = 1 that is decided by the columns and ONE byte of the
// stream: the record has ceil(slen / bpl) lines, and the byte in front of the last line -- x = slen - (lines - 1) * bpl
// bases + elen terminator bytes before the record's end -- is a newline.  (Were the odd line somewhere else, that
// position would lie inside a full last line.  More than one line after it would make two short lines: norm = 0.)
// Slices of records that fail the test are cut from the despaced record (sequence.c:100-110) by every fetch path.
// -> 1 / 0, -1 when the deciding byte is not among the n bytes held here (a record that crosses a shard cut).
__device__ __forceinline__ int line_regular(const uint8_t *__restrict__ data, int64_t n, int64_t gbase, int64_t boff, int64_t blen,
                                            int64_t slen, int64_t llen, int64_t elen, int norm) {
    const int64_t bpl = llen - elen;
    if (!norm || bpl <= 0 || elen <= 0) return 0;
    if (slen <= bpl) return 1;                               // at most one line of bases
    const int64_t lines = (slen + bpl - 1) / bpl;
    if (blen != slen + lines * elen) return 0;               // not ceil(slen / bpl) sequence lines
    const int64_t x = slen - (lines - 1) * bpl;              // bases of the last line if the record is regular
    const int64_t p = boff + blen - (x + elen) - 1 - gbase;
    if (p < 0 || p >= n) return -1;
    return data[p] == '\n' ? 1 : 0;
}

// ======================================================================= K7
// Batched fetch.  One wave per query: lanes read consecutive bytes of
// [off, off+blen), keep-mask = byte not in {10,13,32} (jump_table, util.c:157-164),
// __ballot + popcount of lower lanes gives each kept byte its rank, ranks in
// [skip, skip+take) are emitted at dst[rank-skip] (mirrored for FX_REVERSE),
// optionally upper-cased (Py_TOUPPER, util.c:181-194) and complemented through
// the IUPAC LUT in LDS (comp_map, util.c:228-237).
// Replaces per query: fseeko+fread (index.c:688-689), remove_space*
// (util.c:166-194), reverse/complement (util.c:239-269), memcpy
// (sequence.c:346-347).
struct FetchQ {
    const int64_t *off, *blen, *skip, *take;     // skip may be null (0)
    const int64_t *seq_id, *start, *stop;        // alternative: resolve against the FASTA table
    const uint8_t *qflags;                       // per-query flags or null
    const int64_t *dst_off;
    int64_t *out_len;
};
struct FastaTab {
    const int64_t *boff, *blen, *slen, *llen;
    const int32_t *elen, *norm;      // norm here = the line-regular column (FastaCols::reg), not index.c's norm
    int64_t n_seq;                   // records in the table -- or its capacity while a build is still in flight ...
    const long long *n_seq_dev;      // ... in which case the count is read here (device memory), null otherwise
};

__device__ __forceinline__ void build_comp_lut(uint8_t *lut) {
    // IUPAC complement (util.c:204-237): A<->T C<->G M<->K R<->Y V<->B H<->D, U->A, case kept,
    // W S N and everything else map to themselves.  Bytes >= 128: identity.
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        uint8_t c = (uint8_t)i, u = c & 0xDF, r = c;
        const bool letter = (u >= 'A' && u <= 'Z') && (c < 128);
        if (letter) {
            uint8_t m = u;
            switch (u) {
            case 'A': m = 'T'; break; case 'T': m = 'A'; break; case 'U': m = 'A'; break;
            case 'C': m = 'G'; break; case 'G': m = 'C'; break;
            case 'M': m = 'K'; break; case 'K': m = 'M'; break;
            case 'R': m = 'Y'; break; case 'Y': m = 'R'; break;
            case 'V': m = 'B'; break; case 'B': m = 'V'; break;
            case 'H': m = 'D'; break; case 'D': m = 'H'; break;
            default: break;
            }
            r = m | (c & 0x20);
        }
        lut[i] = r;
    }
}

// ---- helpers of the line-arithmetic fetch path (k_fetch, BY_ID, norm = 1 records) ------------------
typedef uint4 __attribute__((aligned(1))) uint4_u;       // 16 bytes at any byte address (gfx9 unaligned access mode)
typedef uint2 __attribute__((aligned(1))) uint2_u;
typedef uint32_t __attribute__((aligned(1))) uint32_u;
typedef uint16_t __attribute__((aligned(1))) uint16_u;

// 32-bit word i of the 128-bit mask whose low T bytes (0 <= T <= 16) are 0xFF
__device__ __forceinline__ uint32_t lowbytes_word(int T, int i) {
    const int tt = T - 4 * i;
    return tt >= 4 ? 0xFFFFFFFFu : (tt <= 0 ? 0u : ((1u << (8 * tt)) - 1u));
}
// 0x80 in every byte of w that is < 0x80 and <= 0x20 (white space and control bytes)
__device__ __forceinline__ uint32_t le20_bytes(uint32_t w) { return ~(((w & 0x7F7F7F7Fu) + 0x5F5F5F5Fu) | w) & 0x80808080u; }
// Py_TOUPPER on four bytes (util.c:181-194): 'a'..'z' -> 'A'..'Z', everything else unchanged
__device__ __forceinline__ uint32_t upper4(uint32_t w) {
    const uint32_t x = w & 0x7F7F7F7Fu;
    const uint32_t lower = (x + 0x1F1F1F1Fu) & ~(x + 0x05050505u) & ~w & 0x80808080u;     // 0x61 <= c <= 0x7A
    return w - (lower >> 2);
}
__device__ __forceinline__ uint32_t lut4(const uint8_t *lut, uint32_t w) {
    return (uint32_t)lut[w & 0xFF] | ((uint32_t)lut[(w >> 8) & 0xFF] << 8) | ((uint32_t)lut[(w >> 16) & 0xFF] << 16) |
           ((uint32_t)lut[w >> 24] << 24);
}
__device__ __forceinline__ bool is_space3(uint32_t c) { return c == 10u || c == 13u || c == 32u; }   // jump_table, util.c:157-164
// the low `len` (1..16) bytes of v to p, any alignment
__device__ __forceinline__ void store_low_bytes(uint8_t *p, uint4 v, int len) {
    if (len >= 16) { *reinterpret_cast<uint4_u *>(p) = v; return; }
    if (len & 8) { *reinterpret_cast<uint2_u *>(p) = make_uint2(v.x, v.y); p += 8; v.x = v.z; v.y = v.w; }
    if (len & 4) { *reinterpret_cast<uint32_u *>(p) = v.x; p += 4; v.x = v.y; }
    if (len & 2) { *reinterpret_cast<uint16_u *>(p) = (uint16_t)v.x; p += 2; v.x >>= 16; }
    if (len & 1) *p = (uint8_t)v.x;
}

// G lanes cooperate on one query (64/G queries in flight per wave); each lane
// loads V aligned bytes per step, so a step covers a G*V-byte window:
//   < 8,16>  128-byte window, 8 queries per wave  -- ~100-bp random access
//   <64,16>  1 KiB window, 1 query per wave       -- long ranges (whole records)
// General path: the keep mask of a lane's V bytes is SWAR, the rank of its first kept byte is
// an exclusive prefix over the G lanes (__shfl_up, width G), kept bytes whose
// rank falls in [skip, skip+take) are stored (mirrored for FX_REVERSE).
// Line-arithmetic path (BY_ID, norm = 1, >= 16 bases per line): base i of a record sits at byte
// boff + i + elen * (i / bases_per_line) (sequence.c:498-510), so a lane produces 16 OUTPUT bytes
// from one unaligned 16-byte load -- two when a line end falls inside, merged with a byte mask --
// and writes them with one unaligned 16-byte store: no keep mask, no ranks, no per-byte loop.  It
// checks that the skipped bytes are line terminators and that no white space hides inside the
// lines; if either fails the query is redone by the general path, so results always equal
// "read the byte range, drop 10/13/32" (index.c:694-707, util.c:157-194).
template <bool BY_ID, int G, int V>
__global__ __launch_bounds__(BLOCK) void k_fetch(const uint8_t *__restrict__ data, int64_t gbase, int64_t n_bytes,
                                                FetchQ q, FastaTab tab, int64_t nq, int flags_all,
                                                uint8_t *__restrict__ dst, const int32_t *__restrict__ list = nullptr,
                                                const int *__restrict__ list_n = nullptr) {
    __shared__ uint8_t lut[256];
    if (list) {                                                // list mode: only the queries k_fetch_lines left over
        nq = *list_n;
        if (nq == 0) return;                                   // (for a genome: none -- leave before the tables are set up, the launch is all this costs)
    }
    // BY_ID: the record table of a genome (a few hundred rows) is copied to LDS once per workgroup, so
    // resolving a query costs one LDS read instead of a second dependent trip to memory
    constexpr int TABCAP = BY_ID ? 512 : 1;
    __shared__ int64_t s_boff[TABCAP], s_slen[TABCAP], s_blen[TABCAP];
    __shared__ int32_t s_llen[TABCAP], s_en[TABCAP];           // s_en = elen | norm << 8
    const int64_t n_seq = (BY_ID && tab.n_seq_dev) ? (*tab.n_seq_dev < tab.n_seq ? (int64_t)*tab.n_seq_dev : tab.n_seq) : tab.n_seq;
    const bool tab_lds = BY_ID && n_seq <= TABCAP && n_seq > 0;
    build_comp_lut(lut);
    if (tab_lds)
        for (int r = threadIdx.x; r < (int)n_seq; r += BLOCK) {
            s_boff[r] = tab.boff[r]; s_slen[r] = tab.slen[r]; s_blen[r] = tab.blen[r];
            const int64_t ll = tab.llen[r];
            s_llen[r] = ll > 0x7FFFFFFFll ? 0x7FFFFFFF : (int32_t)ll;
            s_en[r] = tab.elen[r] | (tab.norm[r] << 8);
        }
    __syncthreads();
    constexpr int QPW = 64 / G, NW = V / 4;
    const int lane = lane_id(), sub = lane & (G - 1), grp = lane / G;
    const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
    // query descriptors are fetched one iteration ahead (the loop is a chain of dependent memory round
    // trips: descriptor -> table -> sequence bytes; this takes the first one off the critical path)
    int64_t d_a0 = 0, d_a1 = 0, d_a2 = 0, d_a3 = 0, d_off = 0;
    int d_fl = flags_all;
    int64_t d_i = 0;                                           // the query the descriptor belongs to
    auto fetch_desc = [&](int64_t k) {
        if (k >= nq) return;
        const int64_t i = list ? (int64_t)list[k] : k;
        d_i = i;
        if (BY_ID) { d_a0 = q.seq_id[i]; d_a1 = q.start[i]; d_a2 = q.stop[i]; }
        else       { d_a0 = q.off[i]; d_a1 = q.blen[i]; d_a2 = q.take[i]; d_a3 = q.skip ? q.skip[i] : 0; }
        d_off = q.dst_off[i];
        d_fl = q.qflags ? q.qflags[i] : flags_all;
    };
    fetch_desc(wave * QPW + grp);
    for (int64_t i0 = wave * QPW; i0 < nq; i0 += nwaves * QPW) {
        const int64_t kq = i0 + grp;
        bool ok = kq < nq;
        const int64_t i = d_i;
        const int64_t c_a0 = d_a0, c_a1 = d_a1, c_a2 = d_a2, c_a3 = d_a3, c_off = d_off;
        const int fl = d_fl;
        fetch_desc(kq + nwaves * QPW);
        int64_t off = 0, blen = 0, skip = 0, take = 0;
        int64_t r_boff = 0, r_el = 0, r_bpl = 0;
        bool r_norm = false;
        if (ok) {
            if (BY_ID) {
                const int64_t id = c_a0, a = c_a1, b = c_a2;
                int64_t r_slen = 0, r_blen = 0;
                if (id >= 0 && id < n_seq) {
                    if (tab_lds) { r_boff = s_boff[id]; r_slen = s_slen[id]; r_blen = s_blen[id]; r_el = s_en[id] & 0xFF; r_norm = (s_en[id] >> 8) != 0;
                                   r_bpl = (int64_t)s_llen[id] - r_el; if (s_llen[id] == 0x7FFFFFFF) r_bpl = tab.llen[id] - r_el; }
                    else         { r_boff = tab.boff[id]; r_slen = tab.slen[id]; r_blen = tab.blen[id]; r_el = tab.elen[id]; r_norm = tab.norm[id] != 0;
                                   r_bpl = tab.llen[id] - r_el; }
                }
                if (id < 0 || id >= n_seq || a < 0 || b < a || b > r_slen) {   // caller validates; stay safe
                    if (sub == 0 && q.out_len) q.out_len[i] = -1;
                    ok = false;
                } else {
                    take = b - a;
                    if (r_norm && r_bpl > 0) {                 // sequence.c:498-510
                        int64_t bs, be;
                        if (((uint64_t)b | (uint64_t)r_bpl) >> 32) { bs = a / r_bpl; be = b / r_bpl; }
                        else { bs = (uint32_t)a / (uint32_t)r_bpl; be = (uint32_t)b / (uint32_t)r_bpl; }
                        off = r_boff + a + r_el * bs;
                        blen = take + (be - bs) * r_el;
                    } else {                                   // sequence.c:100-110: despace whole record, then slice
                        off = r_boff; blen = r_blen; skip = a;
                    }
                }
            } else {
                off = c_a0; blen = c_a1; take = c_a2; skip = c_a3;
            }
        }
        uint8_t *out = dst + (ok ? c_off : 0);
        if (BY_ID) {
            // ---- line-arithmetic path
            bool fast = false;
            int64_t a = 0, el = 0, bpl = 0, in_a = 0;
            if (ok && take > 0 && !(fl & 8)) {
                a = c_a1; el = r_el; bpl = r_bpl;
                in_a = off - gbase;                               // local offset of base `a` (off was computed above)
                fast = r_norm && bpl >= 16 && bpl < (1ll << 31) && a + take < (1ll << 31) &&
                       in_a >= 16 && in_a + blen + 32 <= n_bytes;
            }
            bool redo = false;
            if (fast) {
                const uint32_t bpl32 = (uint32_t)bpl;
                const uint32_t r0 = (uint32_t)a % bpl32;          // column of base `a` in its line
                const bool rev = (fl & 2) != 0;
                bool irregular = false;
                // Two pieces per lane and step (a 100-base query is ONE round of loads for its four lanes, not two): the
                // addresses of both first, then their loads -- the 16 bytes at the piece and, where a line ends inside or right
                // behind it, the 16 bytes `el` further -- and only then the byte work and the stores.  The terminator bytes a
                // piece skips are looked up in the registers it has loaded (byte loads of their own were two more gathers).
                for (int64_t s0 = 0; s0 < take; s0 += 2 * G * 16) {
                    int64_t oc[2], f0[2];
                    int len[2], lead[2];
                    uint32_t tt[2];
                    const uint8_t *p1[2];
                    bool need_w[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        oc[u] = s0 + u * (G * 16) + 16 * sub;         // this lane writes out[oc, oc + len)
                        int l = (int)(take - oc[u] < 16 ? take - oc[u] : 16);
                        len[u] = l < 0 ? 0 : l;
                        // forward indices [f0, f0 + len) of the query; `lead` unused bytes in front of them in the
                        // lane's 16-byte window (reverse strand: the partial chunk is the head of the query)
                        f0[u] = len[u] ? (rev ? take - oc[u] - len[u] : oc[u]) : 0;
                        lead[u] = len[u] && rev ? 16 - len[u] : 0;
                        const uint32_t xx = r0 + (uint32_t)f0[u];
                        const uint32_t k = xx / bpl32;
                        tt[u] = bpl32 - (xx - k * bpl32);               // bases to the end of f0's line
                        p1[u] = data + in_a + f0[u] + el * (int64_t)k - lead[u];
                        need_w[u] = len[u] > 0 && tt[u] <= (uint32_t)len[u];
                    }
                    uint4 v[2], w[2];
                    v[0] = *reinterpret_cast<const uint4_u *>(p1[0]);
                    v[1] = *reinterpret_cast<const uint4_u *>(p1[1]);
                    w[0] = w[1] = make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au);
                    if (need_w[0]) w[0] = *reinterpret_cast<const uint4_u *>(p1[0] + el);
                    if (need_w[1]) w[1] = *reinterpret_cast<const uint4_u *>(p1[1] + el);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (len[u] <= 0) continue;
                        uint4 x = v[u];
                        const uint4 y = w[u];
                        const int ln = len[u], ld = lead[u];
                        const uint32_t t = tt[u];
                        const int T = ld + (int)t;                      // index in the window of the first byte behind the line's bases
                        if (t < (uint32_t)ln || (t == (uint32_t)ln && f0[u] + ln < take)) {   // the terminator after base f0 + t - 1 is skipped
                            // window byte i is byte i of x (i < 16) or byte i - el of y: T <= 16, T + 1 <= 17
                            const int i0 = T < 16 ? T : T - (int)el, i1 = T + 1 < 16 ? T + 1 : T + 1 - (int)el;
                            const uint4 &s0v = T < 16 ? x : y, &s1v = T + 1 < 16 ? x : y;
                            const uint32_t a01 = (i0 & 4) ? s0v.y : s0v.x, a23 = (i0 & 4) ? s0v.w : s0v.z;
                            const uint32_t g0 = (((i0 & 8) ? a23 : a01) >> ((i0 & 3) * 8)) & 0xFFu;
                            const uint32_t b01 = (i1 & 4) ? s1v.y : s1v.x, b23 = (i1 & 4) ? s1v.w : s1v.z;
                            const uint32_t g1 = (((i1 & 8) ? b23 : b01) >> ((i1 & 3) * 8)) & 0xFFu;
                            irregular |= !is_space3(g0) || (el == 2 && !is_space3(g1));
                        }
                        if (t < (uint32_t)ln) {                         // a line ends inside: bytes from index T on come from el further
                            const uint32_t m0 = lowbytes_word(T, 0), m1 = lowbytes_word(T, 1), m2 = lowbytes_word(T, 2), m3 = lowbytes_word(T, 3);
                            x.x = (x.x & m0) | (y.x & ~m0); x.y = (x.y & m1) | (y.y & ~m1);
                            x.z = (x.z & m2) | (y.z & ~m2); x.w = (x.w & m3) | (y.w & ~m3);
                        }
                        {   // no white space among the bases taken
                            const int lo_b = ld, hi_b = ld + ln;
                            uint32_t f = le20_bytes(x.x) & lowbytes_word(hi_b, 0) & ~lowbytes_word(lo_b, 0);
                            f |= le20_bytes(x.y) & lowbytes_word(hi_b, 1) & ~lowbytes_word(lo_b, 1);
                            f |= le20_bytes(x.z) & lowbytes_word(hi_b, 2) & ~lowbytes_word(lo_b, 2);
                            f |= le20_bytes(x.w) & lowbytes_word(hi_b, 3) & ~lowbytes_word(lo_b, 3);
                            irregular |= f != 0;
                        }
                        if (fl & 1) { x.x = upper4(x.x); x.y = upper4(x.y); x.z = upper4(x.z); x.w = upper4(x.w); }
                        if (fl & 4) { x.x = lut4(lut, x.x); x.y = lut4(lut, x.y); x.z = lut4(lut, x.z); x.w = lut4(lut, x.w); }
                        if (rev) x = make_uint4(__builtin_bswap32(x.w), __builtin_bswap32(x.z), __builtin_bswap32(x.y), __builtin_bswap32(x.x));
                        store_low_bytes(out + oc[u], x, ln);
                    }
                }
                const unsigned long long ib = __ballot(irregular);
                constexpr unsigned long long gmask = G >= 64 ? ~0ull : ((1ull << (G & 63)) - 1ull);
                redo = ((ib >> (grp * G)) & gmask) != 0;
                if (!redo && sub == 0 && q.out_len) q.out_len[i] = take;
            }
            if (fast && !redo) ok = false;                         // done: nothing left for the general path
            if (!ok) { take = 0; blen = 0; }
        }
        // ---- general path
        // clamp to the bytes we hold (fread past EOF returns short, index.c:689)
        int64_t lo = off - gbase, hi = lo + blen;
        if (lo < 0) lo = 0;
        if (hi > n_bytes) hi = n_bytes;
        if (!ok) { lo = 0; hi = 0; }
        const int64_t end = skip + take;
        int64_t rank = 0;                                      // kept bytes before this window
        for (int64_t p = lo & ~(int64_t)(V - 1); p < hi && rank < end; p += G * V) {
            const int64_t pp = p + (int64_t)sub * V;
            uint32_t w[NW];
#pragma unroll
            for (int k = 0; k < NW; ++k) w[k] = 0;
            if (pp < hi) {
                if (V == 16) { const uint4 t = *reinterpret_cast<const uint4 *>(data + pp); w[0] = t.x; w[1] = t.y; w[NW - 2] = t.z; w[NW - 1] = t.w; }
                else         { const uint2 t = *reinterpret_cast<const uint2 *>(data + pp); w[0] = t.x; w[NW - 1] = t.y; }
            }
            // bits of the lane's V bytes that are inside [lo,hi) and not white space
            int64_t a0 = lo - pp, a1 = hi - pp;
            a0 = a0 < 0 ? 0 : (a0 > V ? V : a0);
            a1 = a1 < 0 ? 0 : (a1 > V ? V : a1);
            uint32_t km = (a1 > a0) ? (((1u << a1) - 1u) & ~((1u << a0) - 1u)) & ((V == 16) ? 0xFFFFu : 0xFFu) : 0u;
            if (!(fl & 8)) {                                   // jump_table (util.c:157-164): drop 10, 13, 32
                uint32_t sp = 0;
#pragma unroll
                for (int k = 0; k < NW; ++k)
                    sp |= flags4(zero_bytes(w[k] ^ 0x0A0A0A0Au) | zero_bytes(w[k] ^ 0x0D0D0D0Du) |
                                 zero_bytes(w[k] ^ 0x20202020u)) << (4 * k);
                km &= ~sp;
            }
            const int kcnt = __popc(km);
            int inc = kcnt;
#pragma unroll
            for (int d = 1; d < G; d <<= 1) { const int t = __shfl_up(inc, d, G); if (sub >= d) inc += t; }
            const int total = __shfl(inc, G - 1, G);
            int64_t r = rank + inc - kcnt;
            while (km) {
                const int j = __ffs(km) - 1;
                km &= km - 1;
                if (r >= skip && r < end) {
                    uint32_t word = w[0];
#pragma unroll
                    for (int k = 1; k < NW; ++k) word = (j >> 2) == k ? w[k] : word;
                    uint8_t c = (uint8_t)(word >> ((j & 3) * 8));
                    if ((fl & 1) && c >= 'a' && c <= 'z') c -= 32;
                    if (fl & 4) c = lut[c];
                    const int64_t o = r - skip;
                    out[(fl & 2) ? (take - 1 - o) : o] = c;
                }
                ++r;
            }
            rank += total;
        }
        if (ok) {
            int64_t got = rank - skip;
            got = got < 0 ? 0 : (got > take ? take : got);
            // FX_REVERSE mirrors around `take`; when the range held fewer bases than that (irregular records: the
            // reference reverses the bytes it actually got, util.c:251-261) the result sits `take - got` too high
            if ((fl & 2) && got < take && got > 0) {
                const int64_t delta = take - got;
                for (int64_t j0 = 0; j0 < got; j0 += G) {           // lanes move in lock step: load, then store
                    const int64_t j = j0 + sub;
                    uint8_t c = 0;
                    if (j < got) c = out[delta + j];
                    if (j < got) out[j] = c;
                }
            }
            if (sub == 0 && q.out_len) q.out_len[i] = got;
        }
    }
}

// The line-arithmetic path ALONE, for batches of short intervals by record id (the benchmark's 1 M x 100 bases): k_fetch
// carries the general path with it -- keep masks, ranks, per-byte stores, the mirror fix-up -- and with it 100 registers,
// four waves per SIMD.  The kernel is bound by how many random reads it keeps in flight, so here is the same arithmetic
// with nothing else: a query that it cannot answer exactly (a record that is not line-regular, lines shorter than 16
// bases, the edge of the stream, FX_RAW, a terminator that is not where the arithmetic says, white space among the bases,
// an invalid id) goes on a list, and k_fetch runs over that list afterwards (for a genome: empty).
template <int G, int NP, bool COAL = false>
__global__ __launch_bounds__(BLOCK) void k_fetch_lines(const uint8_t *__restrict__ data, int64_t gbase, int64_t n_bytes, FetchQ q, FastaTab tab,
                                                      int64_t nq, int flags_all, uint8_t *__restrict__ dst, int32_t *__restrict__ list,
                                                      int *__restrict__ list_n) {
    __shared__ uint8_t lut[256];
    // COAL: the answers of a wave's 16 queries are put together in LDS when they lie back to back in the output (dst_off[i + 1]
    // = dst_off[i] + take: what every caller that lets the library lay the answers out gets) and leave as ALIGNED 16-byte
    // stores that cover whole lines -- 100-byte answers at a 100-byte stride are otherwise unaligned stores that leave both
    // ends of most 128-byte lines partly written (PMC WRITE_SIZE 1.45 x the answers' bytes, profiles/pmc_k_span_scan.json)
    constexpr int CO_CAP = 2048;                               // bytes of answers per wave and step that the LDS path takes
    __shared__ __attribute__((aligned(16))) uint32_t s_out[COAL ? BLOCK / 64 : 1][COAL ? (CO_CAP + 32) / 4 : 1];
    constexpr int TABCAP = 512;
    __shared__ int64_t s_boff[TABCAP], s_slen[TABCAP];
    __shared__ int32_t s_llen[TABCAP], s_en[TABCAP];           // s_en = elen | line-regular << 8
    const int64_t n_seq = tab.n_seq_dev ? (*tab.n_seq_dev < tab.n_seq ? (int64_t)*tab.n_seq_dev : tab.n_seq) : tab.n_seq;
    const bool tab_lds = n_seq <= TABCAP && n_seq > 0;
    build_comp_lut(lut);
    if (tab_lds)
        for (int r = threadIdx.x; r < (int)n_seq; r += BLOCK) {
            s_boff[r] = tab.boff[r]; s_slen[r] = tab.slen[r];
            const int64_t ll = tab.llen[r];
            s_llen[r] = ll > 0x7FFFFFFFll ? 0x7FFFFFFF : (int32_t)ll;
            s_en[r] = tab.elen[r] | (tab.norm[r] << 8);
        }
    __syncthreads();
    constexpr int QPW = 64 / G;
    const int lane = lane_id(), sub = lane & (G - 1), grp = lane / G;
    const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
    int64_t d_id = 0, d_a = 0, d_b = 0, d_off = 0;
    int d_fl = flags_all;
    auto fetch_desc = [&](int64_t i) {
        if (i >= nq) return;
        d_id = q.seq_id[i]; d_a = q.start[i]; d_b = q.stop[i];
        d_off = q.dst_off[i];
        d_fl = q.qflags ? q.qflags[i] : flags_all;
    };
    fetch_desc(wave * QPW + grp);
    for (int64_t i0 = wave * QPW; i0 < nq; i0 += nwaves * QPW) {
        const int64_t i = i0 + grp;
        const bool live = i < nq;
        const int64_t id = d_id, a = d_a, b = d_b;
        uint8_t *out = dst + (live ? d_off : 0);
        const int fl = d_fl;
        fetch_desc(i + nwaves * QPW);
        bool fast = false, irregular = false;
        int64_t take = 0, in_a = 0, el = 0;
        uint32_t bpl32 = 1, r0 = 0;
        if (live && id >= 0 && id < n_seq) {
            int64_t r_boff, r_slen, llen;
            int en;
            if (tab_lds) { r_boff = s_boff[id]; r_slen = s_slen[id]; en = s_en[id]; llen = s_llen[id] == 0x7FFFFFFF ? tab.llen[id] : (int64_t)s_llen[id]; }
            else         { r_boff = tab.boff[id]; r_slen = tab.slen[id]; en = tab.elen[id] | (tab.norm[id] << 8); llen = tab.llen[id]; }
            el = en & 0xFF;
            const int64_t bpl = llen - el;
            take = b - a;
            if (a >= 0 && take > 0 && b <= r_slen && (en >> 8) != 0 && bpl >= 16 && bpl < (1ll << 31) && a + take < (1ll << 31) && !(fl & 8)) {
                bpl32 = (uint32_t)bpl;
                const uint32_t bs = (uint32_t)a / bpl32, be = (uint32_t)b / bpl32;
                in_a = r_boff + a + el * (int64_t)bs - gbase;                          // sequence.c:498-510
                const int64_t blen = take + (int64_t)(be - bs) * el;
                r0 = (uint32_t)a - bs * bpl32;
                fast = in_a >= 16 && in_a + blen + 32 <= n_bytes;
            }
        }
        // ---- COAL: do the 16 answers of this step lie back to back, dword-aligned, and fit the LDS buffer?
        bool co = false;
        uint32_t co_rel = 0, co_span = 0, co_sh = 0;           // this query's place in the wave's span; the span; (first byte of the span) & 15
        uint8_t *co_base = nullptr;
        if (COAL) {
            const uint32_t lo32 = (uint32_t)(uintptr_t)out, tk = (uint32_t)take;
            const uint32_t nxt = (uint32_t)__shfl((int)lo32, (lane + G) & 63, 64);
            const bool chain = grp == QPW - 1 || nxt == lo32 + tk;                 // the next query's answer begins where this one ends
            const bool okq = live && fast && take > 0 && take <= CO_CAP && (tk & 3u) == 0 && (lo32 & 3u) == 0 && chain;
            if (__ballot(okq) == ~0ull) {
                const uint32_t first = (uint32_t)__shfl((int)lo32, 0, 64), last = (uint32_t)__shfl((int)lo32, 63, 64), ltk = (uint32_t)__shfl((int)tk, 63, 64);
                co_span = last + ltk - first;
                if (co_span <= (uint32_t)CO_CAP) {
                    co = true;
                    co_rel = lo32 - first;
                    co_sh = first & 15u;
                    const uint64_t b0 = (uint64_t)(uintptr_t)out - co_rel;       // address of the span's first byte (lane-uniform)
                    co_base = reinterpret_cast<uint8_t *>(((uint64_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b0 >> 32)) << 32) |
                                                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b0));
                }
            }
        }
        uint32_t *const so = COAL ? s_out[threadIdx.x >> 6] : nullptr;
        if (fast) {
            const bool rev = (fl & 2) != 0;
            for (int64_t s0 = 0; s0 < take; s0 += NP * G * 16) {      // NP pieces per lane and step, all their loads first (see k_fetch)
                int64_t oc[NP], f0[NP];
                int len[NP], lead[NP];
                uint32_t tt[NP];
                const uint8_t *p1[NP];
                bool need_w[NP];
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    oc[u] = s0 + u * (G * 16) + 16 * sub;
                    const int l = (int)(take - oc[u] < 16 ? take - oc[u] : 16);
                    len[u] = l < 0 ? 0 : l;
                    f0[u] = len[u] ? (rev ? take - oc[u] - len[u] : oc[u]) : 0;
                    lead[u] = len[u] && rev ? 16 - len[u] : 0;
                    const uint32_t xx = r0 + (uint32_t)f0[u];
                    const uint32_t k = xx / bpl32;
                    tt[u] = bpl32 - (xx - k * bpl32);
                    p1[u] = data + in_a + f0[u] + el * (int64_t)k - lead[u];
                    need_w[u] = len[u] > 0 && tt[u] <= (uint32_t)len[u];
                }
                uint4 v[NP], w[NP];
#pragma unroll
                for (int u = 0; u < NP; ++u) v[u] = *reinterpret_cast<const uint4_u *>(p1[u]);
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    w[u] = make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au);
                    if (need_w[u]) w[u] = *reinterpret_cast<const uint4_u *>(p1[u] + el);
                }
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    if (len[u] <= 0) continue;
                    uint4 x = v[u];
                    const uint4 y = w[u];
                    const int ln = len[u], ld = lead[u];
                    const uint32_t t = tt[u];
                    const int T = ld + (int)t;
                    if (t < (uint32_t)ln || (t == (uint32_t)ln && f0[u] + ln < take)) {
                        const int i0b = T < 16 ? T : T - (int)el, i1b = T + 1 < 16 ? T + 1 : T + 1 - (int)el;
                        const uint4 &s0v = T < 16 ? x : y, &s1v = T + 1 < 16 ? x : y;
                        const uint32_t a01 = (i0b & 4) ? s0v.y : s0v.x, a23 = (i0b & 4) ? s0v.w : s0v.z;
                        const uint32_t g0 = (((i0b & 8) ? a23 : a01) >> ((i0b & 3) * 8)) & 0xFFu;
                        const uint32_t b01 = (i1b & 4) ? s1v.y : s1v.x, b23 = (i1b & 4) ? s1v.w : s1v.z;
                        const uint32_t g1 = (((i1b & 8) ? b23 : b01) >> ((i1b & 3) * 8)) & 0xFFu;
                        irregular |= !is_space3(g0) || (el == 2 && !is_space3(g1));
                    }
                    if (t < (uint32_t)ln) {
                        const uint32_t m0 = lowbytes_word(T, 0), m1 = lowbytes_word(T, 1), m2 = lowbytes_word(T, 2), m3 = lowbytes_word(T, 3);
                        x.x = (x.x & m0) | (y.x & ~m0); x.y = (x.y & m1) | (y.y & ~m1);
                        x.z = (x.z & m2) | (y.z & ~m2); x.w = (x.w & m3) | (y.w & ~m3);
                    }
                    {
                        const int lo_b = ld, hi_b = ld + ln;
                        uint32_t f = le20_bytes(x.x) & lowbytes_word(hi_b, 0) & ~lowbytes_word(lo_b, 0);
                        f |= le20_bytes(x.y) & lowbytes_word(hi_b, 1) & ~lowbytes_word(lo_b, 1);
                        f |= le20_bytes(x.z) & lowbytes_word(hi_b, 2) & ~lowbytes_word(lo_b, 2);
                        f |= le20_bytes(x.w) & lowbytes_word(hi_b, 3) & ~lowbytes_word(lo_b, 3);
                        irregular |= f != 0;
                    }
                    if (fl & 1) { x.x = upper4(x.x); x.y = upper4(x.y); x.z = upper4(x.z); x.w = upper4(x.w); }
                    if (fl & 4) { x.x = lut4(lut, x.x); x.y = lut4(lut, x.y); x.z = lut4(lut, x.z); x.w = lut4(lut, x.w); }
                    if (rev) x = make_uint4(__builtin_bswap32(x.w), __builtin_bswap32(x.z), __builtin_bswap32(x.y), __builtin_bswap32(x.x));
                    if (COAL && co) {                            // into the wave's span in LDS: ln is a multiple of four here
                        uint32_t *w = so + ((co_sh + co_rel + (uint32_t)oc[u]) >> 2);
                        w[0] = x.x;
                        if (ln > 4) w[1] = x.y;
                        if (ln > 8) w[2] = x.z;
                        if (ln > 12) w[3] = x.w;
                    } else store_low_bytes(out + oc[u], x, ln);
                }
            }
        }
        if (COAL && co) {
            // the span leaves as aligned 16-byte pieces: piece c covers bytes [16 c, 16 c + 16) of the buffer, whose byte co_sh is
            // the span's first; the two ends are stored dword by dword
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            uint8_t *const ab = co_base - co_sh;                // 16-byte aligned
            const uint32_t lo_b = co_sh, hi_b = co_sh + co_span;
            for (uint32_t c = (uint32_t)lane; c * 16u < hi_b; c += 64u) {
                const uint32_t a0 = c * 16u;
                const uint4 v = *reinterpret_cast<const uint4 *>(so + (a0 >> 2));
                if (a0 >= lo_b && a0 + 16u <= hi_b) *reinterpret_cast<uint4 *>(ab + a0) = v;
                else {
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (a0 + 4u * k >= lo_b && a0 + 4u * k + 4u <= hi_b) *reinterpret_cast<uint32_t *>(ab + a0 + 4u * k) = w[k];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (the buffer is written again in the next step)
        }
        const unsigned long long ib = __ballot(irregular);
        constexpr unsigned long long gmask = G >= 64 ? ~0ull : ((1ull << (G & 63)) - 1ull);
        const bool redo = live && (!fast || ((ib >> (grp * G)) & gmask) != 0);
        if (sub == 0 && live) {
            if (redo) list[atomicAdd(list_n, 1)] = (int32_t)i;       // (the general kernel writes its answer and out_len)
            else if (q.out_len) q.out_len[i] = take;
        }
    }
}

// ---------------------------------------------------------------- one getter at a time (fx_fetch_one)
// The reference's object API makes one call per getter (fa[name][a:b].seq: pyfastx_index_fill_cache + a copy, 4 us of
// pread and memcpy).  A kernel launch and a stream synchronisation per getter cost 16 us before any byte moves, so single
// getters are served by a RESIDENT kernel instead: one wave that polls a mailbox in pinned host memory, answers the
// request it finds there (the general path of k_fetch on one range, staged in LDS and written to pinned memory in
// 16-byte pieces), acknowledges, and leaves by itself after `idle_limit` empty polls (it is a guest on the device: a
// hipFree or a device-wide synchronisation elsewhere waits at most that long).  The host relaunches it on demand.
struct Mailbox {
    // the request: ONE 64-byte line, read by the device with one load (eight lanes, eight bytes each -- a PCIe round trip
    // per field was most of the first version's 13 us); head and tail carry the request number and are written last and
    // first: a torn read shows different numbers and is repeated
    unsigned long long head;         // host -> device: number of the request (monotonic), written LAST
    long long off, blen, skip, take;
    long long flags_quit;            // flags | quit << 32
    unsigned long long pad0;
    unsigned long long tail;         // = head, written FIRST
    // the answer, its own line
    unsigned long long ack;          // device -> host: request number << 20 | bytes of the answer, written after the bytes are visible
    unsigned int state;              // 1 serving, 2 about to leave (the host then watches the stream), 0 gone
    unsigned int pad1[13];
};
constexpr int MB_OUT = 65536;        // longest answer the resident kernel stages (longer ones take the launch path)

// (relaxed: the request's fields arrive in the same load, and everything else the kernel reads is the immutable blob --
// an acquire here would invalidate caches under every poll, for every kernel that runs beside this one)
__device__ __forceinline__ unsigned long long sys_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ long long bcast64(long long v, int l) {
    return ((long long)__shfl((int)(v >> 32), l, 64) << 32) | (unsigned)__shfl((int)(v & 0xFFFFFFFFll), l, 64);
}
__global__ __launch_bounds__(64) void k_mailbox(const uint8_t *__restrict__ data, int64_t gbase, int64_t n_bytes, Mailbox *mb,
                                               uint8_t *out_host, unsigned long long served, int idle_limit) {
    __shared__ uint8_t lut[256];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[MB_OUT];
    build_comp_lut(lut);
    __syncthreads();
    const int lane = threadIdx.x;
    const unsigned long long *line = reinterpret_cast<const unsigned long long *>(mb);
    int idle = 0;
    bool leaving = false;
    for (;;) {
        const long long w = lane < 8 ? (long long)sys_load(line + lane) : 0;        // the whole request line, one load
        const unsigned long long r = (unsigned long long)bcast64(w, 0), rt = (unsigned long long)bcast64(w, 7);
        if (r == served || r != rt) {
            if (leaving) break;                              // announced, looked once more, still nothing
            if (++idle < idle_limit) { __builtin_amdgcn_s_sleep(1); continue; }
            // nothing for a while: announce the departure, then look once more (the host may have posted in between)
            if (lane == 0) { __hip_atomic_store(&mb->state, 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); __threadfence_system(); }
            leaving = true;
            continue;
        }
        if (leaving) { leaving = false; if (lane == 0) __hip_atomic_store(&mb->state, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        long long f[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) f[k] = bcast64(w, 1 + k);
        const long long fq = bcast64(w, 5);
        const int fl = (int)(fq & 0xFFFFFFFFll), quit = (int)(fq >> 32);
        if (quit) break;
        const int64_t skip = f[2], take = f[3] < MB_OUT ? f[3] : MB_OUT;
        // ---- the general path of k_fetch for one range, 1 KiB of the stream per step (jump_table, util.c:157-194)
        int64_t lo = f[0] - gbase, hi = lo + f[1];
        if (lo < 0) lo = 0;
        if (hi > n_bytes) hi = n_bytes;
        const int64_t end = skip + take;
        int64_t rank = 0;
        for (int64_t p = lo & ~(int64_t)15; p < hi && rank < end; p += 64 * 16) {
            const int64_t pp = p + (int64_t)lane * 16;
            uint32_t w[4] = {0, 0, 0, 0};
            if (pp < hi) { const uint4 t = *reinterpret_cast<const uint4 *>(data + pp); w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w; }
            int64_t a0 = lo - pp, a1 = hi - pp;
            a0 = a0 < 0 ? 0 : (a0 > 16 ? 16 : a0);
            a1 = a1 < 0 ? 0 : (a1 > 16 ? 16 : a1);
            uint32_t km = (a1 > a0) ? (((1u << a1) - 1u) & ~((1u << a0) - 1u)) & 0xFFFFu : 0u;
            if (!(fl & 8)) {
                uint32_t sp = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    sp |= flags4(zero_bytes(w[k] ^ 0x0A0A0A0Au) | zero_bytes(w[k] ^ 0x0D0D0D0Du) | zero_bytes(w[k] ^ 0x20202020u)) << (4 * k);
                km &= ~sp;
            }
            const uint32_t kcnt = __popc(km), inc = wave_incl_scan(kcnt);
            const uint32_t total = (uint32_t)__shfl((int)inc, 63, 64);
            int64_t rr = rank + inc - kcnt;
            while (km) {
                const int j = __ffs(km) - 1;
                km &= km - 1;
                if (rr >= skip && rr < end) {
                    uint8_t c = (uint8_t)(w[j >> 2] >> ((j & 3) * 8));
                    if ((fl & 1) && c >= 'a' && c <= 'z') c -= 32;
                    if (fl & 4) c = lut[c];
                    const int64_t o = rr - skip;
                    s_out[(fl & 2) ? (take - 1 - o) : o] = c;
                }
                ++rr;
            }
            rank += total;
        }
        int64_t got = rank - skip;
        got = got < 0 ? 0 : (got > take ? take : got);
        __syncthreads();
        const int64_t delta = ((fl & 2) && got < take) ? take - got : 0;       // a reversed answer shorter than asked for sits too high
        for (int64_t j = (int64_t)lane * 16; j < got; j += 64 * 16) {
            uint4 v;
            if (delta == 0) v = *reinterpret_cast<const uint4 *>(s_out + j);
            else { uint8_t t[16]; for (int k = 0; k < 16; ++k) t[k] = j + k < got ? s_out[delta + j + k] : 0; v = *reinterpret_cast<const uint4 *>(t); }
            *reinterpret_cast<uint4 *>(out_host + j) = v;       // (the staging buffer is 16 bytes longer than MB_OUT)
        }
        __threadfence_system();                              // the bytes are visible to the host ...
        __syncthreads();
        if (lane == 0) __hip_atomic_store(&mb->ack, (r << 20) | (unsigned long long)got, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // ... before their count
        served = r;
        idle = 0;
    }
    if (lane == 0) __hip_atomic_store(&mb->state, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// In-place reverse / complement of one buffer (pyfastx.reverse_complement, module.c:44-59).
__global__ __launch_bounds__(BLOCK) void k_revcomp(uint8_t *__restrict__ buf, int64_t n, int mode) {
    __shared__ uint8_t lut[256];
    build_comp_lut(lut);
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    const int64_t half = (mode & 2) ? (n + 1) / 2 : n;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < half; i += stride) {
        if (mode & 2) {
            const int64_t j = n - 1 - i;
            uint8_t a = buf[i], b = buf[j];
            if (mode & 4) { a = lut[a]; b = lut[b]; }
            buf[i] = b; buf[j] = a;
        } else if (mode & 4) {
            buf[i] = lut[buf[i]];
        }
    }
}

// out[i] = col[idx[i]], -1 for an index outside [0, n)
__global__ __launch_bounds__(BLOCK) void k_gather_i64(const int64_t *__restrict__ col, int64_t n, const int64_t *__restrict__ idx,
                                                     int64_t nq, int64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < nq) { const int64_t k = idx[i]; out[i] = (k >= 0 && k < n) ? col[k] : -1; }
}

// FASTQ read fetch (read.c:37-45,152-167,237-278).  16 lanes per read, 4 reads per wave; a lane moves 16
// bytes of the sequence line and 16 of the quality line per step with unaligned 16-byte loads / stores
// (a 150-base read is one step).  quali = qual - phred as int8, four bytes at a time.  The chain of
// dependent round trips (id -> table row -> bytes) is software-pipelined: ids are fetched two iterations
// ahead and table rows one iteration ahead, so only the byte loads are on the critical path.
__device__ __forceinline__ uint32_t sub_bytes(uint32_t x, uint32_t y) {        // per-byte x - y (mod 256), y < 0x80 in every byte
    return ((x | 0x80808080u) - y) ^ (~x & 0x80808080u);
}
__global__ __launch_bounds__(BLOCK) void k_fastq_fetch(const uint8_t *__restrict__ data, int64_t gbase, int64_t n_bytes,
                                                      const int64_t *__restrict__ rlen, const int64_t *__restrict__ soff,
                                                      const int64_t *__restrict__ qoff, int64_t n_reads,
                                                      const int64_t *__restrict__ ids, int64_t nq, int phred, int flags,
                                                      uint8_t *__restrict__ seq, uint8_t *__restrict__ qual,
                                                      int8_t *__restrict__ quali, const int64_t *__restrict__ dst_off) {
    __shared__ uint8_t lut[256];
    build_comp_lut(lut);
    __syncthreads();
    const int lane = lane_id(), sub = lane & 15, grp = lane >> 4;
    const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const int64_t stride = (((int64_t)gridDim.x * BLOCK) >> 6) * 4;
    const uint32_t ph4 = (uint32_t)(phred & 0x7F) * 0x01010101u;
    const bool rev = (flags & 2) != 0;
    // pipeline registers: id two iterations ahead, table row one iteration ahead
    int64_t i = wave * 4 + grp;
    auto get_id = [&](int64_t q) -> int64_t { return q < nq ? (ids ? ids[q] : q) : -1; };   // ids == null: the arrays are per query already
    int64_t id1 = get_id(i), id2 = get_id(i + stride);
    // (the row by value: with the four fields as variables a lambda assigned through references, the compiler kept two of them
    // in scratch memory -- a store and a load per query on the path that is all latency)
    struct Row { int64_t n, so, qo, d; };
    auto get_row = [=](int64_t id, int64_t q) -> Row {
        Row r{-1, 0, 0, 0};
        if (id >= 0 && id < n_reads) { r.n = rlen[id]; r.so = soff[id] - gbase; r.qo = qoff[id] - gbase; r.d = dst_off[q]; }
        return r;
    };
    Row row = get_row(id1, i);
    for (; i - grp < nq; i += stride) {                       // wave-uniform trip count
        const int64_t n = row.n, so = row.so, qo = row.qo, d = row.d;
        id1 = id2;
        id2 = get_id(i + 2 * stride);
        row = get_row(id1, i + stride);
        if (n <= 0) continue;
        const bool inside = so >= 16 && qo >= 16 && so + n + 16 <= n_bytes && qo + n + 16 <= n_bytes;   // room for whole 16-byte accesses
        for (int64_t s0 = 0; s0 < n; s0 += 256) {
            const int64_t oc = s0 + 16 * sub;                 // this lane's output chunk [oc, oc + len)
            const int len = (int)(n - oc < 16 ? n - oc : 16);
            if (len <= 0) continue;
            if (inside) {
                const uint4 qv = *reinterpret_cast<const uint4_u *>(data + qo + oc);
                if (seq) {
                    // reverse strand: forward bytes [n - oc - len, n - oc), reversed; the partial chunk is the head of the read
                    const int64_t f0 = rev ? n - oc - len : oc;
                    const int lead = rev ? 16 - len : 0;
                    uint4 v = *reinterpret_cast<const uint4_u *>(data + so + f0 - lead);
                    if (flags & 4) { v.x = lut4(lut, v.x); v.y = lut4(lut, v.y); v.z = lut4(lut, v.z); v.w = lut4(lut, v.w); }
                    if (rev) v = make_uint4(__builtin_bswap32(v.w), __builtin_bswap32(v.z), __builtin_bswap32(v.y), __builtin_bswap32(v.x));
                    store_low_bytes(seq + d + oc, v, len);
                }
                if (qual) store_low_bytes(qual + d + oc, qv, len);
                if (quali) store_low_bytes(reinterpret_cast<uint8_t *>(quali) + d + oc,
                                           make_uint4(sub_bytes(qv.x, ph4), sub_bytes(qv.y, ph4), sub_bytes(qv.z, ph4), sub_bytes(qv.w, ph4)), len);
            } else {                                          // reads at the very edge of the blob: byte by byte
                for (int k = 0; k < len; ++k) {
                    const int64_t j = oc + k;
                    if (seq) {
                        uint8_t c = data[so + j];
                        if (flags & 4) c = lut[c];
                        seq[d + (rev ? (n - 1 - j) : j)] = c;
                    }
                    const uint8_t qc = data[qo + j];
                    if (qual) qual[d + j] = qc;
                    if (quali) quali[d + j] = (int8_t)((int)(signed char)qc - phred);
                }
            }
        }
    }
}

}  // namespace fx
