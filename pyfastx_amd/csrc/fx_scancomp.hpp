// fx_scancomp.hpp -- the FASTA index scan and the letter composition in ONE read of the stream (gfx950).
//
// The reference counts the letters of every record in a second pass over the file (fasta.c:851-961, after
// index.c:230-372); so did this engine until round 2 (k_span_scan, then k_fasta_comp: 2 x the stream).  A build that is
// asked for the composition as well (fx_fasta_build_begin, bit 1 of the flags: Fasta(..., full_index=True)) runs
// k_scan_comp in place of k_span_scan<0>:
//
//   a wave takes a RUN of gpw consecutive 4 KiB granules (the run geometry of k_fasta_comp), two granules in flight; every
//   granule goes through granule_body<0> (the scan's summary, exactly as k_span_scan writes it) and, from the same
//   registers, through comp_add_granule<true> (the bit-plane counters of fx_comp.hpp); at the end of the run the eleven
//   class counts (A C G T N upper / lower case, '\r') go to the run's record of 16 words, with word 15 = 1 when the run
//   held no header line and no byte outside the expected set.
//
// Which record a run belongs to is not known while the stream is read -- the header prefixes come later in the build.
// k_comp_attribute settles that afterwards, one thread per run: a run that lies inside one record's sequence block gives
// its counts to that record's row (gathered per workgroup in LDS first: neighbouring runs belong to the same
// chromosome); every other run -- header lines, record boundaries, unexpected bytes, the tail of the stream -- goes on the
// edge list that k_fasta_comp<false> counts from the bytes, as before.  For a genome that is a few hundred runs of 64 k.
// The scan kernel alone is HBM-bound (0.44 ms for 3 GB) and the counting alone VALU-bound (0.52 ms); together they are
// VALU-bound: see DESIGN.md for the measured sum.
#pragma once
#include "fx_spanscan.hpp"
#include "fx_comp.hpp"

namespace fx {

constexpr int RUN_WORDS = 16;                   // slots 0..5: A C G T N \r upper case, 8..12: a c g t n, 15: counts valid

#ifdef FX_SC_WPE
__attribute__((amdgpu_waves_per_eu(FX_SC_WPE, FX_SC_WPE)))
#endif
__global__ __launch_bounds__(COMP_WPB * 64) void k_scan_comp(const uint8_t *__restrict__ data, int64_t n, int prev_byte, int is_last,
                                                             int64_t g_end, GranPk *__restrict__ out, GranList hgl, int gpw,
                                                             uint32_t *__restrict__ run_cnt) {
    __shared__ uint32_t hl_n, hl_done, hl_g[COMP_WPB * COMP_GPW];
    if (threadIdx.x == 0) { hl_n = 0; hl_done = 0; }
    __syncthreads();
    const int lane = lane_id(), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t wave = (int64_t)blockIdx.x * COMP_WPB + wv;
    const int64_t gfirst = wave * gpw;
    const int cnt = gfirst < g_end ? (int)(g_end - gfirst < gpw ? g_end - gfirst : gpw) : 0;
    CompState s;
    s.rec = COMP_NONE; s.rare = false;
    comp_reset(s);
    uint32_t L = 0, hsum = 0;
    bool rare = false;
    uint4 buf[COMP_DEPTH][4];
#pragma unroll
    for (int k = 0; k < COMP_DEPTH; ++k) if (k < cnt) granule_load<true>(buf[k], data, n, is_last, gfirst + k);
    for (int i = 0; i < cnt; i += COMP_DEPTH) {
#pragma unroll
        for (int k = 0; k < COMP_DEPTH; ++k) {
            if (i + k >= cnt) break;
            const int64_t g = gfirst + i + k;
            // the counters first: what they find out about the bytes of each row (anything that is no letter of the
            // expected set and no line end) stands in for the scan's own '>' pre-filter
            uint32_t odd[GR_ROWS];
            if (__ballot(comp_add_granule<true>(s, buf[k], 0, FX_GRAN, odd) != 0)) rare = true;
            const uint32_t h_w = granule_body<0>(buf[k], data, prev_byte, g, out, L, odd);
            if (h_w) {
                hsum += h_w;
                if (lane == 0) hl_g[atomicAdd(&hl_n, 1u)] = (uint32_t)g;
            }
            if (i + k + COMP_DEPTH < cnt) granule_load<true>(buf[k], data, n, is_last, g + COMP_DEPTH);
            __builtin_amdgcn_sched_barrier(0);              // one granule at a time: interleaving them only costs registers
        }
    }
    if (cnt > 0) {
        uint32_t mine = 0;                                   // lane c: upper-case count of class c, lane 8 + c: lower-case count
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const uint32_t tot = planes_count(s.all, c);
            const uint32_t lo = c < 5 ? planes_count(s.low, c) : 0u;
            const uint32_t nu = wave_total(tot - lo);
            if (lane == c) mine = nu;
            if (c < 5) {
                const uint32_t nl = wave_total(lo);
                if (lane == 8 + c) mine = nl;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (lane == 15) mine = (cnt == gpw && hsum == 0 && !rare) ? 1u : 0u;
        if (lane < RUN_WORDS) run_cnt[wave * RUN_WORDS + lane] = mine;
    }
    // the granules of the workgroup that hold header lines: one append (the last wave to arrive does it), as k_span_scan
    if (lane == 0) {
        __threadfence_block();
        if (atomicAdd(&hl_done, 1u) == (uint32_t)COMP_WPB - 1u) {
            const uint32_t c = hl_n;
            if (c) {
                const uint32_t base = atomicAdd(hgl.count, c);
                for (uint32_t k = 0; k < c; ++k) hgl.g[base + k] = hl_g[k];
            }
        }
    }
}

// one thread per run: its counts to the record that owns it, or the run to the edge list (see the head of the file)
constexpr int ATTR_BLOCK = 256;
__global__ __launch_bounds__(ATTR_BLOCK) void k_comp_attribute(const uint32_t *__restrict__ run_cnt, int64_t nruns, int gpw, int64_t n,
                                                              int64_t gbase, const int64_t *__restrict__ boff, int64_t n_hdr,
                                                              const int64_t *__restrict__ hdr_prefix, int64_t ngran, int64_t lead_from,
                                                              int32_t *__restrict__ edge_list, unsigned long long *__restrict__ comp) {
    __shared__ uint32_t blk_cnt[RUN_WORDS];
    if (threadIdx.x < RUN_WORDS) blk_cnt[threadIdx.x] = 0;
    __syncthreads();
    int64_t nreal = (n + FX_GRAN - 1) / FX_GRAN;             // granules that hold bytes
    if (nreal > ngran) nreal = ngran;
    const int64_t rmin = lead_from >= 0 ? -1 : 0;            // lowest record index that is counted
    const int64_t gblock = (int64_t)blockIdx.x * ATTR_BLOCK * gpw;
    const int64_t blk_rec = gblock < nreal ? hdr_prefix[gblock] - 1 : COMP_NONE;
    const int64_t run = (int64_t)blockIdx.x * ATTR_BLOCK + threadIdx.x;
    const int64_t gfirst = run * gpw;
    bool pure = false, mine_is_blk = false;
    int64_t r0 = COMP_NONE;
    if (run < nruns && gfirst < nreal) {
        const int cnt = (int)(nreal - gfirst < gpw ? nreal - gfirst : gpw);
        const bool candidate = cnt == gpw && (gfirst + cnt) * (int64_t)FX_GRAN <= n && run_cnt[run * RUN_WORDS + 15] != 0;
        if (candidate) {
            r0 = hdr_prefix[gfirst] - 1;
            pure = hdr_prefix[gfirst + cnt] == r0 + 1 && r0 >= rmin && gbase + gfirst * (int64_t)FX_GRAN >= (r0 >= 0 ? boff[r0] : lead_from);
        }
        if (!pure) edge_list[1 + atomicAdd(&edge_list[0], 1)] = (int32_t)run;
        mine_is_blk = pure && r0 == blk_rec;
    }
    // the runs of the workgroup's own record (nearly all of them): summed per wave, one LDS atomic per wave and class
#pragma unroll
    for (int slot = 0; slot < 13; ++slot) {
        if (slot == 6 || slot == 7) continue;
        const uint32_t v = pure ? run_cnt[run * RUN_WORDS + slot] : 0u;
        const uint32_t w = wave_total(mine_is_blk ? v : 0u);
        if (w && lane_id() == 0) atomicAdd(&blk_cnt[slot], w);
        if (v && !mine_is_blk) atomicAdd(&comp[(r0 >= 0 ? r0 : n_hdr) * 128 + comp_symbol(slot)], (unsigned long long)v);
    }
    __syncthreads();
    if (threadIdx.x < 13 && blk_rec >= rmin) {
        const uint32_t v = blk_cnt[threadIdx.x];
        if (v) atomicAdd(&comp[(blk_rec >= 0 ? blk_rec : n_hdr) * 128 + comp_symbol((int)threadIdx.x)], (unsigned long long)v);
    }
}

}  // namespace fx
