// fxgpu.hip -- libfxgpu.so: C ABI (include/fxgpu.h) over the gfx950 kernels in
// fx_kernels.hpp.  Host side: staging (file -> pinned -> HBM), launch
// sequencing, table read-back.  No CPU fallback: every compute entry point
// needs a HIP device and fails with FX_EDEVICE otherwise.
#include <hip/hip_runtime.h>
#include <zlib.h>
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <map>
#include <memory>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fxgpu.h"
#include "fx_kernels.hpp"
#include "fx_spanscan.hpp"
#include "fx_fastq.hpp"
#include "fx_comp.hpp"
#include "fx_fastq_stream.hpp"
#include "fx_scancomp.hpp"
#include "fx_names.hpp"
#include "fx_inflate.hpp"
#include "fx_inflate_par.hpp"
#include "fx_bgzf_walk.hpp"
#include "fx_fxi.hpp"
#include "fx_fxi_dev.hpp"
#include "fx_pgzip.hpp"
#include "fx_sort.hpp"
#include "fx_kseq.hpp"

using namespace fx;

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return fail(e__ == hipErrorOutOfMemory ? FX_ENOMEM : FX_EDEVICE, "%s: %s (%s:%d)", #expr,  \
                        hipGetErrorString(e__), __FILE__, __LINE__);                                   \
    } while (0)

extern "C" const char *fx_last_error(void) { return g_err.c_str(); }
// the last fx_open_file of a plain file by this thread: seconds for the device allocation (the FIRST block of tens of GB a
// process asks the driver for takes seconds, the next one of that size microseconds) and for page cache -> pinned -> HBM
static thread_local double g_open_laps[2] = {0.0, 0.0};
// ... and of the last FASTQ build by this thread (fx_fastq_build / _build_comp), seconds: the sample of the stream and its wait, the
// allocations + launches of the count pass, the wait for it, the recount of rejected runs, plan + table allocation, row kernels + wait
static thread_local double g_build_laps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
extern "C" int fx_build_laps(double *out8) {
    if (out8) memcpy(out8, g_build_laps, sizeof g_build_laps);
    return FX_OK;
}
struct LapClock {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(int i) { const auto n = std::chrono::steady_clock::now(); g_build_laps[i] = std::chrono::duration<double>(n - t).count(); t = n; }
};
extern "C" int fx_open_laps(double *alloc_s, double *stage_s) {
    if (alloc_s) *alloc_s = g_open_laps[0];
    if (stage_s) *stage_s = g_open_laps[1];
    return FX_OK;
}
extern "C" const char *fx_version(void) { return "fxgpu 0.1.0 (gfx950)"; }
extern "C" int fx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------ handle
static void scratch_trim();              // the idle blocks of the scratch pool below go back to the driver
static hipError_t dev_malloc(void **p, size_t bytes) {     // hipMalloc; when memory is short the pool is emptied and it is tried again
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); scratch_trim(); e = hipMalloc(p, bytes); }
    return e;
}

namespace fx { hipError_t pool_malloc(void **p, size_t bytes) { return dev_malloc(p, bytes); } }   // fx_sort.hip allocates through it

template <class T> struct DevBuf {      // grow-only device array: rebuilds reuse the allocation
    T *p = nullptr;
    int64_t n = 0, cap = 0;
    int alloc(int64_t count) {
        if (count <= cap && p) { n = count; return FX_OK; }
        release();
        if (count <= 0) return FX_OK;
        hipError_t e = dev_malloc((void **)&p, (size_t)count * sizeof(T));
        if (e != hipSuccess) { p = nullptr; return fail(FX_ENOMEM, "hipMalloc(%lld B): %s", (long long)(count * sizeof(T)), hipGetErrorString(e)); }
        n = cap = count;
        return FX_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = cap = 0; }
    ~DevBuf() { release(); }
};

// Scratch of an open (the compressed bytes of a BGZF file, its match map, ...): hundreds of MB that live for tens of
// milliseconds.  hipFree waits for the device and unmaps -- 20 ms for 1.4 GB -- and the hipMalloc of the next open maps
// again, so released blocks are kept (per device, up to FX_SCRATCH_CACHE_MB, default 24576 since round 6 -- the temporaries of a
// 10^8-read index file with its sort are 14 GB on top of what the opens before left --; 0 = off) and handed to the
// next request they fit (at most twice its size).
// Which of a device's LARGE idle blocks (each larger than the whole FX_SCRATCH_CACHE_MB limit: the blobs of closed streams)
// stay when one more of `cap` bytes comes back, `keep` bytes of them allowed in all: the new block stays if it fits beside
// the others; else the smallest ones go until it does -- a block smaller than all that would have to go is itself the
// one that goes.  -> 1: keep the new block (evict[i] = 1: idle block i goes), 0: drop it.  Pure: tests pin it (fx_scratch_policy).
static int big_block_policy(const int64_t *caps, int n, int64_t cap, int64_t keep, int32_t *evict) {
    for (int i = 0; i < n; ++i) evict[i] = 0;
    if (cap > keep) return 0;
    int64_t held = 0;
    for (int i = 0; i < n; ++i) held += caps[i];
    while (held + cap > keep) {
        int at = -1;
        for (int i = 0; i < n; ++i) if (!evict[i] && (at < 0 || caps[i] < caps[at])) at = i;
        if (at < 0 || caps[at] >= cap) { for (int i = 0; i < n; ++i) evict[i] = 0; return 0; }      // what would go is no smaller than the newcomer
        evict[at] = 1;
        held -= caps[at];
    }
    return 1;
}
extern "C" int fx_scratch_policy(const int64_t *idle_caps, int n, int64_t cap, int64_t keep_bytes, int32_t *evict) {
    if (n < 0 || (n > 0 && (!idle_caps || !evict)) || cap < 0 || keep_bytes < 0) return fail(FX_EINVAL, "bad argument");
    return big_block_policy(idle_caps, n, cap, keep_bytes, evict);
}

struct ScratchPool {
    struct Block { void *p; size_t cap; int dev; std::chrono::steady_clock::time_point since; };
    std::mutex mu;
    std::vector<Block> idle;
    // Blocks LARGER than the whole limit idle here (round 5: one per device; round 6: several, under a cap and a time to live):
    // the blobs of the last large streams.  hipFree of tens of GB returns at once, but the driver takes the block down in the
    // background, and a hipMalloc that comes before it is done WAITS for it -- 2.65 s for 34.8 GB, now and then
    // (tools/first_open_probe.py: open / close / open of one file in a fresh process: 0.3 ms, 2.6 ms, 2 650 ms; 5.9 s in front of
    // a constructor on a 0.7 GB file in one bench run of round 4).  So a large block is not freed when its stream closes: the
    // next open of that size class takes it (no driver call at all), and an open of ANOTHER size finds no free in flight to
    // wait behind.  What bounds them: FX_SCRATCH_KEEP_BIG_MB per device (default: half the device's memory; 0: keep none),
    // big_block_policy above; FX_SCRATCH_BIG_TTL_S (default 300; 0: no limit) -- a block idle for longer goes back at the next
    // call into the pool, so a process that has moved on does not sit on tens of GB other tenants of the GPU could use --;
    // fx_release_scratch, a failed allocation and windows.hbm_budget give everything back at once.  In a process that shares
    // its GPU (several ranks on one device) call fx_release_scratch after closing large files, or set FX_SCRATCH_KEEP_BIG_MB=0.
    std::vector<Block> big;
    size_t held = 0;
    const size_t limit = [] { const char *e = getenv("FX_SCRATCH_CACHE_MB"); return (size_t)(e ? std::max(0, atoi(e)) : 24576) << 20; }();
    const double big_ttl = [] { const char *e = getenv("FX_SCRATCH_BIG_TTL_S"); return e ? atof(e) : 300.0; }();
    int64_t keep_big(int dev) {
        if (const char *e = getenv("FX_SCRATCH_KEEP_BIG_MB")) return (int64_t)std::max(0ll, atoll(e)) << 20;
        static std::mutex m;
        static std::vector<int64_t> half;                    // per device: half its memory
        std::lock_guard<std::mutex> g(m);
        if ((int)half.size() <= dev) half.resize((size_t)dev + 1, -1);
        if (half[(size_t)dev] < 0) {
            size_t fr = 0, tot = 0;
            int cur = 0;
            (void)hipGetDevice(&cur);
            if (hipSetDevice(dev) == hipSuccess && hipMemGetInfo(&fr, &tot) == hipSuccess) half[(size_t)dev] = (int64_t)(tot / 2);
            else half[(size_t)dev] = 0;
            (void)hipSetDevice(cur);
        }
        return half[(size_t)dev];
    }
    // large blocks that have idled past their time to live -> out (freed by the caller, outside the lock)
    void expire_locked(std::vector<Block> &out) {
        if (big_ttl <= 0 || big.empty()) return;
        const auto now = std::chrono::steady_clock::now();
        for (size_t i = 0; i < big.size();)
            if (std::chrono::duration<double>(now - big[i].since).count() > big_ttl) { out.push_back(big[i]); big.erase(big.begin() + (long)i); }
            else ++i;
    }
    void *get(int dev, size_t bytes, size_t *cap) {
        std::vector<Block> old;
        {
            std::lock_guard<std::mutex> g(mu);
            int best = -1;
            for (int i = 0; i < (int)idle.size(); ++i)
                if (idle[i].dev == dev && idle[i].cap >= bytes && idle[i].cap <= 2 * bytes + (1u << 20) && (best < 0 || idle[i].cap < idle[best].cap)) best = i;
            if (best >= 0) {
                Block b = idle[best];
                idle.erase(idle.begin() + best);
                held -= b.cap;
                *cap = b.cap;
                return b.p;
            }
            best = -1;
            for (int i = 0; i < (int)big.size(); ++i)
                if (big[i].dev == dev && big[i].cap >= bytes && big[i].cap <= 2 * bytes + (1u << 20) && (best < 0 || big[i].cap < big[best].cap)) best = i;
            if (best >= 0) {
                Block b = big[best];
                big.erase(big.begin() + best);
                *cap = b.cap;
                return b.p;
            }
            expire_locked(old);
        }
        void *p = nullptr;
        static const bool trace = [] { const char *e = getenv("FX_TRACE_ALLOC"); return e && atoi(e) != 0; }();
        const auto t0 = std::chrono::steady_clock::now();
        // (the new block FIRST, what has expired afterwards: a hipMalloc behind a large hipFree waits for the driver)
        if (hipMalloc(&p, bytes) != hipSuccess) {            // make room: give the idle blocks back and try once more
            (void)hipGetLastError();
            for (auto &b : old) (void)hipFree(b.p);
            old.clear();
            trim();
            if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        }
        for (auto &b : old) (void)hipFree(b.p);
        if (trace) fprintf(stderr, "[fxgpu] scratch miss: hipMalloc(%zu) %.2f ms\n", bytes, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        *cap = bytes;
        return p;
    }
    void put(int dev, void *p, size_t cap) {
        std::vector<Block> drop;
        {
            std::lock_guard<std::mutex> g(mu);
            const auto now = std::chrono::steady_clock::now();
            if (held + cap <= limit) { idle.push_back(Block{p, cap, dev, now}); held += cap; return; }
            bool kept = false;
            if (limit && cap > limit) {                      // larger than the whole limit: one of the device's large idle blocks, if the policy says so
                std::vector<int64_t> caps;
                std::vector<int> where;
                for (int i = 0; i < (int)big.size(); ++i) if (big[i].dev == dev) { caps.push_back((int64_t)big[i].cap); where.push_back(i); }
                std::vector<int32_t> ev(caps.size() + 1, 0);
                kept = big_block_policy(caps.data(), (int)caps.size(), (int64_t)cap, keep_big(dev), ev.data()) != 0;
                if (kept) {
                    for (int k = (int)where.size() - 1; k >= 0; --k)
                        if (ev[(size_t)k]) { drop.push_back(big[(size_t)where[(size_t)k]]); big.erase(big.begin() + where[(size_t)k]); }
                    big.push_back(Block{p, cap, dev, now});
                }
            }
            if (!kept) drop.push_back(Block{p, cap, dev, now});
            expire_locked(drop);
        }
        for (auto &b : drop) {
            if (getenv("FX_TRACE_ALLOC")) fprintf(stderr, "[fxgpu] scratch full: hipFree(%zu)\n", b.cap);
            (void)hipFree(b.p);
        }
    }
    void trim() {
        std::vector<Block> v, w;
        { std::lock_guard<std::mutex> g(mu); v.swap(idle); w.swap(big); held = 0; }
        for (auto &b : v) (void)hipFree(b.p);
        for (auto &b : w) (void)hipFree(b.p);
    }
};
static ScratchPool g_scratch;
static void scratch_trim() { g_scratch.trim(); }
namespace fx {                                              // (fx_sort.hip takes its buffers from the pool)
void *scratch_get(int device, size_t bytes, size_t *cap) { return g_scratch.get(device, bytes, cap); }
void scratch_put(int device, void *p, size_t cap) { g_scratch.put(device, p, cap); }
}
extern "C" int fx_release_scratch(void) { g_scratch.trim(); return FX_OK; }
template <class T> struct ScratchBuf {                      // device array out of the pool; returned to it when it goes out of scope
    T *p = nullptr;
    size_t cap_bytes = 0;
    int dev = 0;
    hipStream_t stream = nullptr;                            // what uses the block runs on this stream: waited for before the block changes hands
    int alloc(int device, int64_t count, hipStream_t st = nullptr) {
        release();
        if (count <= 0) return FX_OK;
        dev = device;
        stream = st;
        p = (T *)g_scratch.get(device, (size_t)count * sizeof(T), &cap_bytes);
        if (!p) return fail(FX_ENOMEM, "hipMalloc(%lld B) failed", (long long)(count * sizeof(T)));
        return FX_OK;
    }
    void release() {
        if (p) { (void)hipStreamSynchronize(stream); g_scratch.put(dev, p, cap_bytes); }
        p = nullptr; cap_bytes = 0;
    }
    ~ScratchBuf() { release(); }
};

// ------------------------------------------------------------ kernel timing
// Optional per-kernel HIP-event timing on the handle's own stream (bench.py's
// roofline leg reads it; off by default so the hot path records no events).
// The tables of k_bgzf_crc: the byte table of the reflected polynomial 0xEDB88320, the matrices of "append 2^j zero
// bytes" (column b = image of bit b; squared up from the one-zero-bit operator, as zlib's crc32_combine does), and the
// seven the kernel's hot loop uses expanded into byte-indexed tables.
static void crc_tables(CrcTables *T) {
    for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; T->crc[0][i] = c; }
    for (int j = 1; j < 4; ++j)                              // slicing: crc[j][i] = crc[0] applied to i followed by j zero bytes
        for (uint32_t i = 0; i < 256; ++i) T->crc[j][i] = (T->crc[j - 1][i] >> 8) ^ T->crc[0][T->crc[j - 1][i] & 0xFFu];
    uint32_t a[32], b[32];
    a[0] = 0xEDB88320u;                                      // one zero bit
    for (int n = 1; n < 32; ++n) a[n] = 1u << (n - 1);
    auto times = [](const uint32_t *mat, uint32_t vec) { uint32_t s = 0; for (int i = 0; vec; vec >>= 1, ++i) if (vec & 1) s ^= mat[i]; return s; };
    for (int r = 0; r < 3; ++r) { for (int n = 0; n < 32; ++n) b[n] = times(a, a[n]); memcpy(a, b, sizeof a); }      // -> one zero byte
    for (int j = 0; j < 17; ++j) {
        memcpy(T->pow2[j], a, sizeof a);
        for (int n = 0; n < 32; ++n) b[n] = times(a, a[n]);
        memcpy(a, b, sizeof a);
    }
    for (int k = 0; k < CRC_NSH; ++k)                        // shift by CRC_LB << k bytes = 2^(CRC_SH0 + k)
        for (int byte = 0; byte < 4; ++byte)
            for (uint32_t v = 0; v < 256; ++v) T->sh[k][byte][v] = times(T->pow2[CRC_SH0 + k], v << (8 * byte));
}

enum KernelId { K_SPAN_SCAN = 0, K_GRAN_REDUCE, K_GRAN_PREFIX, K_HDR_REC, K_GRAN_LINES, K_GRAN_EXACT, K_FASTA_FINALIZE, K_FETCH,
                K_FASTA_COMP, K_FASTA_COMP_EDGE, K_FASTA_COMP_SMALL, K_FASTQ_LINES, K_FASTQ_ROWS, K_FASTQ_EMIT, K_FASTQ_STATS, K_FASTQ_COMP, K_FASTQ_FETCH, K_BGZF_INFLATE, K_BGZF_COPY, K_BGZF_CRC, K_SCAN_COMP, K_COMP_ATTRIBUTE, K_BGZF_SERIAL, K_FETCH_REST, K_KQ_LINES, K_KQ_PREFIX, K_KQ_WALK, K_KQ_GATHER, K_NKERN };
static const char *const kKernelNames[K_NKERN] = {
    "k_span_scan", "k_gran_reduce", "k_gran_prefix", "k_hdr_rec", "k_gran_lines", "k_gran_exact", "k_fasta_finalize", "k_fetch",
    "k_fasta_comp", "k_fasta_comp_edge", "k_fasta_comp_small", "k_fastq_lines", "k_fastq_rows", "k_fastq_emit", "k_fastq_stats", "k_fastq_comp", "k_fastq_fetch", "k_bgzf_decode", "k_bgzf_copy", "k_bgzf_crc", "k_scan_comp", "k_comp_attribute", "k_bgzf_decode_serial", "k_fetch_rest", "k_kq_lines", "k_kq_prefix", "k_kq_walk", "k_kq_gather"};

struct Prof {
    bool on = false;
    uint32_t mask = ~0u;          // which kernel ids get events (bit i = KernelId i)
    struct Span { int id; hipEvent_t a, b; };
    std::vector<Span> pending;
    std::vector<hipEvent_t> pool;
    double ms[K_NKERN] = {0};
    int64_t cnt[K_NKERN] = {0};
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    bool hit = false;
    void begin(int id, hipStream_t s) {
        hit = on && ((mask >> id) & 1u);
        if (!hit) return;
        Span sp{id, get(), get()};
        (void)hipEventRecord(sp.a, s);
        pending.push_back(sp);
    }
    void end(hipStream_t s) { if (hit) (void)hipEventRecord(pending.back().b, s); }
    void drain() {            // call after the stream has been synchronised
        for (auto &sp : pending) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, sp.a, sp.b) == hipSuccess) { ms[sp.id] += t; cnt[sp.id]++; }
            pool.push_back(sp.a); pool.push_back(sp.b);
        }
        pending.clear();
    }
    ~Prof() { drain(); for (auto e : pool) (void)hipEventDestroy(e); }
};
#define FX_LAUNCH(h, id, kern, grid, block, ...)                                  \
    do {                                                                          \
        (h)->prof.begin(id, (h)->stream);                                         \
        hipLaunchKernelGGL(kern, grid, block, 0, (h)->stream, __VA_ARGS__);       \
        (h)->prof.end((h)->stream);                                               \
    } while (0)

struct StageAsync;                                         // (a file on its way to the device in the background: fx_open_file_async)
static void stage_drop(fx_handle *h);
struct fx_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    StageAsync *stage = nullptr;              // fx_open_file_async: the lanes that are still copying the file into the blob
    int stage_fd = -1;
    hipStream_t stream2 = nullptr;            // a second stream for work beside the handle's own (fx_fxi_dev_build's sort); made on first use, kept:
                                              // hipStreamDestroy behind the munmap of a 10 GB mapping waited 214 ms for the process's mm lock
    // resident stream
    uint8_t *d_data = nullptr;
    uint8_t *d_alloc = nullptr;               // what is freed when owns (d_data points into it for a BGZF byte range)
    size_t blob_cap = 0;                      // != 0: the blob is a block of the scratch pool of that capacity (alloc_blob) and goes back there
    bool owns = false;
    int64_t n = 0;
    bool gz = false;
    // shard context
    int64_t base = 0;
    int prev_byte = '\n';
    bool is_last = true;
    int64_t n_nl = 0;         // newlines of the shard, the virtual end-of-stream one included
    bool scanned = false;     // granule summaries + prefixes are valid (FASTQ count pass)
    // FASTA scan products (fx_spanscan.hpp): per-granule summaries and their prefixes
    int64_t ngran = 0;
    DevBuf<GranPk> gran;
    DevBuf<uint32_t> hdr_grans, irr_grans;    // granules holding a header line / needing the exact walk
    DevBuf<ChunkTot> chunks;
    DevBuf<unsigned long long> ctl;           // Totals (8 words) + list counter + shard summary
    Totals *pin_tot = nullptr;                // pinned host copy of Totals (async read-back without staging)
    struct OneBox { int64_t off, blen, skip, take, dst_off, out_len; } *one_box = nullptr;   // fx_fetch_one: descriptor in pinned host memory
    uint8_t *one_out = nullptr;               // ... and its result buffer (pinned, ONE_CAP bytes): no copies either way
    // the resident kernel that serves single getters (fx_kernels.hpp: k_mailbox)
    Mailbox *mb = nullptr;                    // pinned host memory
    hipStream_t mb_stream = nullptr;
    unsigned long long mb_seq = 0;
    bool mb_running = false, mb_off = false;  // mb_off: the mailbox failed once, single getters keep to the launch path
    // name -> id table (fx_names.hpp)
    DevBuf<uint32_t> nm_table;
    DevBuf<int64_t> nm_off;                   // FASTA: hoff + 1 materialised; FASTQ uses fq_name_off directly
    uint64_t nm_mask = 0;
    int nm_kind = -1;                         // 0 FASTA, 1 FASTQ, -1 none
    DevBuf<int64_t> nl_prefix, hdr_prefix, prevnl;
    // FASTA table
    DevBuf<int64_t> hdr, fa_boff, fa_blen, fa_slen, fa_llen, fa_hdr_line;
    DevBuf<int32_t> fa_elen, fa_norm, fa_dlen, fa_name_len, fa_reg;
    DevBuf<uint32_t> fa_bad;
    int64_t n_hdr = 0, fa_seqlen = 0;
    bool fasta_built = false;
    bool build_pending = false;               // fx_fasta_build_begin enqueued, totals not read back yet
    int pending_full_name = 0;
    DevBuf<uint32_t> comp_runs;               // a build with composition (k_scan_comp): 16 words per run of comp_gpw granules
    int comp_gpw = 0;
    bool comp_runs_valid = false;
    // FASTQ table
    DevBuf<int64_t> fq_name_off, fq_rlen, fq_soff, fq_qoff;
    DevBuf<int32_t> fq_name_len, fq_dlen, fq_qlen;
    DevBuf<FastqAcc> fq_acc;
    DevBuf<FqRun> fq_runs;                    // k_fastq_lines_comp: one record per run of FQLC_G granules (composition counted on the scan)
    DevBuf<FastqAcc> fq_acc_build;            // ... added up by k_fastq_comp_reduce
    DevBuf<uint32_t> fq_rej;                  // [0]: runs whose guess was wrong or missing, [1 ..]: which (counted again from the prefixes)
    int64_t fq_comp_runs = 0, fq_comp_rejected = 0;   // of the last fx_fastq_build_comp: runs of the stream, runs counted again (-1: too many, the table kernel counts)
    bool fq_comp_valid = false;               // the build counted the composition and every run's guess was right
    int64_t fq_comp_base[5] = {0, 0, 0, 0, 0};
    int fq_comp_minqs = 104, fq_comp_maxqs = 33;
    DevBuf<uint32_t> fq_lines;                // k_fastq_lines: FQL_CAP line records per granule
    bool fq_by_lines = false;                 // the last count pass wrote line records (else: counts only)
    bool fq_crlf = false;                     // the sampled windows of the stream hold "\r\n" (fastq_count): k_fastq_lines_comp<true>
    int64_t fq_nlist = 0;                     // granules k_fastq_emit has to read again (overflowing ones + the partial last)
    int64_t n_reads = 0, fq_size = 0, fq_seq_rows = 0;    // complete records; rows that have a sequence line (>= n_reads)
    int64_t fq_c2 = 0;         // newlines of the shard below core_end - 1 (ownership of records, fx_fastq_scan)
    long long fq_maxlen = 0, fq_minlen = 0;
    bool fastq_built = false;
    // Fastx (fx_kseq.hpp): the line table, where each line's bytes go, the records of the kseq walk
    DevBuf<int64_t> kq_nl, kq_ldst;
    DevBuf<uint32_t> kq_lcon;
    DevBuf<KqRec> kq_recs;
    int64_t kq_nrec = -1, kq_lines = 0, kq_seq_bytes = 0, kq_prefix_lines = 0;   // kq_prefix_lines: lines the parallel prefix passes took
    int kq_code = 0;
    // the sorted order of the record names, kept between fx_fxi_dev_sort and fx_fxi_dev_write (fx_fxi_dev.hpp)
    ScratchBuf<int64_t> fxi_order;           // (a block of the scratch pool: 0.8 GB for 10^8 reads; hipFree of it would wait for the device)
    ScratchBuf<int64_t> fxi_soff;            // offset and length of the i-th smallest name (sort_names: what the index kernels read instead of three gathers)
    ScratchBuf<int32_t> fxi_slen;
    int fxi_order_kind = -1;
    int64_t fxi_order_n = 0;
    // this handle's part of a table that several handles write (fx_fxi_part_*): first row of each of its leaves
    ScratchBuf<int64_t> fxi_part_first;
    int64_t fxi_part_nleaf = 0, fxi_part_row_base = 0;
    int fxi_part_kind = -1;
    DevBuf<uint8_t> arena;     // scratch for host-array calls (Staged)
    int64_t arena_used = 0;
    uint8_t *pin_in = nullptr; // pinned staging for the query arrays of host-array calls: pageable source -> here (threads) -> one DMA each
    int64_t pin_in_cap = 0, pin_in_used = 0;
    int64_t halo = 0;          // trailing bytes of the blob that belong to the next shard's core
    // BGZF member table (compressed offset of each member, offset of its data in the inflated stream)
    std::vector<int64_t> gz_moff, gz_coff, gz_uoff;          // member start, start of its deflate data (behind the header), offset of its bytes in the inflated stream
    int64_t gz_csize = 0;
    bool bgzf = false;
    int64_t bgzf_members = 0, bgzf_handed_over = 0;    // members inflated by this open; of them, decoded by the serial kernel
    int bgzf_reason = 0;                               // INFL_RETRY + reason of the first member handed over (fx_inflate_par.hpp)
    int gz_mode = 0;                                   // how a gzip input was inflated: 1 BGZF on the device, 2 one stream serially (zlib), 3 one stream on all host cores (fx_pgzip.hpp), 4 from the restart points of its index
    // restart points of a single gzip stream (captured while it is inflated: GzSerial below)
    std::vector<int64_t> gzp_cin, gzp_cout;
    std::vector<uint8_t> gzp_bits, gzp_has, gzp_win;          // gzp_win: GZ_WINDOW bytes per point that has data, in order
    Prof prof;
};

static bool g_prof_default = false;       // fx_prof_default(): handles are created with timing on (covers staging kernels)

static int use_device(const fx_handle *h) {
    HIPCHK(hipSetDevice(h->device));
    return FX_OK;
}

static int new_handle(int device, fx_handle **out) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(FX_EDEVICE, "no HIP device available (%s); libfxgpu has no CPU fallback", hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(FX_EDEVICE, "device %d out of range (have %d)", device, ndev);
    HIPCHK(hipSetDevice(device));
    fx_handle *h = new fx_handle();
    h->device = device;
    h->prof.on = g_prof_default;
    e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail(FX_EDEVICE, "hipStreamCreate: %s", hipGetErrorString(e)); }
    *out = h;
    return FX_OK;
}

// The blob of an open comes out of the scratch pool too and goes back there when the handle is closed (round 4): the driver
// clears device memory it hands out or takes back -- 3.3 GB: 20 ms on a copy engine, in the way of the next open's staging
// (FX_TRACE_BGZF=1 showed the first 64 MiB of C4 arriving after 6 ms or after 22, depending on what had just been freed) --,
// and a server that opens and closes files keeps asking for the same sizes.  fx_release_scratch gives everything idle back.
static int alloc_blob(fx_handle *h, int64_t n) {
    // pad to a whole tile so vector loads of the last chunk stay inside the allocation
    const int64_t padded = ((n + TILE - 1) / TILE) * TILE + TILE;
    size_t cap = 0;
    h->d_data = (uint8_t *)g_scratch.get(h->device, (size_t)padded, &cap);
    if (!h->d_data) return fail(FX_ENOMEM, "hipMalloc(%lld B) failed", (long long)padded);
    h->blob_cap = cap;
    h->owns = true;
    h->n = n;
    if (padded > n) HIPCHK(hipMemsetAsync(h->d_data + n, 0, (size_t)(padded - n), h->stream));
    return FX_OK;
}
// the blob the handle owns, back to where it came from (the caller has waited for whatever used it)
static void free_blob(fx_handle *h) {
    uint8_t *p = h->d_alloc ? h->d_alloc : h->d_data;
    if (h->owns && p) {
        if (h->blob_cap) g_scratch.put(h->device, p, h->blob_cap);
        else (void)hipFree(p);
    }
    h->d_data = nullptr; h->d_alloc = nullptr; h->blob_cap = 0; h->owns = false; h->n = 0;
}

extern "C" int fx_close(fx_handle *h) {
    if (!h) return FX_OK;
    (void)hipSetDevice(h->device);
    stage_drop(h);                                           // (lanes of fx_open_file_async still copying into the blob)
    if (h->mb) {                                             // send the resident kernel home
        if (h->mb_running) {
            const unsigned long long q = ++h->mb_seq;
            __atomic_store_n(&h->mb->tail, q, __ATOMIC_RELEASE); h->mb->flags_quit = 1ll << 32; __atomic_store_n(&h->mb->head, q, __ATOMIC_RELEASE);
        }
        if (h->mb_stream) { (void)hipStreamSynchronize(h->mb_stream); (void)hipStreamDestroy(h->mb_stream); }
        (void)hipHostFree(h->mb);
    }
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); h->stream2 = nullptr; }
    h->fxi_order.release();                                  // (back to the pool while the stream it names still exists)
    h->fxi_soff.release(); h->fxi_slen.release();
    h->fxi_part_first.release();
    free_blob(h);
    if (h->pin_tot) (void)hipHostFree(h->pin_tot);
    if (h->one_box) (void)hipHostFree(h->one_box);
    if (h->one_out) (void)hipHostFree(h->one_out);
    if (h->pin_in) (void)hipHostFree(h->pin_in);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return FX_OK;
}

extern "C" int64_t fx_size(const fx_handle *h) { return h ? h->n : 0; }
// free / total bytes of a device's HBM as the runtime sees them now (idle blocks of the library's scratch pool count as used:
// fx_release_scratch first, for the figure a new open can really have)
extern "C" int fx_device_memory(int device, int64_t *free_bytes, int64_t *total_bytes) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(FX_EDEVICE, "no HIP device available; libfxgpu has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(FX_EDEVICE, "device %d out of range (have %d)", device, ndev);
    HIPCHK(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return FX_OK;
}
extern "C" int fx_is_gzip(const fx_handle *h) { return h && h->gz; }
extern "C" const void *fx_device_ptr(const fx_handle *h) { return h ? h->d_data : nullptr; }

extern "C" int fx_set_shard(fx_handle *h, int64_t base, int prev_byte, int is_last) {
    if (!h) return fail(FX_EINVAL, "null handle");
    if (h->build_pending) { (void)hipStreamSynchronize(h->stream); h->build_pending = false; }
    h->base = base;
    h->prev_byte = base == 0 ? '\n' : (prev_byte & 0xFF);
    h->is_last = is_last != 0;
    h->scanned = h->fasta_built = h->fastq_built = h->comp_runs_valid = false;
    h->nm_kind = -1;
    return FX_OK;
}

// ----------------------------------------------------------------- staging
static const int64_t STAGE_BYTES = 64ll << 20;   // pinned chunk size
static const int NSTAGE = 3;

struct Stager {          // pinned ring: producer fills slot, H2D async, event marks reuse
    uint8_t *pin[NSTAGE] = {nullptr, nullptr, nullptr};
    hipEvent_t ev[NSTAGE];
    bool ev_ok[NSTAGE] = {false, false, false};
    bool used[NSTAGE] = {false, false, false};
    int init() {
        for (int i = 0; i < NSTAGE; ++i) {
            HIPCHK(hipHostMalloc((void **)&pin[i], (size_t)STAGE_BYTES, hipHostMallocDefault));
            HIPCHK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
            ev_ok[i] = true;
        }
        return FX_OK;
    }
    ~Stager() {
        for (int i = 0; i < NSTAGE; ++i) {
            if (ev_ok[i]) (void)hipEventDestroy(ev[i]);
            if (pin[i]) (void)hipHostFree(pin[i]);
        }
    }
};

// The CPUs next to the device (sysfs local_cpulist of its PCI function): staging threads bound to them copy out of the
// page cache into pinned buffers of the same NUMA node the DMA engine then reads from (FX_STAGE_NUMA=0 turns it off).
static bool device_cpus(int device, cpu_set_t *set) {
    static const bool on = [] { const char *e = getenv("FX_STAGE_NUMA"); return !e || atoi(e) != 0; }();
    if (!on) return false;
    char bus[64] = {0}, path[160];
    if (hipDeviceGetPCIBusId(bus, sizeof bus, device) != hipSuccess) return false;
    for (char *c = bus; *c; ++c) *c = (char)tolower(*c);
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bus);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    char line[4096] = {0};
    const bool got = fgets(line, sizeof line, f) != nullptr;
    fclose(f);
    if (!got) return false;
    CPU_ZERO(set);
    int n = 0;
    for (char *p = line; *p && *p != '\n';) {
        char *e;
        const long a = strtol(p, &e, 10);
        if (e == p) break;
        long b = a;
        if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET((int)c, set); ++n; }
        p = *e == ',' ? e + 1 : e;
        if (p == e && *e != ',') break;
    }
    return n > 0;
}

static int stage_threads() {
    static const int forced = [] { const char *e = getenv("FX_STAGE_THREADS"); return e ? atoi(e) : 0; }();   // experiments
    if (forced > 0) return std::min(forced, 64);
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)std::min<unsigned>(8u, std::max<unsigned>(4u, hw / 8));     // 3 GB from the page cache: 8 threads 47 GB/s, 16: 41, 32: 37 (tools/stage_probe.py)
}
// Lanes that join in when the file turns out to be COLD.  A file that has just been written is read at 2.4 GB/s per thread the first
// time, with or without a device (tools/firstread_probe.c: 19 GB/s with 8 threads, 30 with 16, no more with 24-64; later passes
// 160-270 GB/s): the first Fastq(path) of C3 after the bench wrote its input staged in 1.65-1.8 s with 8 lanes, 1.17-1.35 s with 16.
// A file that has been read before is link-bound with 8, and 16 lanes take memory bandwidth from the thread that sets the index
// file's room aside beside them (it then finished 0.1-0.2 s after the staging).  So the extra lanes wait for a decision (below)
// and leave at once when the file is not cold.
// The decision: once piece STAGE_COLD_PROBE_PIECES is taken (0.8 GB into the file; the head of a file is no measure -- the size
// estimate of an index file has just read it, and one thread alone reads a cold file at 4-10 GB/s), slower than 33 GB/s so far -> cold.
// (A file in the page cache arrives at the link's 50-56 GB/s, a fresh one at 19.)  Files of less than 2 GiB: no extra lanes.
// FX_STAGE_COLD_FORCE=1 (tests): every file of at least two pieces counts as cold from its second piece on.
static const bool STAGE_COLD_FORCE = [] { const char *e = getenv("FX_STAGE_COLD_FORCE"); return e && atoi(e) != 0; }();
static const int64_t STAGE_COLD_PROBE_PIECES = STAGE_COLD_FORCE ? 1 : 96, STAGE_COLD_MIN_PIECES = STAGE_COLD_FORCE ? 2 : 256;
static bool stage_is_cold(double seconds, int64_t bytes) { return STAGE_COLD_FORCE || seconds * 33e9 > (double)bytes; }
static int stage_extra_threads() {
    static const int forced = [] { const char *e = getenv("FX_STAGE_EXTRA_THREADS"); return e ? atoi(e) : -1; }();
    if (forced >= 0) return std::min(forced, 32);
    return stage_threads();
}

// Plain files: T host threads (stage_threads(); more for a cold file: stage_extra_threads()), each with its own pair of
// pinned 8 MiB buffers and its own HIP stream, take the pieces of the file in order, whoever is free: pread into pinned memory,
// hipMemcpyAsync to the blob, double-buffered.  The pinned buffers are allocated once per process
// (pinning 256 MiB costs about as much as moving 1 GB) and reused by later opens.
static const int64_t PIECE_BYTES = [] { const char *e = getenv("FX_STAGE_PIECE_MB"); const int mb = e ? atoi(e) : 0; return (int64_t)(mb > 0 && mb <= 256 ? mb : 8) << 20; }();
struct PinPool {
    std::mutex mu;
    std::vector<uint8_t *> bufs;
    uint8_t *get() {
        {
            std::lock_guard<std::mutex> g(mu);
            if (!bufs.empty()) { uint8_t *p = bufs.back(); bufs.pop_back(); return p; }
        }
        uint8_t *p = nullptr;
        if (hipHostMalloc((void **)&p, (size_t)PIECE_BYTES, hipHostMallocDefault) != hipSuccess) return nullptr;
        return p;
    }
    // at most PIN_KEEP idle buffers stay pinned for the life of the process (256 MiB at the default piece size: what the
    // 8 staging threads + 8 read-back threads use); the rest goes back to the system
    void put(uint8_t *p) {
        static const size_t PIN_KEEP = [] { const char *e = getenv("FX_PIN_KEEP"); const int v = e ? atoi(e) : 0; return (size_t)(v > 0 && v <= 1024 ? v : 32); }();
        {
            std::lock_guard<std::mutex> g(mu);
            if (bufs.size() < PIN_KEEP) { bufs.push_back(p); return; }
        }
        (void)hipHostFree(p);
    }
};
static PinPool g_pins;

// The streams of the staging / copy-out lanes, kept between opens: hipStreamCreate takes ~0.4 ms and the runtime creates them one
// after the other -- the eighth lane of an open had its stream after 3.8 ms and the first 64 MiB of a file were on the device
// after 6 ms instead of 2.5 (FX_TRACE_STAGE=1).  A stream goes back idle (synchronised); at most 32 per device are kept.
struct LaneStreams {
    std::mutex mu;
    std::map<int, std::vector<hipStream_t>> idle;
    hipStream_t get(int device) {
        {
            std::lock_guard<std::mutex> g(mu);
            auto &v = idle[device];
            if (!v.empty()) { hipStream_t s = v.back(); v.pop_back(); return s; }
        }
        hipStream_t s = nullptr;
        return hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess ? s : nullptr;      // (the caller has set the device)
    }
    void put(int device, hipStream_t s) {
        if (hipStreamSynchronize(s) == hipSuccess) {
            std::lock_guard<std::mutex> g(mu);
            auto &v = idle[device];
            if (v.size() < 32) { v.push_back(s); return; }
        }
        (void)hipStreamDestroy(s);
    }
};
static LaneStreams g_lane_streams;

// Pinned host memory for CALLERS (fx_pinned_alloc / fx_pinned_free): answers of a batch land in it by DMA, with no bounce
// buffer and no first-touch page faults (a fresh 100 MB numpy array costs more to fault in than the 1 M answers cost to
// fetch), and query arrays that live in it go up without a staging copy.  Pinning is expensive (hipHostMalloc of 100 MB:
// milliseconds), so released blocks are kept -- up to FX_PINNED_CACHE_MB (default 2048) -- and handed to the next request
// they fit; every live block is in a registry so that the library can tell a pinned pointer from a pageable one.
struct HostPool {
    struct Block { uint8_t *p; size_t cap; };
    std::mutex mu;
    std::vector<Block> idle;
    std::map<const uint8_t *, size_t> live;                   // base -> capacity of the blocks handed out (and of the idle ones: they stay pinned)
    size_t held = 0;
    const size_t limit = [] { const char *e = getenv("FX_PINNED_CACHE_MB"); return (size_t)(e ? std::max(0, atoi(e)) : 2048) << 20; }();
    static size_t round_up(size_t b) {                        // few distinct sizes: 4 KiB steps up to 1 MiB, then 1/8 of the next power of two
        if (b <= (1u << 20)) return (b + 4095) & ~(size_t)4095;
        size_t step = 1;
        while ((step << 4) <= b) step <<= 1;
        return (b + step - 1) & ~(step - 1);
    }
    void *get(size_t bytes) {
        const size_t want = round_up(std::max<size_t>(bytes, 1));
        {
            std::lock_guard<std::mutex> g(mu);
            int best = -1;
            for (int i = 0; i < (int)idle.size(); ++i)
                if (idle[i].cap >= want && idle[i].cap <= 2 * want && (best < 0 || idle[i].cap < idle[best].cap)) best = i;
            if (best >= 0) {
                Block b = idle[best];
                idle.erase(idle.begin() + best);
                held -= b.cap;
                live[b.p] = b.cap;
                return b.p;
            }
        }
        uint8_t *p = nullptr;
        if (hipHostMalloc((void **)&p, want, hipHostMallocDefault) != hipSuccess) {
            trim();
            if (hipHostMalloc((void **)&p, want, hipHostMallocDefault) != hipSuccess) return nullptr;
        }
        std::lock_guard<std::mutex> g(mu);
        live[p] = want;
        return p;
    }
    bool put(void *ptr) {
        uint8_t *p = (uint8_t *)ptr;
        size_t cap = 0;
        {
            std::lock_guard<std::mutex> g(mu);
            auto it = live.find(p);
            if (it == live.end()) return false;
            cap = it->second;
            live.erase(it);
            if (held + cap <= limit) { idle.push_back(Block{p, cap}); held += cap; return true; }
        }
        (void)hipHostFree(p);
        return true;
    }
    // is [p, p + bytes) inside one block that was handed out?
    bool holds(const void *ptr, size_t bytes) {
        const uint8_t *p = (const uint8_t *)ptr;
        std::lock_guard<std::mutex> g(mu);
        auto it = live.upper_bound(p);
        if (it == live.begin()) return false;
        --it;
        return p >= it->first && p + bytes <= it->first + it->second;
    }
    void trim() {
        std::vector<Block> v;
        { std::lock_guard<std::mutex> g(mu); v.swap(idle); held = 0; }
        for (auto &b : v) (void)hipHostFree(b.p);
    }
};
static HostPool g_hostpool;
extern "C" void *fx_pinned_alloc(int64_t bytes) {
    if (bytes < 0) { (void)fail(FX_EINVAL, "fx_pinned_alloc: negative size"); return nullptr; }
    void *p = g_hostpool.get((size_t)bytes);
    if (!p) (void)fail(FX_ENOMEM, "hipHostMalloc(%lld B) failed", (long long)bytes);
    return p;
}
extern "C" void fx_pinned_free(void *p) { if (p) (void)g_hostpool.put(p); }
extern "C" int fx_pinned_holds(const void *p, int64_t bytes) { return p && bytes >= 0 && g_hostpool.holds(p, (size_t)bytes) ? 1 : 0; }
extern "C" int fx_pinned_trim(void) { g_hostpool.trim(); return FX_OK; }

// memcpy with a few threads once it is worth their start (query arrays of a million entries: 8 MB each)
static void par_memcpy(void *dst, const void *src, size_t bytes) {
    const size_t PIECE = 2u << 20;
    if (bytes < 2 * PIECE) { memcpy(dst, src, bytes); return; }
    const int T = (int)std::min<size_t>(6, bytes / PIECE);
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t)
        th.emplace_back([=]() { const size_t a = bytes * t / T, b = bytes * (t + 1) / T; memcpy((uint8_t *)dst + a, (const uint8_t *)src + a, b - a); });
    memcpy(dst, src, bytes / T);
    for (auto &x : th) x.join();
}

static int stage_plain_file(fx_handle *h, int fd, int64_t n, const char *path, uint8_t *d_dst, int64_t file_off = 0) {
    const int64_t npieces = (n + PIECE_BYTES - 1) / PIECE_BYTES;
    const int T0 = (int)std::min<int64_t>(stage_threads(), std::max<int64_t>(1, npieces));
    const int T = npieces >= STAGE_COLD_MIN_PIECES ? T0 + stage_extra_threads() : T0;       // lanes T0 .. T-1: only for a cold file (stage_extra_threads)
    const auto S0 = std::chrono::steady_clock::now();
    std::atomic<int> err(0);                 // 1: read error, 2: device error
    std::atomic<int64_t> next(0);            // the pieces are taken in file order by whoever is free
    std::atomic<int> cold(T > T0 ? -1 : 0);  // -1: not known yet
    std::vector<std::thread> th;
    cpu_set_t near_cpus;
    const bool bind = device_cpus(h->device, &near_cpus);
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t]() {
            if (bind) (void)pthread_setaffinity_np(pthread_self(), sizeof near_cpus, &near_cpus);
            if (t >= T0) {
                while (cold.load() == -1 && !err.load()) usleep(100);
                if (cold.load() != 1) return;
            }
            if (hipSetDevice(h->device) != hipSuccess) { err.store(2); return; }
            uint8_t *pin[2] = {g_pins.get(), g_pins.get()};
            hipStream_t st = nullptr;
            hipEvent_t ev[2] = {nullptr, nullptr};
            bool used[2] = {false, false};
            bool ok = pin[0] && pin[1] && (st = g_lane_streams.get(h->device)) != nullptr &&
                      hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) == hipSuccess;
            if (!ok) err.store(2);
            int slot = 0;
            for (; ok && !err.load(); slot ^= 1) {
                const int64_t k = next.fetch_add(1);
                if (k >= npieces) break;
                const int64_t off = k * PIECE_BYTES, len = std::min(PIECE_BYTES, n - off);
                if (used[slot] && hipEventSynchronize(ev[slot]) != hipSuccess) { err.store(2); break; }
                int64_t done = 0;
                while (done < len) {
                    const ssize_t r = pread(fd, pin[slot] + done, (size_t)(len - done), (off_t)(file_off + off + done));
                    if (r <= 0) { err.store(1); break; }
                    done += r;
                }
                if (done < len) break;
                if (k == STAGE_COLD_PROBE_PIECES && cold.load() == -1) {
                    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - S0).count();
                    cold.store(stage_is_cold(dt, k * PIECE_BYTES) ? 1 : 0);
                    static const bool trace = [] { const char *e = getenv("FX_TRACE_STAGE"); return e && atoi(e) != 0; }();
                    if (trace) fprintf(stderr, "[fxgpu] staging: the first %.0f MiB read in %.1f ms = %.1f GB/s -> %s\n", k * PIECE_BYTES / 1048576.0, dt * 1e3, k * PIECE_BYTES / dt / 1e9, cold.load() ? "a cold file: the extra lanes join in" : "no extra lanes");
                }
                if (hipMemcpyAsync(d_dst + off, pin[slot], (size_t)len, hipMemcpyHostToDevice, st) != hipSuccess ||
                    hipEventRecord(ev[slot], st) != hipSuccess) { err.store(2); break; }
                used[slot] = true;
            }
            { int e = -1; (void)cold.compare_exchange_strong(e, 0); }     // (a lane that is through before anybody decided: nothing is left for the others)
            if (st) (void)hipStreamSynchronize(st);
            for (int i = 0; i < 2; ++i) { if (ev[i]) (void)hipEventDestroy(ev[i]); if (pin[i]) g_pins.put(pin[i]); }
            if (st) g_lane_streams.put(h->device, st);
        });
    for (auto &x : th) x.join();
    if (err.load() == 1) return fail(FX_EIO, "read error on %s", path);
    if (err.load() == 2) return fail(FX_EDEVICE, "staging %s to the device failed", path);
    return FX_OK;
}

// Large results to pageable host memory.  hipMemcpy to a pageable destination runs at ~5 GB/s (the runtime stages it
// through one bounce buffer); here several threads move 8 MiB pieces device -> pinned (own streams) -> destination, the
// copy out of one piece overlapping the transfer of the next: 20+ GB/s.  Used for the name buffer of an index file,
// the sort order and the composition triples (hundreds of MB to GB); small results keep the plain asynchronous copy.
// Everything already enqueued on the handle's stream is waited for first.
static const int64_t D2H_LARGE = 32ll << 20;
static int d2h_large(fx_handle *h, void *dst, const void *d_src, int64_t n) {
    if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(FX_EDEVICE, "stream synchronisation failed");
    const int T = (int)std::min<int64_t>(std::min(stage_threads(), 8), std::max<int64_t>(1, n / PIECE_BYTES));
    std::atomic<int> err(0);
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t]() {
            if (hipSetDevice(h->device) != hipSuccess) { err.store(1); return; }
            uint8_t *pin[2] = {g_pins.get(), g_pins.get()};
            hipStream_t st = nullptr;
            hipEvent_t ev[2] = {nullptr, nullptr};
            bool ok = pin[0] && pin[1] && (st = g_lane_streams.get(h->device)) != nullptr &&
                      hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) == hipSuccess;
            if (!ok) err.store(1);
            int64_t pend_off[2] = {-1, -1}, pend_len[2] = {0, 0};
            int slot = 0;
            auto drain = [&](int sl) {                       // wait for the piece in slot sl and copy it out
                if (pend_off[sl] < 0) return;
                if (hipEventSynchronize(ev[sl]) != hipSuccess) { err.store(1); return; }
                memcpy((uint8_t *)dst + pend_off[sl], pin[sl], (size_t)pend_len[sl]);
                pend_off[sl] = -1;
            };
            for (int64_t off = (int64_t)t * PIECE_BYTES; ok && off < n && !err.load(); off += (int64_t)T * PIECE_BYTES, slot ^= 1) {
                const int64_t len = std::min(PIECE_BYTES, n - off);
                drain(slot);                                  // the slot's previous piece (two iterations ago)
                if (hipMemcpyAsync(pin[slot], (const uint8_t *)d_src + off, (size_t)len, hipMemcpyDeviceToHost, st) != hipSuccess ||
                    hipEventRecord(ev[slot], st) != hipSuccess) { err.store(1); break; }
                pend_off[slot] = off; pend_len[slot] = len;
                drain(slot ^ 1);                              // copy the other slot out while this one travels
            }
            drain(0); drain(1);
            for (int i = 0; i < 2; ++i) { if (ev[i]) (void)hipEventDestroy(ev[i]); if (pin[i]) g_pins.put(pin[i]); }
            if (st) g_lane_streams.put(h->device, st);
        });
    for (auto &x : th) x.join();
    if (err.load()) return fail(FX_EDEVICE, "device to host copy failed");
    return FX_OK;
}
// device -> host on the handle's stream (small) or through d2h_large; the caller still synchronises the stream
static int to_host(fx_handle *h, void *dst, const void *d_src, int64_t bytes) {
    if (bytes <= 0) return FX_OK;
    if (bytes >= D2H_LARGE) return d2h_large(h, dst, d_src, bytes);
    HIPCHK(hipMemcpyAsync(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, h->stream));
    return FX_OK;
}

// ... and straight by DMA when the destination is pinned memory of fx_pinned_alloc (no bounce buffer, no threads)
static int to_host_any(fx_handle *h, void *dst, const void *d_src, int64_t bytes) {
    if (bytes <= 0) return FX_OK;
    if (bytes >= D2H_LARGE && g_hostpool.holds(dst, (size_t)bytes)) {
        HIPCHK(hipMemcpyAsync(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, h->stream));
        return FX_OK;
    }
    return to_host(h, dst, d_src, bytes);
}

static inline unsigned nblocks(int64_t n, int per) { return (unsigned)std::max<int64_t>(1, (n + per - 1) / per); }

// ------------------------------------------------------------------- BGZF
// Member walk (SAM spec 4.1): gzip header with FEXTRA and a 'B','C' subfield whose
// value is BSIZE = total member size - 1; trailer = CRC32, ISIZE.
struct BgzfTable { std::vector<int64_t> moff, coff, uoff; std::vector<int32_t> clen, isize; int64_t total = 0; };

static bool parse_bgzf(const uint8_t *f, int64_t n, BgzfTable &t) {
    int64_t p = 0;
    while (p < n) {
        if (p + 18 > n || f[p] != 0x1f || f[p + 1] != 0x8b || f[p + 2] != 8) return false;
        const int flg = f[p + 3];
        if (!(flg & 4) || (flg & ~4)) return false;          // FEXTRA only (what bgzip writes)
        const int xlen = f[p + 10] | (f[p + 11] << 8);
        if (p + 12 + xlen > n) return false;
        int64_t bsize = -1;
        for (int64_t q = p + 12; q + 4 <= p + 12 + xlen;) {
            const int slen = f[q + 2] | (f[q + 3] << 8);
            if (f[q] == 'B' && f[q + 1] == 'C' && slen == 2 && q + 6 <= p + 12 + xlen) bsize = f[q + 4] | (f[q + 5] << 8);
            q += 4 + slen;
        }
        if (bsize < 0) return false;
        const int64_t msize = bsize + 1, hlen = 12 + xlen;
        if (p + msize > n || msize < hlen + 8) return false;
        const uint8_t *tr = f + p + msize - 8;
        const uint32_t isz = tr[4] | (tr[5] << 8) | (tr[6] << 16) | ((uint32_t)tr[7] << 24);
        if (isz > 65536) return false;
        t.moff.push_back(p); t.coff.push_back(p + hlen); t.clen.push_back((int32_t)(msize - hlen - 8));
        t.uoff.push_back(t.total); t.isize.push_back((int32_t)isz);
        t.total += isz;
        p += msize;
    }
    return !t.moff.empty();
}

template <class T> static int upload(fx_handle *h, DevBuf<T> &d, const std::vector<T> &v) {
    int rc = d.alloc((int64_t)v.size());
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(d.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
    return FX_OK;
}

// The kernels of a BGZF open over members whose table is ON THE DEVICE already (d_coff: first deflate byte of each member in
// d_cp, d_clen: deflate bytes, d_uoff: offset of its bytes in the inflated stream, d_isize), the compressed bytes staged:
// decode (one wave per member) -> what it handed over, serially -> copy (matches) -> crc -> the blob.  moff_of(m): file offset
// of member m, for an error message.
template <class MoffOf>
static int bgzf_inflate_staged(fx_handle *h, const uint8_t *d_cp, const int64_t *d_coff, const int32_t *d_clen, const int64_t *d_uoff,
                               const int32_t *d_isize, int64_t nmem, int64_t total, int32_t clen_max, const char *path, MoffOf moff_of,
                               const std::function<void(const char *)> &lap, int64_t m_first = -1) {
    // m_first >= 0: members [m_first, m_first + nmem) of a file that is inflated group by group (bgzf_open_pipelined): the blob
    // exists already (d_uoff are offsets into it) and the counts of the handle are added to
    int rc;
    static const bool trace = [] { const char *e = getenv("FX_TRACE"), *b = getenv("FX_TRACE_BGZF"); return (e && atoi(e) != 0) || (b && atoi(b) != 0); }();
    ScratchBuf<int32_t> d_status, d_pstatus;
    // where the matches of a member begin: one bit per output byte (k_bgzf_decode sets them, k_bgzf_copy walks them)
    ScratchBuf<uint64_t> d_map;
    if ((rc = d_map.alloc(h->device, nmem * BM_WORDS, h->stream))) return rc;
    if ((rc = d_status.alloc(h->device, nmem, h->stream)) || (rc = d_pstatus.alloc(h->device, nmem + 1, h->stream))) return rc;      // (+ 1: the counter the waves take their members from)
    ScratchBuf<uint16_t> d_gsym;                             // canonical symbol order of every member's tables (slow path of the decoder)
    if ((rc = d_gsym.alloc(h->device, nmem * GSYM, h->stream))) return rc;
    HIPCHK(hipMemsetAsync(d_status.p, 0xFF, (size_t)nmem * 4, h->stream));
    HIPCHK(hipMemsetAsync(d_pstatus.p, 0, (size_t)(nmem + 1) * 4, h->stream));
    HIPCHK(hipMemsetAsync(d_map.p, 0, (size_t)nmem * BM_WORDS * 8, h->stream));
    if (m_first < 0 && (rc = alloc_blob(h, total))) return rc;
    lap("allocations");
    // one wave per member, the lanes at 64 bit positions of it (fx_inflate_par.hpp); members it hands over (status INFL_RETRY:
    // anything out of the ordinary, damaged members included) are decoded by one lane each, as in round 2.  FX_BGZF_SERIAL=1: only that.
    static const bool serial_only = [] { const char *e = getenv("FX_BGZF_SERIAL"); return e && atoi(e) != 0; }();
    static const int dbg_par = [] { const char *e = getenv("FX_BGZF_DBG"); return e ? atoi(e) : 0; }();
    // LDS of a wave: the tables + the member's payload (sized for the largest member of the file, at most 64 KiB of the 160 per CU)
    const int lds_payload = (int)std::min<int64_t>(((int64_t)clen_max + 16 + 255) & ~255ll, 65536);
    static const bool stage = [] { const char *e = getenv("FX_BGZF_STAGE"); return e && atoi(e) != 0; }();   // the payload through LDS (experiment)
    static const bool fixed_shares = [] { const char *e = getenv("FX_BGZF_FIXED_SHARES"); return e && atoi(e) != 0; }();   // members m, m + grid, ... per wave (the form before; comparison)
    static const bool replay = [] { const char *e = getenv("FX_BGZF_REPLAY"); return e && atoi(e) != 0; }();   // phase B replays the symbols phase A left behind (experiment: no faster -- what B costs is its stores)
    const size_t par_lds = ((sizeof(PTab) + 15) & ~(size_t)15) + (stage ? (size_t)lds_payload : 0);
    const auto par_kernel = stage ? (replay ? k_bgzf_decode_par<true, true> : k_bgzf_decode_par<true, false>)
                                  : (replay ? k_bgzf_decode_par<false, true> : k_bgzf_decode_par<false, false>);
    static bool par_attr = false;
    if (!par_attr) {
        (void)hipFuncSetAttribute((const void *)k_bgzf_decode_par<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        (void)hipFuncSetAttribute((const void *)k_bgzf_decode_par<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        par_attr = true;
    }
    // a grid of as many waves as the device holds at once; every wave owns SYM_ROWS rows of 64 symbols of scratch
    constexpr int SYM_ROWS = 2048;
    int per_cu = 0, n_cu = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, par_kernel, 64, par_lds) != hipSuccess || per_cu <= 0) per_cu = 8;
    // (the LDS of a workgroup is handed out in steps: 14 880 bytes were reported as 11 workgroups per CU and 10 were resident -- the
    // persistent grid's eleventh wave per CU then ran alone after the others, doubling the kernel's time; round 6)
    per_cu = std::min<int>(per_cu, (int)((160 * 1024) / ((par_lds + 1023) & ~(size_t)1023)));
    { const char *e = getenv("FX_BGZF_WAVES_PER_CU"); if (e && atoi(e) > 0) per_cu = std::min(per_cu, atoi(e)); }      // experiments: fewer members in flight than fit
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device);
    const unsigned par_grid = (unsigned)std::min<int64_t>(nmem, (int64_t)per_cu * n_cu);
    ScratchBuf<uint32_t> d_sym;
    if (replay && !serial_only && (rc = d_sym.alloc(h->device, (int64_t)par_grid * SYM_ROWS * 64, h->stream))) return rc;
    if (!serial_only) {
        h->prof.begin(K_BGZF_INFLATE, h->stream);
        hipLaunchKernelGGL(par_kernel, dim3(par_grid), dim3(64), par_lds, h->stream, d_cp, d_coff, d_clen, d_uoff, d_isize, nmem,
                           h->d_data, d_status.p, d_map.p, dbg_par, lds_payload, d_pstatus.p, d_sym.p, SYM_ROWS, fixed_shares ? nullptr : d_pstatus.p + nmem);
        h->prof.end(h->stream);
        if (trace) {                                         // how many members the wave-per-member kernel handed over, and why
            std::vector<int32_t> stv((size_t)nmem);
            HIPCHK(hipMemcpyAsync(stv.data(), d_status.p, (size_t)nmem * 4, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            int64_t hist[64] = {0};
            for (int32_t v : stv) if (v >= INFL_RETRY) hist[std::min(v - INFL_RETRY, 63)]++;
            for (int i = 0; i < 64; ++i) if (hist[i]) fprintf(stderr, "[fxgpu] bgzf handed over: reason %d x %lld\n", i, (long long)hist[i]);
        }
    }
    if (dbg_par && !serial_only) {                          // timing probe of the kernel's phases: wrong answers, so nothing after it runs
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipStreamSynchronize(h->stream);
        (void)hipMemsetAsync(d_pstatus.p + nmem, 0, 4, h->stream);       // (the members' counter, used up by the launch above)
        (void)hipEventRecord(e0, h->stream);
        hipLaunchKernelGGL(par_kernel, dim3(par_grid), dim3(64), par_lds, h->stream, d_cp, d_coff, d_clen, d_uoff, d_isize, nmem,
                           h->d_data, d_status.p, d_map.p, dbg_par, lds_payload, d_pstatus.p, d_sym.p, SYM_ROWS, fixed_shares ? nullptr : d_pstatus.p + nmem);
        (void)hipEventRecord(e1, h->stream);
        (void)hipStreamSynchronize(h->stream);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        fprintf(stderr, "[fxgpu] k_bgzf_decode_par dbg=%d: %.3f ms for %lld members\n", dbg_par, ms, (long long)nmem);
        return fail(FX_EIO, "FX_BGZF_DBG is a timing probe");
    }
    FX_LAUNCH(h, K_BGZF_SERIAL, k_bgzf_decode, dim3(nblocks(nmem, INFL_BLOCK)), dim3(INFL_BLOCK), d_cp, d_coff,
              d_clen, d_uoff, d_isize, nmem, h->d_data, d_status.p, d_map.p, d_gsym.p, serial_only ? -1 : (int)INFL_RETRY);
    FX_LAUNCH(h, K_BGZF_COPY, k_bgzf_copy, dim3(nblocks(nmem, COPY_BLOCK / 64)), dim3(COPY_BLOCK), d_uoff, d_isize, nmem, h->d_data, d_map.p);
    static const bool no_crc = [] { const char *e = getenv("FX_BGZF_NO_CRC"); return e && atoi(e) != 0; }();
    ScratchBuf<CrcTables> d_crc;
    if (!no_crc) {                                           // every member against the CRC-32 of its trailer, as zlib does in gzread
        static const CrcTables *tabs = [] { CrcTables *t = new CrcTables; crc_tables(t); return t; }();
        if ((rc = d_crc.alloc(h->device, 1, h->stream))) return rc;
        HIPCHK(hipMemcpyAsync(d_crc.p, tabs, sizeof(CrcTables), hipMemcpyHostToDevice, h->stream));
        FX_LAUNCH(h, K_BGZF_CRC, k_bgzf_crc, dim3(nblocks(nmem, 4)), dim3(256), h->d_data, d_uoff, d_isize, d_cp, d_coff, d_clen,
                  nmem, d_crc.p, d_status.p);
    }
    HIPCHK(hipGetLastError());
    std::vector<int32_t> status((size_t)nmem), pstatus((size_t)nmem);
    HIPCHK(hipMemcpyAsync(status.data(), d_status.p, (size_t)nmem * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(pstatus.data(), d_pstatus.p, (size_t)nmem * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (m_first <= 0) { h->bgzf_members = 0; h->bgzf_handed_over = 0; h->bgzf_reason = 0; }
    h->bgzf_members += nmem;
    for (int64_t m = 0; m < nmem; ++m)
        if (serial_only || pstatus[(size_t)m] >= INFL_RETRY) { if (!h->bgzf_handed_over++) h->bgzf_reason = pstatus[(size_t)m]; }
    lap("kernels done");
    for (int64_t m = 0; m < nmem; ++m)
        if (status[m] != INFL_OK)
            return fail(FX_EIO, status[m] == INFL_ECRC ? "BGZF member %lld of %s (offset %lld): CRC-32 of the inflated bytes differs from the trailer (code %d)"
                                                       : "BGZF member %lld of %s (offset %lld) failed to inflate: code %d",
                        (long long)(m + (m_first > 0 ? m_first : 0)), path, (long long)moff_of(m), status[m]);
    return FX_OK;
}

// compressed bytes (file -> pinned pieces -> HBM, stage_plain_file) -> k_bgzf_inflate -> resident blob.
// [m0, m1): the members to inflate (all of them for a whole file; the ones that cover a byte range of the inflated
// stream for fx_open_file_range -- only their compressed bytes are read and staged).
static int bgzf_to_blob(fx_handle *h, int fd, int64_t fsize_all, const BgzfTable &full, const char *path, int64_t m0 = 0, int64_t m1 = -1) {
    ScratchBuf<uint8_t> d_c;
    int rc;
    static const bool trace = [] { const char *e = getenv("FX_TRACE"), *b = getenv("FX_TRACE_BGZF"); return (e && atoi(e) != 0) || (b && atoi(b) != 0); }();
    const auto T0 = std::chrono::steady_clock::now();
    const std::function<void(const char *)> lap = [&](const char *what) {
        if (trace) fprintf(stderr, "[fxgpu] bgzf %-22s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count());
    };
    if (m1 < 0) m1 = (int64_t)full.moff.size();
    BgzfTable t;                                           // the range, offsets relative to its first member
    const int64_t c0 = full.moff[(size_t)m0], u0 = full.uoff[(size_t)m0];
    const int64_t c1 = m1 < (int64_t)full.moff.size() ? full.moff[(size_t)m1] : fsize_all;
    for (int64_t m = m0; m < m1; ++m) {
        t.moff.push_back(full.moff[(size_t)m]); t.coff.push_back(full.coff[(size_t)m] - c0); t.uoff.push_back(full.uoff[(size_t)m] - u0);
        t.clen.push_back(full.clen[(size_t)m]); t.isize.push_back(full.isize[(size_t)m]);
        t.total += full.isize[(size_t)m];
    }
    const int64_t fsize = c1 - c0;
    DevBuf<int64_t> d_coff, d_uoff;
    DevBuf<int32_t> d_clen, d_isize;
    if ((rc = d_c.alloc(h->device, fsize + 48, h->stream))) return rc;          // the bit reader looks three 8-byte words ahead
    HIPCHK(hipMemsetAsync(d_c.p + fsize, 0, 48, h->stream));
    lap("alloc compressed");
    if ((rc = stage_plain_file(h, fd, fsize, path, d_c.p, c0))) return rc;
    lap("staged");
    if ((rc = upload(h, d_coff, t.coff)) || (rc = upload(h, d_uoff, t.uoff)) || (rc = upload(h, d_clen, t.clen)) ||
        (rc = upload(h, d_isize, t.isize)))
        return rc;
    const int64_t nmem = (int64_t)t.moff.size();
    int32_t clen_max = 0;
    for (int32_t c : t.clen) clen_max = std::max(clen_max, c);
    if ((rc = bgzf_inflate_staged(h, d_c.p, d_coff.p, d_clen.p, d_uoff.p, d_isize.p, nmem, t.total, clen_max, path,
                                  [&](int64_t m) { return t.moff[(size_t)m]; }, lap)))
        return rc;
    h->bgzf = true;
    h->gz_mode = 1;
    h->gz_moff = full.moff; h->gz_coff = full.coff; h->gz_uoff = full.uoff; h->gz_csize = fsize_all;
    return FX_OK;
}

// stage_plain_file with the caller free to go on: the same lanes (threads, own streams, two pinned pieces each), every piece
// with an event of its own that another stream can wait for -- wait_until() makes `s` wait for the pieces that cover the
// first `upto` bytes (the pieces are handed out in file order, so a prefix of the file is complete long before its end).
struct StageAsync {
    fx_handle *h = nullptr;
    int fd = -1;
    int64_t n = 0, piece = 0, npieces = 0;
    uint8_t *d_dst = nullptr;
    int T = 0;
    std::atomic<int> err{0};                                  // 1: read error, 2: device error
    std::atomic<int64_t> next{0};                             // the next piece nobody has taken
    std::atomic<int> cold{0};                                 // -1: not known yet; 1: the extra lanes join in (stage_extra_threads)
    std::vector<std::thread> th;
    std::vector<hipEvent_t> ev;                               // one per piece
    std::unique_ptr<std::atomic<int>[]> issued;               // 1: the piece's copy and its event are in the lane's stream
    int64_t waited = 0;                                       // pieces `wait_until` has been through
    cpu_set_t near_cpus;                                      // the cores next to the device (the lanes are bound to them)
    std::chrono::steady_clock::time_point T_start;
    int start(fx_handle *hh, int fd_, int64_t n_, uint8_t *dst, int64_t piece_bytes) {
        T_start = std::chrono::steady_clock::now();
        h = hh; fd = fd_; n = n_; d_dst = dst; piece = std::min<int64_t>(PIECE_BYTES, piece_bytes);
        npieces = (n + piece - 1) / piece;
        const int T0 = (int)std::min<int64_t>(stage_threads(), std::max<int64_t>(1, npieces));
        T = npieces >= STAGE_COLD_MIN_PIECES && piece == PIECE_BYTES ? T0 + stage_extra_threads() : T0;    // lanes T0 .. T-1: only for a cold file (stage_extra_threads)
        cold.store(T > T0 ? -1 : 0);
        ev.assign((size_t)npieces, nullptr);
        issued.reset(new std::atomic<int>[(size_t)npieces]);
        for (int64_t i = 0; i < npieces; ++i) {
            issued[(size_t)i].store(0);
            if (hipEventCreateWithFlags(&ev[(size_t)i], hipEventDisableTiming) != hipSuccess) { err.store(2); return fail(FX_EDEVICE, "hipEventCreate failed"); }
        }
        const bool bind = device_cpus(h->device, &near_cpus);
        for (int t = 0; t < T; ++t)
            th.emplace_back([this, t, bind, T0]() {
                if (bind) (void)pthread_setaffinity_np(pthread_self(), sizeof near_cpus, &near_cpus);
                if (t >= T0) {
                    while (cold.load() == -1 && !err.load()) usleep(100);
                    if (cold.load() != 1) return;
                }
                if (hipSetDevice(h->device) != hipSuccess) { err.store(2); return; }
                uint8_t *pin[2] = {g_pins.get(), g_pins.get()};
                hipStream_t st = nullptr;
                int64_t last[2] = {-1, -1};                  // the piece whose copy reads the slot
                bool ok = pin[0] && pin[1] && (st = g_lane_streams.get(h->device)) != nullptr;
                if (!ok) err.store(2);
                int slot = 0;
                static const bool trace = [] { const char *e = getenv("FX_TRACE_STAGE"); return e && atoi(e) != 0; }();
                const auto L0 = std::chrono::steady_clock::now();
                auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - L0).count(); };
                if (trace) fprintf(stderr, "[fxgpu] lane %d: stream + pins after %.2f ms (since the lane began: %.2f)\n", t, std::chrono::duration<double, std::milli>(L0 - T_start).count(), 0.0);
                bool first = true;
                for (; ok && !err.load(); slot ^= 1) {
                    const int64_t k = next.fetch_add(1);         // in file order, by whoever is free
                    if (k >= npieces) break;
                    const int64_t off = k * piece, len = std::min(piece, n - off);
                    if (last[slot] >= 0 && hipEventSynchronize(ev[(size_t)last[slot]]) != hipSuccess) { err.store(2); break; }
                    int64_t done = 0;
                    while (done < len) {
                        const ssize_t r = pread(fd, pin[slot] + done, (size_t)(len - done), (off_t)(off + done));
                        if (r <= 0) { err.store(1); break; }
                        done += r;
                    }
                    if (done < len) break;
                    if (k == STAGE_COLD_PROBE_PIECES && cold.load() == -1)
                        cold.store(stage_is_cold(std::chrono::duration<double>(std::chrono::steady_clock::now() - T_start).count(), k * piece) ? 1 : 0);
                    if (trace && first) fprintf(stderr, "[fxgpu] lane %d: first piece read after %.2f ms\n", t, since());
                    if (hipMemcpyAsync(d_dst + off, pin[slot], (size_t)len, hipMemcpyHostToDevice, st) != hipSuccess ||
                        hipEventRecord(ev[(size_t)k], st) != hipSuccess) { err.store(2); break; }
                    if (trace && first) fprintf(stderr, "[fxgpu] lane %d: first copy queued after %.2f ms\n", t, since());
                    first = false;
                    last[slot] = k;
                    issued[(size_t)k].store(1, std::memory_order_release);
                }
                { int e = -1; (void)cold.compare_exchange_strong(e, 0); }
                if (st) (void)hipStreamSynchronize(st);
                for (int i = 0; i < 2; ++i) if (pin[i]) g_pins.put(pin[i]);
                if (st) g_lane_streams.put(h->device, st);
            });
        return FX_OK;
    }
    // the HOST waits until the first `upto` bytes of the file are on the device; -> 0, or the error of a lane.  (Not
    // hipStreamWaitEvent: the lanes' streams and the stream of the kernels share a handful of hardware queues, and a kernel
    // queued behind the markers of copies that are still to come starts when THOSE have landed -- measured: the first group's
    // kernels began when the whole file was there.  The kernels run on a stream of another priority, which has queues of its own.)
    int wait_until(int64_t upto) {
        const int64_t pe = std::min(npieces, (upto + piece - 1) / piece);
        for (; waited < pe; ++waited) {
            while (!issued[(size_t)waited].load(std::memory_order_acquire)) {
                if (err.load()) return err.load();
                usleep(20);
            }
            if (hipEventSynchronize(ev[(size_t)waited]) != hipSuccess) return 2;
        }
        return err.load();
    }
    int finish() {
        for (auto &x : th) x.join();
        th.clear();
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
        ev.clear();
        return err.load();
    }
    ~StageAsync() { (void)finish(); }
};

static void stage_drop(fx_handle *h) {
    if (h->stage) { (void)h->stage->finish(); delete h->stage; h->stage = nullptr; }
    if (h->stage_fd >= 0) { close(h->stage_fd); h->stage_fd = -1; }
}

// A PLAIN file on its way to the device while the caller already works on what has landed (round 6): the blob is allocated,
// the staging lanes start, the handle comes back at once.  The pieces land in file order; fx_stage_wait(h, upto) returns when
// the first `upto` bytes are there (upto < 0: all of them, the lanes joined) -- only then may anything read them: a caller
// builds on PREFIXES through views (fx_open_device on fx_device_ptr(h) + offset, fx_set_shard, fx_set_halo), as the sharded
// build does on byte ranges, and calls nothing else on THIS handle before fx_stage_wait(h, -1).  What it buys: the table
// leaves of an index file leave for the host (fx_fxi_part_leaves of the views) while later parts of the input still arrive
// -- the link is full duplex, the reference's loop (fastq.c:8-182) knows neither direction.  gzip input: FX_EINVAL (fx_open_file).
extern "C" int fx_open_file_async(const char *path, int device, fx_handle **out) {
    if (!path || !out) return fail(FX_EINVAL, "null argument");
    struct stat st;
    if (stat(path, &st) != 0 || !S_ISREG(st.st_mode)) return fail(FX_ENOENT, "the input file %s does not exists", path);
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(FX_ENOENT, "cannot open %s", path);
    unsigned char magic[4] = {0, 0, 0, 0};
    if (pread(fd, magic, 4, 0) == 4 && magic[0] == 0x1f && magic[1] == 0x8b) { close(fd); return fail(FX_EINVAL, "%s is gzip-compressed: fx_open_file", path); }
    fx_handle *h = nullptr;
    int rc = new_handle(device, &h);
    if (rc) { close(fd); return rc; }
    const int64_t n = (int64_t)st.st_size;
    const auto t0 = std::chrono::steady_clock::now();
    if ((rc = alloc_blob(h, n)) || hipStreamSynchronize(h->stream) != hipSuccess) { close(fd); fx_close(h); return rc ? rc : fail(FX_EDEVICE, "stream sync failed"); }
    g_open_laps[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    g_open_laps[1] = 0;
    h->stage_fd = fd;
    if (n > 0) {
        h->stage = new StageAsync();
        if ((rc = h->stage->start(h, fd, n, h->d_data, PIECE_BYTES))) { fx_close(h); return rc; }
    }
    *out = h;
    return FX_OK;
}
extern "C" int fx_stage_wait(fx_handle *h, int64_t upto) {
    if (!h) return fail(FX_EINVAL, "null handle");
    if (!h->stage) return FX_OK;                             // nothing in flight
    const int e = upto < 0 ? h->stage->finish() : h->stage->wait_until(std::min<int64_t>(upto, h->n));
    if (upto < 0) stage_drop(h);
    if (e == 1) return fail(FX_EIO, "read error on the input file");
    if (e) return fail(FX_EDEVICE, "staging the input file to the device failed");
    return FX_OK;
}

// A large BGZF file, the inflate running BEHIND the staging instead of after it (round 4).  The compressed bytes reach the
// device at what the copy out of the page cache gives (~46 GB/s: 21 ms for C4's 0.97 GB) and the kernels of the whole file take
// about as long again (18 ms); neither needs the other's engine.  The file is taken in groups of FX_BGZF_GROUP bytes (256 MiB; the first ones smaller):
// as soon as a group has landed, its members are found (the signature search of fx_bgzf_walk.hpp over the group's granules --
// all but the last one, whose hits may need bytes of the next group; the last member found waits for the next group too, which
// knows where it ends), their ISIZE scanned on from the running total, and they are inflated -- decode, copy, CRC -- while the
// next groups arrive.  The size of the inflated stream is not known before the last group: the blob is allocated after the
// first group for that group's ratio + 10 % (a file whose later groups inflate further than that takes the one-shot path:
// -> 1, as for any file this path does not take; what is left over at the end stays with the blob).
static int bgzf_open_pipelined(fx_handle *h, int fd, int64_t fsize, const char *path, int64_t group) {
    static const bool trace = [] { const char *e = getenv("FX_TRACE"), *b = getenv("FX_TRACE_BGZF"); return (e && atoi(e) != 0) || (b && atoi(b) != 0); }();
    const auto T0 = std::chrono::steady_clock::now();
    const std::function<void(const char *)> lap = [&](const char *what) {
        if (trace) fprintf(stderr, "[fxgpu] bgzf/p %-20s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count());
    };
    const std::function<void(const char *)> quiet = [](const char *) {};
    int rc;
    // the kernels on a stream of the highest priority for the length of this open: hardware queues of its own (see wait_until)
    struct PrioStream {
        fx_handle *h; hipStream_t saved, mine = nullptr;
        explicit PrioStream(fx_handle *hh) : h(hh), saved(hh->stream) {
            int lo = 0, hi = 0;
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo &&
                hipStreamCreateWithPriority(&mine, hipStreamNonBlocking, hi) == hipSuccess) { (void)hipStreamSynchronize(saved); h->stream = mine; }
            else mine = nullptr;
        }
        ~PrioStream() { if (mine) { (void)hipStreamSynchronize(mine); h->stream = saved; (void)hipStreamDestroy(mine); } }
    } prio(h);
    ScratchBuf<uint8_t> d_c;
    if ((rc = d_c.alloc(h->device, fsize + 48, h->stream))) return rc;
    HIPCHK(hipMemsetAsync(d_c.p + fsize, 0, 48, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (trace) lap("  staging area, stream");
    StageAsync stg;
    if ((rc = stg.start(h, fd, fsize, d_c.p, group))) return rc;
    if (trace) lap("  lanes started");
    // whatever happens below, the lanes are joined before the staging buffer goes back to the pool (d_c is declared first)
    const int64_t ngran_all = (fsize + 4095) / 4096;
    // where the groups end: the first ones are smaller (a quarter, a quarter, a half of `group`), so that the kernels begin
    // when the first few pieces are there and not after the first 128 MiB
    std::vector<int64_t> edge;
    for (int64_t at = 0, k = 0; at < fsize; ++k) {
        const int64_t step = std::max<int64_t>(4096, (k < 2 ? group / 4 : k == 2 ? group / 2 : group) & ~4095ll);
        at = std::min(fsize, at + step);
        if (fsize - at < step / 2) at = fsize;                 // no sliver at the end
        edge.push_back(at);
    }
    const int64_t ngroups = (int64_t)edge.size();
    int64_t carry = -1;                                        // start of the member the group before left over
    int64_t U = 0, M = 0, cap = 0;                             // inflated bytes / members so far; bytes the blob can take
    h->gz_moff.clear(); h->gz_coff.clear(); h->gz_uoff.clear();
    auto give_up = [&](int code) {                             // 1: the one-shot path decides; < 0: an error
        (void)stg.finish();
        if (h->owns && h->d_data) { (void)hipStreamSynchronize(h->stream); free_blob(h); }
        h->gz_moff.clear(); h->gz_coff.clear(); h->gz_uoff.clear();
        return code;
    };
    for (int64_t g = 0; g < ngroups; ++g) {
        const bool last = g + 1 == ngroups;
        const int64_t upto = edge[(size_t)g];
        const int e = stg.wait_until(upto);
        if (e) { (void)give_up(0); return e == 1 ? fail(FX_EIO, "read error on %s", path) : fail(FX_EDEVICE, "staging %s to the device failed", path); }
        if (trace) lap("  bytes of the group there");
        const int64_t ga = g == 0 ? 0 : edge[(size_t)g - 1] / 4096 - 1, gb = last ? ngran_all : upto / 4096 - 1, ng = gb - ga;
        if (ng <= 0) continue;
        const int64_t nchunks = (ng + SCAN_CHUNK - 1) / SCAN_CHUNK;
        ScratchBuf<int64_t> d_i64;
        ScratchBuf<int32_t> d_cnt;
        if ((rc = d_i64.alloc(h->device, ng + 1 + nchunks + 1, h->stream)) || (rc = d_cnt.alloc(h->device, ng, h->stream))) return give_up(rc);
        int64_t *d_off = d_i64.p, *d_sums = d_i64.p + ng + 1;
        hipLaunchKernelGGL(k_bgzf_sig_count, dim3(nblocks(ng, BLOCK / 64)), dim3(BLOCK), 0, h->stream, (const uint8_t *)d_c.p, fsize, ga, ng, d_cnt.p);
        hipLaunchKernelGGL(k_cnt_chunk_sums, dim3((unsigned)nchunks), dim3(BLOCK), 0, h->stream, (const int32_t *)d_cnt.p, ng, d_sums);
        hipLaunchKernelGGL(k_cnt_chunk_bases, dim3(1), dim3(BLOCK), 0, h->stream, d_sums, nchunks);
        hipLaunchKernelGGL(k_cnt_offsets, dim3((unsigned)nchunks), dim3(BLOCK), 0, h->stream, (const int32_t *)d_cnt.p, ng, (const int64_t *)d_sums, d_off);
        int64_t found = 0;
        if (hipMemcpyAsync(&found, d_off + ng, 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
            return give_up(fail(FX_EDEVICE, "BGZF member search failed on the device"));
        if (found < 0 || found > (gb - ga) * 4096 / (BGZF_HDR + 8)) return give_up(1);
        if (g == 0 && found == 0) return give_up(1);
        const int64_t nstarts = found + (carry >= 0 ? 1 : 0);
        const int64_t nmem = last ? nstarts : nstarts - 1;     // the last start of a group waits for the next one
        if (nstarts == 0) continue;
        const int64_t mchunks = (std::max<int64_t>(nmem, 1) + SCAN_CHUNK - 1) / SCAN_CHUNK;
        ScratchBuf<int64_t> d_t64;
        ScratchBuf<int32_t> d_t32;
        if ((rc = d_t64.alloc(h->device, nstarts + 2 * nmem + 2 + mchunks + 1, h->stream)) || (rc = d_t32.alloc(h->device, 2 * nmem + 4, h->stream))) return give_up(rc);
        int64_t *d_mstart = d_t64.p, *d_coff = d_t64.p + nstarts, *d_uoff = d_coff + nmem, *d_msums = d_uoff + nmem + 1;
        int32_t *d_clen = d_t32.p, *d_isize = d_t32.p + nmem, *d_flags = d_t32.p + 2 * nmem;
        HIPCHK(hipMemsetAsync(d_flags, 0, 8, h->stream));
        if (carry >= 0) HIPCHK(hipMemcpyAsync(d_mstart, &carry, 8, hipMemcpyHostToDevice, h->stream));
        if (found) hipLaunchKernelGGL(k_bgzf_sig_emit, dim3(nblocks(ng, BLOCK / 64)), dim3(BLOCK), 0, h->stream, (const uint8_t *)d_c.p, fsize, ga, ng,
                                      (const int64_t *)d_off, d_mstart + (carry >= 0 ? 1 : 0));
        const int64_t first_start = carry >= 0 ? carry : 0;
        int64_t next_carry = -1;
        if (!last) HIPCHK(hipMemcpyAsync(&next_carry, d_mstart + nstarts - 1, 8, hipMemcpyDeviceToHost, h->stream));
        if (nmem == 0) {                                       // one start only: it waits
            HIPCHK(hipStreamSynchronize(h->stream));
            if (g == 0 && next_carry != 0) return give_up(1);
            carry = next_carry;
            continue;
        }
        hipLaunchKernelGGL(k_bgzf_member_rows, dim3(nblocks(nmem, BLOCK)), dim3(BLOCK), 0, h->stream, (const uint8_t *)d_c.p, fsize, (const int64_t *)d_mstart, nmem,
                           nstarts, first_start, d_coff, d_clen, d_isize, (int *)d_flags, d_flags + 1);
        hipLaunchKernelGGL(k_cnt_chunk_sums, dim3((unsigned)mchunks), dim3(BLOCK), 0, h->stream, (const int32_t *)d_isize, nmem, d_msums);
        hipLaunchKernelGGL(k_cnt_chunk_bases, dim3(1), dim3(BLOCK), 0, h->stream, d_msums, mchunks);
        hipLaunchKernelGGL(k_cnt_offsets, dim3((unsigned)mchunks), dim3(BLOCK), 0, h->stream, (const int32_t *)d_isize, nmem, (const int64_t *)d_msums, d_uoff);
        if (U) hipLaunchKernelGGL(k_add_base, dim3(nblocks(nmem + 1, BLOCK)), dim3(BLOCK), 0, h->stream, d_uoff, nmem + 1, U);
        int32_t flags[2] = {0, 0};
        int64_t u_end = 0;
        HIPCHK(hipMemcpyAsync(flags, d_flags, 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(&u_end, d_uoff + nmem, 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (flags[0] || u_end < U) return give_up(1);          // the chain does not tile the file
        if (!last && (next_carry <= first_start)) return give_up(1);
        if (!cap) {                                            // the blob, from the ratio of the first members
            const int64_t cbytes = (last ? fsize : next_carry) - first_start;
            if (u_end <= 0 || cbytes <= 0) return give_up(1);
            const double ratio = (double)u_end / (double)cbytes;
            cap = last ? u_end : (int64_t)((double)fsize * ratio * 1.10) + (16ll << 20);
            if ((rc = alloc_blob(h, cap))) return give_up(rc);
        }
        if (u_end > cap) return give_up(1);                    // later groups inflate further than the first ones promised
        if (trace) lap("  members found, blob");
        const size_t at = h->gz_moff.size();
        h->gz_moff.resize(at + (size_t)nmem); h->gz_coff.resize(at + (size_t)nmem); h->gz_uoff.resize(at + (size_t)nmem);
        HIPCHK(hipMemcpyAsync(h->gz_moff.data() + at, d_mstart, (size_t)nmem * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(h->gz_coff.data() + at, d_coff, (size_t)nmem * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(h->gz_uoff.data() + at, d_uoff, (size_t)nmem * 8, hipMemcpyDeviceToHost, h->stream));
        if ((rc = bgzf_inflate_staged(h, d_c.p, d_coff, d_clen, d_uoff, d_isize, nmem, (int64_t)-1, flags[1], path,
                                      [&](int64_t m) { return h->gz_moff[at + (size_t)m]; }, M == 0 ? lap : quiet, M)))
            return give_up(rc);
        U = u_end; M += nmem;
        carry = next_carry;
        if (trace) { char what[64]; snprintf(what, sizeof what, "group %lld (%lld members)", (long long)g, (long long)nmem); lap(what); }
    }
    if ((rc = stg.finish())) { (void)give_up(0); return rc == 1 ? fail(FX_EIO, "read error on %s", path) : fail(FX_EDEVICE, "staging %s to the device failed", path); }
    if (M == 0 || U <= 0) return give_up(1);
    h->n = U;                                                  // what the blob holds (its allocation is a little larger)
    HIPCHK(hipMemsetAsync(h->d_data + U, 0, (size_t)std::min<int64_t>(2 * TILE, ((cap + TILE - 1) / TILE) * TILE + TILE - U), h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->bgzf = true;
    h->gz_mode = 1;
    h->gz_csize = fsize;
    lap("done");
    return FX_OK;
}

// A whole BGZF file with the member table made on the device (fx_bgzf_walk.hpp): stage the compressed bytes, find the members
// in HBM, inflate.  -> FX_OK; 1: not a file this path takes (another header layout, a chain that does not tile the file: the
// host walk decides); < 0: an error.
static int bgzf_open_on_device(fx_handle *h, int fd, int64_t fsize, const char *path) {
    static const bool off = [] { const char *e = getenv("FX_BGZF_HOST_WALK"); return e && atoi(e) != 0; }();
    if (off || fsize < BGZF_HDR + 8) return 1;
    uint8_t head[16];
    static const uint8_t sig[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0x00, 'B', 'C', 0x02, 0x00};
    if (pread(fd, head, 16, 0) != 16 || memcmp(head, sig, 16) != 0) return 1;
    static const bool trace = [] { const char *e = getenv("FX_TRACE"), *b = getenv("FX_TRACE_BGZF"); return (e && atoi(e) != 0) || (b && atoi(b) != 0); }();
    const auto T0 = std::chrono::steady_clock::now();
    const std::function<void(const char *)> lap = [&](const char *what) {
        if (trace) fprintf(stderr, "[fxgpu] bgzf %-22s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count());
    };
    int rc;
    {   // large files: the inflate behind the staging, group by group (FX_BGZF_GROUP bytes, a multiple of 4096; 0: never)
        const char *e = getenv("FX_BGZF_GROUP");
        int64_t group = e ? atoll(e) : (256ll << 20);       // tools/bgzf_group_probe.py: C4 opens in 37 ms (median) with 256 MiB, 40 with 128, 54 with 64, 49 all at once
        group &= ~4095ll;
        if (group >= 8192 && fsize >= 2 * group) {
            rc = bgzf_open_pipelined(h, fd, fsize, path, group);
            if (rc != 1) return rc;                           // 1: not for that path (a ratio that grew, a chain that breaks): all at once, below
        }
    }
    ScratchBuf<uint8_t> d_c;
    if ((rc = d_c.alloc(h->device, fsize + 48, h->stream))) return rc;
    HIPCHK(hipMemsetAsync(d_c.p + fsize, 0, 48, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));                  // the staging lanes copy on streams of their own
    if ((rc = stage_plain_file(h, fd, fsize, path, d_c.p, 0))) return rc;
    lap("staged");
    const int64_t ngran = (fsize + 4095) / 4096;
    const int64_t nchunks = (ngran + SCAN_CHUNK - 1) / SCAN_CHUNK;
    ScratchBuf<int64_t> d_i64;                                // off[ngran + 1] | sums[nchunks + 1]
    ScratchBuf<int32_t> d_cnt;
    if ((rc = d_i64.alloc(h->device, ngran + 1 + nchunks + 1, h->stream)) || (rc = d_cnt.alloc(h->device, ngran, h->stream))) return rc;
    int64_t *d_off = d_i64.p, *d_sums = d_i64.p + ngran + 1;
    hipLaunchKernelGGL(k_bgzf_sig_count, dim3(nblocks(ngran, BLOCK / 64)), dim3(BLOCK), 0, h->stream, (const uint8_t *)d_c.p, fsize, (int64_t)0, ngran, d_cnt.p);
    hipLaunchKernelGGL(k_cnt_chunk_sums, dim3((unsigned)nchunks), dim3(BLOCK), 0, h->stream, (const int32_t *)d_cnt.p, ngran, d_sums);
    hipLaunchKernelGGL(k_cnt_chunk_bases, dim3(1), dim3(BLOCK), 0, h->stream, d_sums, nchunks);
    hipLaunchKernelGGL(k_cnt_offsets, dim3((unsigned)nchunks), dim3(BLOCK), 0, h->stream, (const int32_t *)d_cnt.p, ngran, (const int64_t *)d_sums, d_off);
    int64_t nmem = 0;
    HIPCHK(hipMemcpyAsync(&nmem, d_off + ngran, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (nmem <= 0 || nmem > fsize / (BGZF_HDR + 8)) return 1;
    // the member table: mstart | coff | uoff[nmem + 1] | scan sums, then clen | isize | flags
    const int64_t mchunks = (nmem + SCAN_CHUNK - 1) / SCAN_CHUNK;
    ScratchBuf<int64_t> d_t64;
    ScratchBuf<int32_t> d_t32;
    if ((rc = d_t64.alloc(h->device, 3 * nmem + 1 + mchunks + 1, h->stream)) || (rc = d_t32.alloc(h->device, 2 * nmem + 2, h->stream))) return rc;
    int64_t *d_mstart = d_t64.p, *d_coff = d_t64.p + nmem, *d_uoff = d_t64.p + 2 * nmem, *d_msums = d_t64.p + 3 * nmem + 1;
    int32_t *d_clen = d_t32.p, *d_isize = d_t32.p + nmem, *d_flags = d_t32.p + 2 * nmem;      // flags: [0] bad, [1] longest deflate payload
    HIPCHK(hipMemsetAsync(d_flags, 0, 8, h->stream));
    hipLaunchKernelGGL(k_bgzf_sig_emit, dim3(nblocks(ngran, BLOCK / 64)), dim3(BLOCK), 0, h->stream, (const uint8_t *)d_c.p, fsize, (int64_t)0, ngran, (const int64_t *)d_off, d_mstart);
    hipLaunchKernelGGL(k_bgzf_member_rows, dim3(nblocks(nmem, BLOCK)), dim3(BLOCK), 0, h->stream, (const uint8_t *)d_c.p, fsize, (const int64_t *)d_mstart, nmem,
                       nmem, (int64_t)0, d_coff, d_clen, d_isize, (int *)d_flags, d_flags + 1);
    hipLaunchKernelGGL(k_cnt_chunk_sums, dim3((unsigned)mchunks), dim3(BLOCK), 0, h->stream, (const int32_t *)d_isize, nmem, d_msums);
    hipLaunchKernelGGL(k_cnt_chunk_bases, dim3(1), dim3(BLOCK), 0, h->stream, d_msums, mchunks);
    hipLaunchKernelGGL(k_cnt_offsets, dim3((unsigned)mchunks), dim3(BLOCK), 0, h->stream, (const int32_t *)d_isize, nmem, (const int64_t *)d_msums, d_uoff);
    HIPCHK(hipGetLastError());
    int32_t flags[2] = {0, 0};
    int64_t total = 0;
    HIPCHK(hipMemcpyAsync(flags, d_flags, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(&total, d_uoff + nmem, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    lap("member table (device)");
    if (flags[0] || total <= 0) return 1;                     // the chain does not tile the file: the host walk decides what this file is
    // the host's copy of the table (restart points, fx_gz_points; an error message): on its way while the members inflate
    h->gz_moff.resize((size_t)nmem); h->gz_coff.resize((size_t)nmem); h->gz_uoff.resize((size_t)nmem);
    HIPCHK(hipMemcpyAsync(h->gz_moff.data(), d_mstart, (size_t)nmem * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(h->gz_coff.data(), d_coff, (size_t)nmem * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(h->gz_uoff.data(), d_uoff, (size_t)nmem * 8, hipMemcpyDeviceToHost, h->stream));
    if ((rc = bgzf_inflate_staged(h, d_c.p, d_coff, d_clen, d_uoff, d_isize, nmem, total, flags[1], path,
                                  [&](int64_t m) { return h->gz_moff[(size_t)m]; }, lap))) {
        h->gz_moff.clear(); h->gz_coff.clear(); h->gz_uoff.clear();
        return rc;
    }
    h->bgzf = true;
    h->gz_mode = 1;
    h->gz_csize = fsize;
    return FX_OK;
}

// ------------------------------------------------------------------- single-stream gzip
// A single deflate stream is inflated by zlib on the host (bit-serial, one core).  While that happens the restart
// points of zran (indexed_gzip, the reference's random-access layer: index.c:68-70 spacing 1 MiB, window 32 KiB;
// export layout util.c:461-529) are captured at deflate-block boundaries: compressed offset, pending bits, uncompressed
// offset and the 32 KiB of output before it.  They go into the .fxi; the NEXT open of the file reads them back and
// inflates the segments between points in parallel (gzip_indexed_to_blob): every segment is an independent raw
// inflate primed with its window.
static const int64_t GZ_SPACING = 1048576, GZ_WINDOW = 32768;

struct GzSerial {
    z_stream z;
    const uint8_t *in = nullptr;
    int64_t nin = 0, fed = 0, last_cout = 0;
    bool open = false, done = false;
    fx_handle *h = nullptr;
    const uint8_t *prev_buf = nullptr;                         // the ring slot filled before this one (windows that straddle two slots)
    int64_t prev_fill = 0;
    int init(fx_handle *hh, const uint8_t *p, int64_t n) {
        memset(&z, 0, sizeof z);
        if (inflateInit2(&z, 47) != Z_OK) return -1;          // 32 + 15: gzip or zlib header, detected
        open = true; h = hh; in = p; nin = n;
        return 0;
    }
    ~GzSerial() { if (open) inflateEnd(&z); }
    void point(const uint8_t *buf, int64_t pos) {
        const int64_t cout = (int64_t)z.total_out + base_out;
        const bool first = h->gzp_cin.empty();
        if (!first && cout - last_cout < GZ_SPACING) return;
        h->gzp_cin.push_back((int64_t)z.total_in + base_in); h->gzp_cout.push_back(cout);
        h->gzp_bits.push_back((uint8_t)(z.data_type & 7));
        h->gzp_has.push_back(first && cout == 0 ? 0 : 1);     // the first point is the start of the deflate data: nothing before it
        last_cout = cout;
        if (first && cout == 0) return;
        const size_t o = h->gzp_win.size();
        h->gzp_win.resize(o + GZ_WINDOW, 0);
        uint8_t *w = h->gzp_win.data() + o;
        if (pos >= GZ_WINDOW) memcpy(w, buf + pos - GZ_WINDOW, GZ_WINDOW);
        else {
            const int64_t need = GZ_WINDOW - pos, from_prev = std::min(need, prev_fill);
            if (from_prev > 0) memcpy(w + (need - from_prev), prev_buf + prev_fill - from_prev, (size_t)from_prev);
            if (pos > 0) memcpy(w + need, buf, (size_t)pos);
        }
    }
    int64_t base_in = 0, base_out = 0;                         // totals of the members finished so far (inflateReset zeroes zlib's)
    // inflate into buf[0, cap): bytes produced (0 at the end of the stream), -1 on a data error
    int64_t fill(uint8_t *buf, int64_t cap) {
        int64_t pos = 0;
        while (pos < cap && !done) {
            if (z.avail_in == 0 && fed < nin) {
                const int64_t c = std::min<int64_t>(nin - fed, 1 << 30);
                z.next_in = const_cast<Bytef *>(in + fed); z.avail_in = (uInt)c; fed += c;
            }
            const uInt room = (uInt)std::min<int64_t>(cap - pos, 1 << 30);
            z.next_out = buf + pos; z.avail_out = room;
            const int ret = inflate(&z, Z_BLOCK);
            pos += (int64_t)(room - z.avail_out);
            if (ret == Z_STREAM_END) {
                // gzread semantics: further gzip members are part of the stream, anything else after the trailer is ignored
                const int64_t left = (int64_t)z.avail_in + (nin - fed);
                const uint8_t *nx = z.next_in;
                if (left >= 18 && nx[0] == 0x1f && nx[1] == 0x8b) {
                    base_in += (int64_t)z.total_in; base_out += (int64_t)z.total_out;
                    if (inflateReset(&z) != Z_OK) return -1;
                } else done = true;
                continue;
            }
            if (ret == Z_BUF_ERROR) {
                if (z.avail_in == 0 && fed >= nin) return -1;   // the stream ends in the middle of a block
                continue;
            }
            if (ret != Z_OK) return -1;
            if ((z.data_type & 128) && !(z.data_type & 64)) point(buf, pos);
        }
        return pos;
    }
};

// The parallel form: segment i = the bytes between restart point i and point i + 1 of the uncompressed stream.
static int gzip_indexed_to_blob(fx_handle *h, const uint8_t *in, int64_t nin, int64_t npts, const int64_t *cin, const int64_t *cout,
                                const uint8_t *bits, const uint8_t *has, const uint8_t *wins, int64_t usize) {
    if (npts < 1 || cout[0] != 0 || usize <= 0) return 1;
    std::vector<int64_t> woff((size_t)npts, -1);               // window of point i inside wins
    int64_t nw = 0;
    for (int64_t i = 0; i < npts; ++i) {
        if (cin[i] < 0 || cin[i] > nin || cout[i] > usize || (i && cout[i] <= cout[i - 1]) || (i && !has[i])) return 1;
        if (has[i]) woff[(size_t)i] = (nw++) * GZ_WINDOW;
    }
    int rc = alloc_blob(h, usize);
    if (rc) return rc;
    if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(FX_EDEVICE, "stream sync failed");
    // (each thread holds two pinned pieces: 64 threads = 1 GiB pinned while the open runs; PinPool keeps 256 MiB of them afterwards)
    const int T = (int)std::min<int64_t>(std::max(8u, std::min(64u, std::thread::hardware_concurrency() / 2)), npts);
    // What the serial path gets from zlib's gzip wrapper has to be checked by hand here (raw inflate, -15): the CRC-32 of
    // every segment's output (folded in order with crc32_combine) against the trailer, ISIZE, and that the deflate stream
    // ENDS where the last segment does.  A file with several gzip members: the members after the first are inflated with
    // the wrapper (zlib checks them); the folded CRC then describes no single trailer and only the end of the stream is checked.
    std::vector<uint32_t> seg_crc((size_t)npts, 0);
    std::atomic<int> members_seen(0);
    std::atomic<int64_t> trailer_at(-1);
    std::atomic<int64_t> next(0);
    std::atomic<int> err(0);                                   // 1: data does not match the index, 2: device
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&]() {
            if (hipSetDevice(h->device) != hipSuccess) { err.store(2); return; }
            uint8_t *pin[2] = {g_pins.get(), g_pins.get()};
            hipStream_t st = nullptr;
            hipEvent_t ev[2] = {nullptr, nullptr};
            bool used[2] = {false, false};
            bool ok = pin[0] && pin[1] && (st = g_lane_streams.get(h->device)) != nullptr &&
                      hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) == hipSuccess;
            if (!ok) err.store(2);
            int slot = 0;
            z_stream z;
            for (int64_t i; ok && !err.load() && (i = next.fetch_add(1)) < npts;) {
                const int64_t o0 = cout[i], o1 = i + 1 < npts ? cout[i + 1] : usize;
                memset(&z, 0, sizeof z);
                if (inflateInit2(&z, -15) != Z_OK) { err.store(1); break; }
                bool bad = false;
                if (bits[i]) bad = cin[i] < 1 || inflatePrime(&z, bits[i], in[cin[i] - 1] >> (8 - bits[i])) != Z_OK;
                if (!bad && has[i]) bad = inflateSetDictionary(&z, wins + woff[(size_t)i], (uInt)GZ_WINDOW) != Z_OK;
                int64_t ipos = cin[i], done = o0;
                bool wrapped = false;                                          // inside a later gzip member (zlib eats its trailer itself)
                while (!bad && done < o1) {
                    if (used[slot] && hipEventSynchronize(ev[slot]) != hipSuccess) { err.store(2); bad = true; break; }
                    const int64_t want = std::min<int64_t>(o1 - done, PIECE_BYTES);
                    int64_t got = 0;
                    while (!bad && got < want) {
                        if (z.avail_in == 0) {
                            const int64_t c = std::min<int64_t>(nin - ipos, 1 << 26);
                            if (c <= 0) { bad = true; break; }
                            z.next_in = const_cast<Bytef *>(in + ipos); z.avail_in = (uInt)c; ipos += c;
                        }
                        const uInt room = (uInt)(want - got);
                        z.next_out = pin[slot] + got; z.avail_out = room;
                        const int ret = inflate(&z, Z_NO_FLUSH);
                        got += (int64_t)(room - z.avail_out);
                        if (ret == Z_STREAM_END && got < want) {          // the next gzip member goes on: trailer, then its header
                            ipos -= (int64_t)z.avail_in;
                            if (!wrapped) ipos += 8;
                            inflateEnd(&z);
                            memset(&z, 0, sizeof z);
                            if (ipos + 18 > nin || inflateInit2(&z, 47) != Z_OK) { bad = true; break; }
                            wrapped = true;
                        } else if (ret != Z_OK && ret != Z_STREAM_END && ret != Z_BUF_ERROR) bad = true;
                        else if (ret == Z_BUF_ERROR && z.avail_in == 0 && ipos >= nin) bad = true;
                    }
                    if (bad) break;
                    seg_crc[(size_t)i] = (uint32_t)crc32(seg_crc[(size_t)i], pin[slot], (uInt)got);
                    if (hipMemcpyAsync(h->d_data + done, pin[slot], (size_t)got, hipMemcpyHostToDevice, st) != hipSuccess ||
                        hipEventRecord(ev[slot], st) != hipSuccess) { err.store(2); bad = true; break; }
                    used[slot] = true;
                    slot ^= 1;
                    done += got;
                }
                if (wrapped) members_seen.store(1);
                if (!bad && i == npts - 1) {                                  // the stream has to end here: Z_STREAM_END with no byte more
                    uint8_t extra[8];
                    int ret = Z_OK;
                    for (int tries = 0; tries < 4 && ret == Z_OK; ++tries) {
                        if (z.avail_in == 0) {
                            const int64_t c = std::min<int64_t>(nin - ipos, 1 << 26);
                            if (c <= 0) break;
                            z.next_in = const_cast<Bytef *>(in + ipos); z.avail_in = (uInt)c; ipos += c;
                        }
                        z.next_out = extra; z.avail_out = sizeof extra;
                        ret = inflate(&z, Z_NO_FLUSH);
                        if (z.avail_out != sizeof extra) { ret = Z_DATA_ERROR; break; }
                    }
                    if (ret != Z_STREAM_END) bad = true;
                    else if (!wrapped) trailer_at.store(ipos - (int64_t)z.avail_in);
                }
                inflateEnd(&z);
                if (bad && !err.load()) err.store(1);
            }
            if (st) (void)hipStreamSynchronize(st);
            for (int k = 0; k < 2; ++k) { if (ev[k]) (void)hipEventDestroy(ev[k]); if (pin[k]) g_pins.put(pin[k]); }
            if (st) g_lane_streams.put(h->device, st);
        });
    for (auto &x : th) x.join();
    if (err.load() == 2) return fail(FX_EDEVICE, "staging the inflated segments failed");
    if (!err.load() && !members_seen.load()) {                 // one member: its trailer describes the whole output
        const int64_t tp = trailer_at.load();
        if (tp < 0 || tp + 8 > nin) err.store(1);
        else {
            uint32_t crc = seg_crc[0];
            for (int64_t i = 1; i < npts; ++i) crc = (uint32_t)crc32_combine(crc, seg_crc[(size_t)i], (z_off_t)((i + 1 < npts ? cout[i + 1] : usize) - cout[i]));
            const uint8_t *t8 = in + tp;
            const uint32_t want = t8[0] | (t8[1] << 8) | (t8[2] << 16) | ((uint32_t)t8[3] << 24);
            const uint32_t isz = t8[4] | (t8[5] << 8) | (t8[6] << 16) | ((uint32_t)t8[7] << 24);
            if (crc != want || isz != (uint32_t)usize) err.store(1);
        }
    }
    if (err.load() == 1) {                                     // the index does not describe this file: inflate it serially instead
        (void)hipStreamSynchronize(h->stream);
        free_blob(h);
        return 1;
    }
    return FX_OK;
}

static int open_file_impl(const char *path, int device, fx_handle **out, int64_t npts, const int64_t *p_cin, const int64_t *p_cout,
                          const uint8_t *p_bits, const uint8_t *p_has, const uint8_t *p_wins, int64_t p_usize);

extern "C" int fx_open_file(const char *path, int device, fx_handle **out) {
    return open_file_impl(path, device, out, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
}

// fx_open_file for a gzip file whose restart points are known (the gzindex rows of its .fxi): the segments between the
// points are inflated by many host threads at once.  Points that do not fit the file fall back to the serial inflate
// (which captures fresh points).  Plain and BGZF files ignore the points.
extern "C" int fx_open_file_indexed(const char *path, int device, int64_t n_points, const int64_t *cmp_off, const int64_t *uncmp_off,
                                    const uint8_t *bits, const uint8_t *has_data, const uint8_t *windows, int64_t uncompressed_size,
                                    fx_handle **out) {
    if (n_points > 0 && (!cmp_off || !uncmp_off || !bits || !has_data)) return fail(FX_EINVAL, "null argument");
    if (!windows)                                            // windows holds GZ_WINDOW bytes for every point whose has_data is set
        for (int64_t i = 0; i < n_points; ++i)
            if (has_data[i]) return fail(FX_EINVAL, "restart point %lld has window data but no windows were passed", (long long)i);
    return open_file_impl(path, device, out, n_points, cmp_off, uncmp_off, bits, has_data, windows, uncompressed_size);
}

static int open_file_impl(const char *path, int device, fx_handle **out, int64_t npts, const int64_t *p_cin, const int64_t *p_cout,
                          const uint8_t *p_bits, const uint8_t *p_has, const uint8_t *p_wins, int64_t p_usize) {
    if (!path || !out) return fail(FX_EINVAL, "null argument");
    struct stat st;
    if (stat(path, &st) != 0 || !S_ISREG(st.st_mode)) return fail(FX_ENOENT, "the input file %s does not exists", path);
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(FX_ENOENT, "cannot open %s", path);
    unsigned char magic[4] = {0, 0, 0, 0};
    ssize_t got = pread(fd, magic, 4, 0);
    const bool gz = got == 4 && magic[0] == 0x1f && magic[1] == 0x8b && magic[2] == 0x08;   // util.c:307-325

    fx_handle *h = nullptr;
    int rc = new_handle(device, &h);
    if (rc) { close(fd); return rc; }
    h->gz = gz;
    Stager st_;

    auto bail = [&](int code) { close(fd); fx_close(h); return code; };

    if (!gz) {
        const int64_t n = (int64_t)st.st_size;
        static const bool trace_p = [] { const char *e = getenv("FX_TRACE"); return e && atoi(e) != 0; }();
        const auto tp0 = std::chrono::steady_clock::now();
        rc = alloc_blob(h, n);
        if (rc) return bail(rc);
        if (hipStreamSynchronize(h->stream) != hipSuccess)   // the pad memset precedes the copies of the staging streams
            return bail(fail(FX_EDEVICE, "stream sync failed"));
        const auto tp1 = std::chrono::steady_clock::now();
        rc = stage_plain_file(h, fd, n, path, h->d_data);
        if (rc) return bail(rc);
        g_open_laps[0] = std::chrono::duration<double>(tp1 - tp0).count();
        g_open_laps[1] = std::chrono::duration<double>(std::chrono::steady_clock::now() - tp1).count();
        if (trace_p) fprintf(stderr, "[fxgpu] open plain %.2f GB: blob %.1f ms, page cache -> pinned -> HBM %.1f ms\n", n / 1e9, g_open_laps[0] * 1e3, g_open_laps[1] * 1e3);
    } else {
        // BGZF (bgzip): every member inflates independently -> GPU (k_bgzf_inflate)
        {
            const int64_t fsize = (int64_t)st.st_size;
            // the member walk only touches the 18-byte header and the trailer of each member: map the file
            // (page cache) instead of copying 1 GB to the host first
            static const bool trace = [] { const char *e = getenv("FX_TRACE"), *b = getenv("FX_TRACE_BGZF"); return (e && atoi(e) != 0) || (b && atoi(b) != 0); }();
            const auto T0 = std::chrono::steady_clock::now();
            auto lap = [&](const char *what) {
                if (trace) fprintf(stderr, "[fxgpu] open %-27s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count());
            };
            int brc = bgzf_open_on_device(h, fd, fsize, path);      // the member table found in HBM (fx_bgzf_walk.hpp); 1: not that kind of file
            lap("(device walk)");
            if (brc == FX_OK) { close(fd); *out = h; return FX_OK; }
            if (brc < 0) return bail(brc);
            if (h->d_data || h->d_alloc) {                            // (a blob the attempt allocated: none on the ways it gives up)
                (void)hipStreamSynchronize(h->stream); free_blob(h);
            }
            void *mp = mmap(nullptr, (size_t)fsize, PROT_READ, MAP_PRIVATE, fd, 0);
            BgzfTable tab;
            brc = 1;
            if (mp != MAP_FAILED) {
                // (the walk touches 47 k pages of the mapping: 16-20 ms for C4.  Run in a thread of its own beside the staging
                // of the compressed bytes it made the staging four times slower -- page faults on the mapping against the
                // preads of the same file -- 116 ms for the pair instead of 45.)
                const bool is_bgzf = parse_bgzf((const uint8_t *)mp, fsize, tab);
                lap("member walk");
                if (is_bgzf) brc = bgzf_to_blob(h, fd, fsize, tab, path);
                (void)munmap(mp, (size_t)fsize);
                lap("inflated, scratch back");
            }
            if (brc == FX_OK) { close(fd); *out = h; return FX_OK; }
            if (brc < 0) return bail(brc);
            // brc == 1: not BGZF -> fall through to the single-stream path
        }
        // single-stream gzip: with restart points from the index file the segments are inflated in parallel ...
        const int64_t fsize = (int64_t)st.st_size;
        void *mp = mmap(nullptr, (size_t)fsize, PROT_READ, MAP_PRIVATE, fd, 0);
        if (mp == MAP_FAILED) return bail(fail(FX_EIO, "cannot map %s", path));
        (void)madvise(mp, (size_t)fsize, MADV_SEQUENTIAL);
        h->gz_csize = fsize;
        if (npts > 0) {
            const int prc = gzip_indexed_to_blob(h, (const uint8_t *)mp, fsize, npts, p_cin, p_cout, p_bits, p_has, p_wins, p_usize);
            if (prc == FX_OK) {
                (void)munmap(mp, (size_t)fsize);
                h->gzp_cin.assign(p_cin, p_cin + npts); h->gzp_cout.assign(p_cout, p_cout + npts);
                h->gzp_bits.assign(p_bits, p_bits + npts); h->gzp_has.assign(p_has, p_has + npts);
                h->gz_mode = 4;
                close(fd);
                if (hipStreamSynchronize(h->stream) != hipSuccess) { fx_close(h); return fail(FX_EDEVICE, "stream sync failed"); }
                *out = h;
                return FX_OK;
            }
            if (prc < 0) { (void)munmap(mp, (size_t)fsize); return bail(prc); }
        }
        // ... without: the first open on all cores of the host (fx_pgzip.hpp: block starts searched behind T cuts, the pieces
        // decoded with markers for what lies in front of them, resolved in order) straight into pinned pieces and HBM, with the
        // restart points for the index file; anything it is not sure of -> the serial path below.
        static const bool no_par = [] { const char *e = getenv("FX_GZIP_SERIAL"); return e && atoi(e) != 0; }();
        if (npts <= 0 && !no_par && fsize >= (32ll << 20) && fsize <= (6ll << 30)) {
            // The pieces leave through a FEW staging lanes (a stream, two pinned 8 MiB buffers, a mutex each) shared by all
            // worker threads: PCIe is one link, and a stream + two pinned buffers per thread meant ~200 hipHostMalloc /
            // hipHostFree calls and ~100 streams per open -- more time than the inflate itself (0.5 of 0.9 s for 1.5 GB).
            constexpr int N_LANES = 8;
            struct Lane { std::mutex mu; uint8_t *pin[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; bool used[2] = {false, false}; hipStream_t st = nullptr; int slot = 0; };
            std::vector<Lane> lanes(N_LANES);
            std::atomic<int> dev_err(0);
            static const bool trace_o = [] { const char *e = getenv("FX_TRACE"), *b = getenv("FX_TRACE_PGZ"); return (e && atoi(e) != 0) || (b && atoi(b) != 0); }();
            const auto o0 = std::chrono::steady_clock::now();
            auto olap = [&](const char *what) {
                if (trace_o) fprintf(stderr, "[fxgpu] gzip open %-24s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - o0).count());
            };
            auto alloc = [&](uint64_t total, int workers) {
                (void)workers;
                if (alloc_blob(h, (int64_t)total) || hipStreamSynchronize(h->stream) != hipSuccess) return false;
                for (Lane &k : lanes) {
                    k.pin[0] = g_pins.get(); k.pin[1] = g_pins.get();
                    if (!k.pin[0] || !k.pin[1] || hipStreamCreateWithFlags(&k.st, hipStreamNonBlocking) != hipSuccess ||
                        hipEventCreateWithFlags(&k.ev[0], hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&k.ev[1], hipEventDisableTiming) != hipSuccess) return false;
                }
                olap("blob + lanes");
                return true;
            };
            auto sink = [&](int w, uint64_t off, const uint8_t *data, size_t len) {
                Lane &k = lanes[(size_t)w % N_LANES];
                std::lock_guard<std::mutex> lk(k.mu);
                if (hipSetDevice(h->device) != hipSuccess) { dev_err.store(1); return false; }
                for (size_t a = 0; a < len; a += (size_t)PIECE_BYTES) {
                    const size_t m = std::min<size_t>((size_t)PIECE_BYTES, len - a);
                    if (k.used[k.slot] && hipEventSynchronize(k.ev[k.slot]) != hipSuccess) { dev_err.store(1); return false; }
                    memcpy(k.pin[k.slot], data + a, m);
                    if (hipMemcpyAsync(h->d_data + off + a, k.pin[k.slot], m, hipMemcpyHostToDevice, k.st) != hipSuccess ||
                        hipEventRecord(k.ev[k.slot], k.st) != hipSuccess) { dev_err.store(1); return false; }
                    k.used[k.slot] = true;
                    k.slot ^= 1;
                }
                return true;
            };
            pgz::Result res;
            const char *te = getenv("FX_PGZ_THREADS");                     // (experiments; read at every open)
            const int T = te && atoi(te) > 1 ? atoi(te) : (int)std::max(2u, std::min(128u, std::thread::hardware_concurrency() / 2));
            const bool ok = pgz::inflate_parallel((const uint8_t *)mp, (uint64_t)fsize, T, (uint64_t)GZ_SPACING, alloc, sink, res);
            olap("inflate_parallel");
            for (Lane &k : lanes) {
                if (k.st) { (void)hipSetDevice(h->device); (void)hipStreamSynchronize(k.st); (void)hipStreamDestroy(k.st); }
                for (int i = 0; i < 2; ++i) { if (k.ev[i]) (void)hipEventDestroy(k.ev[i]); if (k.pin[i]) g_pins.put(k.pin[i]); }
            }
            olap("lanes drained");
            // (res goes out of scope at the end of this block: only then are the pieces' buffers handed to the thread that unmaps them)
            if (dev_err.load()) { (void)munmap(mp, (size_t)fsize); return bail(fail(FX_EDEVICE, "staging the inflated pieces of %s failed", path)); }
            if (ok) {
                (void)munmap(mp, (size_t)fsize);
                h->gzp_cin.assign(res.pt_cin.begin(), res.pt_cin.end()); h->gzp_cout.assign(res.pt_cout.begin(), res.pt_cout.end());
                h->gzp_bits = res.pt_bits; h->gzp_has = res.pt_has; h->gzp_win = std::move(res.pt_win);
                h->gz_mode = 3;
                close(fd);
                olap("unmapped, points kept");
                if (hipStreamSynchronize(h->stream) != hipSuccess) { fx_close(h); return fail(FX_EDEVICE, "stream sync failed"); }
                olap("done");
                *out = h;
                return FX_OK;
            }
            if (h->d_data && h->owns) { (void)hipStreamSynchronize(h->stream); free_blob(h); }   // (it had got as far as the blob)
        }
        // ... else (or with points that do not fit): serial (zlib on the host), the inflated bytes stream through a
        // pinned ring into a growing blob and the restart points are captured on the way.
        rc = st_.init();
        if (rc) { (void)munmap(mp, (size_t)fsize); return bail(rc); }
        GzSerial gs;
        if (gs.init(h, (const uint8_t *)mp, fsize)) { (void)munmap(mp, (size_t)fsize); return bail(fail(FX_EIO, "inflateInit failed for %s", path)); }
        int64_t cap = std::max<int64_t>((int64_t)st.st_size * 5, STAGE_BYTES), n = 0;
        uint8_t *d = nullptr;
        hipError_t e = dev_malloc((void **)&d, (size_t)cap + 2 * TILE);
        if (e != hipSuccess) { (void)munmap(mp, (size_t)fsize); return bail(fail(FX_ENOMEM, "hipMalloc: %s", hipGetErrorString(e))); }
        int slot = 0;
        for (;;) {
            if (st_.used[slot]) (void)hipEventSynchronize(st_.ev[slot]);
            const int64_t fill = gs.fill(st_.pin[slot], STAGE_BYTES);
            if (fill < 0) { (void)munmap(mp, (size_t)fsize); (void)hipFree(d); return bail(fail(FX_EIO, "gzip inflate error in %s", path)); }
            if (fill == 0) break;
            if (n + fill > cap) {           // grow: allocate bigger, device-to-device copy
                int64_t ncap = std::max(cap * 2, n + fill);
                uint8_t *nd = nullptr;
                e = dev_malloc((void **)&nd, (size_t)ncap + 2 * TILE);
                if (e == hipSuccess) e = hipMemcpyAsync(nd, d, (size_t)n, hipMemcpyDeviceToDevice, h->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
                if (e != hipSuccess) { (void)munmap(mp, (size_t)fsize); (void)hipFree(d); return bail(fail(FX_ENOMEM, "grow: %s", hipGetErrorString(e))); }
                (void)hipFree(d);
                d = nd; cap = ncap;
            }
            e = hipMemcpyAsync(d + n, st_.pin[slot], (size_t)fill, hipMemcpyHostToDevice, h->stream);
            if (e == hipSuccess) e = hipEventRecord(st_.ev[slot], h->stream);
            if (e != hipSuccess) { (void)munmap(mp, (size_t)fsize); (void)hipFree(d); return bail(fail(FX_EDEVICE, "H2D: %s", hipGetErrorString(e))); }
            st_.used[slot] = true;
            gs.prev_buf = st_.pin[slot]; gs.prev_fill = fill;
            n += fill;
            slot = (slot + 1) % NSTAGE;
            if (fill < STAGE_BYTES) break;
        }
        (void)munmap(mp, (size_t)fsize);
        h->d_data = d; h->owns = true; h->n = n;
        h->gz_mode = 2;
        e = hipMemsetAsync(d + n, 0, (size_t)(cap + 2 * TILE - n), h->stream);
        if (e != hipSuccess) return bail(fail(FX_EDEVICE, "memset: %s", hipGetErrorString(e)));
    }
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return bail(fail(FX_EDEVICE, "stream sync: %s", hipGetErrorString(e)));
    close(fd);
    *out = h;
    return FX_OK;
}

// What kind of stream a file holds and how long it is once inflated: 0 plain, 1 BGZF (sum of the members' ISIZE from
// a walk over their headers -- no inflation), 2 single-stream gzip (length unknown without inflating it: -1).
extern "C" int fx_stream_size(const char *path, int64_t *n_bytes, int *kind) {
    if (!path || !n_bytes || !kind) return fail(FX_EINVAL, "null argument");
    struct stat st;
    if (stat(path, &st) != 0 || !S_ISREG(st.st_mode)) return fail(FX_ENOENT, "the input file %s does not exists", path);
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(FX_ENOENT, "cannot open %s", path);
    unsigned char magic[4] = {0, 0, 0, 0};
    const bool gz = pread(fd, magic, 4, 0) == 4 && magic[0] == 0x1f && magic[1] == 0x8b && magic[2] == 0x08;
    *kind = 0; *n_bytes = (int64_t)st.st_size;
    if (gz) {
        *kind = 2; *n_bytes = -1;
        void *mp = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (mp != MAP_FAILED) {
            BgzfTable tab;
            if (parse_bgzf((const uint8_t *)mp, (int64_t)st.st_size, tab)) { *kind = 1; *n_bytes = tab.total; }
            (void)munmap(mp, (size_t)st.st_size);
        }
    }
    close(fd);
    return FX_OK;
}

// One byte-range shard of a file ("each GPU ingests only its own range from the host"): bytes
// [off, off + len + halo) of the UNCOMPRESSED stream, clamped to its end, become the handle's blob, and the shard
// context is set from the file itself (base = off, prev_byte = the byte before it, is_last, halo).  Plain files: only
// that range is read.  BGZF: only the members that cover it are read, staged and inflated.  Single-stream gzip cannot
// be entered in the middle (FX_EINVAL unless the range starts at 0 and covers everything: "replicas only").
extern "C" int fx_open_file_range(const char *path, int64_t off, int64_t len, int64_t halo, int device, fx_handle **out) {
    if (!path || !out || off < 0 || len < 0 || halo < 0) return fail(FX_EINVAL, "bad argument");
    int64_t total = 0;
    int kind = 0;
    int rc = fx_stream_size(path, &total, &kind);
    if (rc) return rc;
    if (kind == 2) return fail(FX_EINVAL, "%s is a single gzip stream: it cannot be opened by byte range (bgzip it, or open it whole)", path);
    if (off > total) return fail(FX_ERANGE, "range starts past the end of the stream (%lld > %lld)", (long long)off, (long long)total);
    const int64_t core = std::min(len, total - off);
    const int64_t n = std::min(len + halo, total - off);    // bytes held
    if (n <= 0) return fail(FX_EFORMAT, "empty range");
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(FX_ENOENT, "cannot open %s", path);
    fx_handle *h = nullptr;
    rc = new_handle(device, &h);
    if (rc) { close(fd); return rc; }
    auto bail = [&](int code) { close(fd); fx_close(h); return code; };
    int prev = '\n';
    if (kind == 0) {
        if ((rc = alloc_blob(h, n))) return bail(rc);
        if (hipStreamSynchronize(h->stream) != hipSuccess) return bail(fail(FX_EDEVICE, "stream sync failed"));
        if ((rc = stage_plain_file(h, fd, n, path, h->d_data, off))) return bail(rc);
        unsigned char c = '\n';
        if (off > 0 && pread(fd, &c, 1, (off_t)(off - 1)) != 1) return bail(fail(FX_EIO, "read error on %s", path));
        prev = c;
    } else {
        struct stat st;
        if (fstat(fd, &st) != 0) return bail(fail(FX_EIO, "stat failed"));
        void *mp = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (mp == MAP_FAILED) return bail(fail(FX_EIO, "cannot map %s", path));
        BgzfTable tab;
        const bool okb = parse_bgzf((const uint8_t *)mp, (int64_t)st.st_size, tab);
        (void)munmap(mp, (size_t)st.st_size);
        if (!okb) return bail(fail(FX_EFORMAT, "%s is not BGZF", path));
        // members that cover [off - 1, off + n): the byte before the range comes along (prev_byte) when there is one
        const int64_t lo = off > 0 ? off - 1 : 0;
        int64_t m0 = (int64_t)(std::upper_bound(tab.uoff.begin(), tab.uoff.end(), lo) - tab.uoff.begin()) - 1;
        int64_t m1 = (int64_t)(std::lower_bound(tab.uoff.begin(), tab.uoff.end(), off + n) - tab.uoff.begin());
        if (m0 < 0) m0 = 0;
        if (m1 <= m0) m1 = m0 + 1;
        if ((rc = bgzf_to_blob(h, fd, (int64_t)st.st_size, tab, path, m0, m1))) return bail(rc);
        const int64_t skip = off - tab.uoff[(size_t)m0];   // the blob starts `skip` bytes into the first member inflated
        h->d_alloc = h->d_data;
        if (off > 0) {
            unsigned char c = '\n';
            if (hipMemcpy(&c, h->d_data + skip - 1, 1, hipMemcpyDeviceToHost) != hipSuccess) return bail(fail(FX_EDEVICE, "D2H failed"));
            prev = c;
        }
        if (skip & 15) {                                   // kernels load 16-byte chunks from the blob's start: keep it aligned --
            uint8_t *tmp = nullptr;                        // the range moves to the front of the allocation (through a copy: the spans overlap)
            if (dev_malloc((void **)&tmp, (size_t)n) != hipSuccess) return bail(fail(FX_ENOMEM, "hipMalloc(%lld B) failed", (long long)n));
            hipError_t e = hipMemcpy(tmp, h->d_data + skip, (size_t)n, hipMemcpyDeviceToDevice);
            if (e == hipSuccess) e = hipMemcpy(h->d_data, tmp, (size_t)n, hipMemcpyDeviceToDevice);
            if (e == hipSuccess && h->n > n) e = hipMemset(h->d_data + n, 0, (size_t)std::min<int64_t>(h->n - n, 2 * TILE));
            (void)hipFree(tmp);
            if (e != hipSuccess) return bail(fail(FX_EDEVICE, "device copy failed: %s", hipGetErrorString(e)));
        } else {
            // the blob is the inflated members from `skip` on: what follows the range is live stream data, not the zero pad
            // alloc_blob leaves behind a blob -- make it one (kernels may look at the pad, never past the allocation)
            const int64_t after = h->n - (skip + n);
            if (after > 0 && hipMemset(h->d_data + skip + n, 0, (size_t)std::min<int64_t>(after, 2 * TILE)) != hipSuccess)
                return bail(fail(FX_EDEVICE, "memset failed"));
            h->d_data += skip;
        }
        h->n = n;
        h->gz = true;
    }
    close(fd);
    h->gz = kind != 0;
    if ((rc = fx_set_shard(h, off, prev, off + core >= total))) { fx_close(h); return rc; }
    if (n > core && (rc = fx_set_halo(h, n - core))) { fx_close(h); return rc; }
    if (hipStreamSynchronize(h->stream) != hipSuccess) { fx_close(h); return fail(FX_EDEVICE, "stream sync failed"); }
    *out = h;
    return FX_OK;
}

extern "C" int fx_open_host(const void *data, int64_t nbytes, int device, fx_handle **out) {
    if ((!data && nbytes) || nbytes < 0 || !out) return fail(FX_EINVAL, "bad argument");
    fx_handle *h = nullptr;
    int rc = new_handle(device, &h);
    if (rc) return rc;
    rc = alloc_blob(h, nbytes);
    if (rc) { fx_close(h); return rc; }
    if (nbytes) {
        hipError_t e = hipMemcpyAsync(h->d_data, data, (size_t)nbytes, hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { fx_close(h); return fail(FX_EDEVICE, "H2D: %s", hipGetErrorString(e)); }
    }
    const unsigned char *m = (const unsigned char *)data;
    h->gz = false;
    (void)m;
    *out = h;
    return FX_OK;
}

extern "C" int fx_open_device(const void *dptr, int64_t nbytes, int device, fx_handle **out) {
    if (!dptr || nbytes < 0 || !out) return fail(FX_EINVAL, "bad argument");
    if (((uintptr_t)dptr & 15) != 0) return fail(FX_EINVAL, "device blob must be 16-byte aligned");
    fx_handle *h = nullptr;
    int rc = new_handle(device, &h);
    if (rc) return rc;
    h->d_data = (uint8_t *)dptr;
    h->owns = false;
    h->n = nbytes;
    *out = h;
    return FX_OK;
}

extern "C" int fx_read_bytes(fx_handle *h, int64_t off, int64_t n, void *dst) {
    if (!h || !dst) return fail(FX_EINVAL, "null argument");
    int rc = use_device(h);
    if (rc) return rc;
    off -= h->base;
    if (n < 0 || off < 0 || off > h->n) return fail(FX_ERANGE, "read_bytes range [%lld,+%lld) outside stream of %lld bytes", (long long)off, (long long)n, (long long)h->n);
    const int64_t m = std::min(n, h->n - off);
    if (m < n) memset((char *)dst + m, 0, (size_t)(n - m));
    if (m > 0) {
        HIPCHK(hipMemcpyAsync(dst, h->d_data + off, (size_t)m, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return FX_OK;
}

extern "C" int fx_first_byte(fx_handle *h, int *out) {
    // fasta_validator / fastq_validator (util.c:95-150): first byte that is not isspace()
    if (!h || !out) return fail(FX_EINVAL, "null argument");
    *out = -1;
    unsigned char buf[4096];
    for (int64_t off = 0; off < h->n; off += (int64_t)sizeof buf) {
        const int64_t m = std::min<int64_t>(sizeof buf, h->n - off);
        int rc = fx_read_bytes(h, h->base + off, m, buf);
        if (rc) return rc;
        for (int64_t i = 0; i < m; ++i) {
            unsigned char c = buf[i];
            if (!(c == ' ' || (c >= 9 && c <= 13))) { *out = c; return FX_OK; }
        }
    }
    return FX_OK;
}

// -------------------------------------------------------------------- scan

// ------------------------------------------------------------- FASTA build
static ScanCtx scan_ctx(const fx_handle *h) {
    ScanCtx x;
    x.data = h->d_data; x.n = h->n; x.gbase = h->base; x.ngran = h->ngran;
    x.go = h->gran.p; x.nl_prefix = h->nl_prefix.p; x.hdr_prefix = h->hdr_prefix.p; x.prevnl = h->prevnl.p;
    return x;
}
static FastaCols fasta_cols(fx_handle *h) {
    FastaCols c;
    c.hoff = h->hdr.p; c.boff = h->fa_boff.p; c.blen = h->fa_blen.p; c.slen = h->fa_slen.p; c.llen = h->fa_llen.p;
    c.hdr_line = h->fa_hdr_line.p; c.elen = h->fa_elen.p; c.dlen = h->fa_dlen.p; c.name_len = h->fa_name_len.p;
    c.norm = h->fa_norm.p; c.bad = h->fa_bad.p; c.reg = h->fa_reg.p;
    return c;
}
static int alloc_fasta_table(fx_handle *h, int64_t cap) {
    int rc;
    if ((rc = h->hdr.alloc(cap)) || (rc = h->fa_hdr_line.alloc(cap)) || (rc = h->fa_boff.alloc(cap)) ||
        (rc = h->fa_blen.alloc(cap)) || (rc = h->fa_slen.alloc(cap)) || (rc = h->fa_llen.alloc(cap)) ||
        (rc = h->fa_elen.alloc(cap)) || (rc = h->fa_norm.alloc(cap)) || (rc = h->fa_dlen.alloc(cap)) ||
        (rc = h->fa_name_len.alloc(cap)) || (rc = h->fa_bad.alloc(cap)) || (rc = h->fa_reg.alloc(cap)))
        return rc;
    return FX_OK;
}

// ctl layout (64 words): [0..8) Totals, [8] / [9] = header-granule / irregular list counts (u32), [16..44) shard summary
static Totals *ctl_totals(fx_handle *h) { return (Totals *)h->ctl.p; }
static uint32_t *ctl_counter(fx_handle *h, int i) { return (uint32_t *)(h->ctl.p + 8 + i); }

// Granule summaries + prefixes of the resident stream.  MODE 0: FASTA (line-length sets, header lines);
// MODE 1: FASTQ (newline count / first / last only).  Enqueues only; h->ctl holds the totals afterwards.
static int comp_run_granules(int64_t ngran) {       // granules per run of the composition kernels (a multiple of the pipeline depth)
    // 8 = 32 KiB measured best on 3 GB (0.525 ms; 16: 0.550, 4: 0.569), shorter runs for small inputs so that the machine still fills
    return ngran >= 65536 ? 4 * COMP_DEPTH : ngran >= 16384 ? 2 * COMP_DEPTH : COMP_DEPTH;
}

template <int MODE>
static int granule_pass(fx_handle *h, bool fq_lines = false, bool with_comp = false) {
    int rc;
    const int64_t nfull = h->n / GRAN, ngran = nfull + 1;
    const bool small = ngran <= (2ll << 20);                  // up to 8 GB of stream: 256 granules per chunk, else 1024
    const int64_t cg = small ? 256 : 1024, nchunks = (ngran + cg - 1) / cg;
    h->ngran = ngran;
    if ((rc = h->gran.alloc(ngran)) || (rc = h->hdr_grans.alloc(ngran)) || (MODE == 0 && (rc = h->irr_grans.alloc(ngran))) ||
        (rc = h->chunks.alloc(nchunks)) ||
        (rc = h->ctl.alloc(64)) || (rc = h->nl_prefix.alloc(ngran + 1)) || (rc = h->hdr_prefix.alloc(ngran + 1)) ||
        (rc = h->prevnl.alloc(ngran + 1)))
        return rc;
    if (!h->pin_tot) HIPCHK(hipHostMalloc((void **)&h->pin_tot, sizeof(Totals), hipHostMallocDefault));
    HIPCHK(hipMemsetAsync(h->ctl.p, 0, 64 * sizeof(unsigned long long), h->stream));
    GranList hgl{h->hdr_grans.p, ctl_counter(h, 0)};
    const int SCAN_WG = 512;                                  // 8 waves = 8 granules per workgroup (tools/scanbench2.hip)
    if (nfull > 0 && MODE == 1 && fq_lines) {                 // FASTQ, one-read build: the count pass also writes the line records
        if ((rc = h->fq_lines.alloc((nfull + FQR_G - 1) / FQR_G * FQR_G * FQL_CAP))) return rc;      // a slot of FQR_G * FQL_CAP records per FQR_G granules
        if (with_comp) {                                      // index and composition in one read of the stream (fx_fastq_stream.hpp)
            const int64_t nruns = (nfull + FQLC_G - 1) / FQLC_G;
            if ((rc = h->fq_runs.alloc(nruns)) || (rc = h->fq_acc_build.alloc(1)) || (rc = h->fq_rej.alloc(nruns + 1))) return rc;
            static const int64_t grid_cap = [] { const char *e = getenv("FX_FQ_FUSED_GRID"); return e && atoll(e) > 0 ? atoll(e) : 6144ll; }();
            static const int force_crlf = [] { const char *e = getenv("FX_FQ_CRLF"); return e ? atoi(e) : -1; }();      // 0 / 1: experiments
            if (force_crlf >= 0 ? force_crlf != 0 : h->fq_crlf)
                FX_LAUNCH(h, K_FASTQ_LINES, k_fastq_lines_comp<true>, dim3((unsigned)std::min<int64_t>(nblocks(nruns, BLOCK / 64), grid_cap)), dim3(BLOCK), h->d_data,
                          h->n, h->prev_byte, nfull, h->gran.p, h->fq_lines.p, hgl, h->fq_runs.p, nruns);
            else
                FX_LAUNCH(h, K_FASTQ_LINES, k_fastq_lines_comp<false>, dim3((unsigned)std::min<int64_t>(nblocks(nruns, BLOCK / 64), grid_cap)), dim3(BLOCK), h->d_data,
                          h->n, h->prev_byte, nfull, h->gran.p, h->fq_lines.p, hgl, h->fq_runs.p, nruns);
        } else
        FX_LAUNCH(h, K_FASTQ_LINES, k_fastq_lines, dim3(nblocks(nfull, (BLOCK / 64) * FQL_G)), dim3(BLOCK), h->d_data, h->n, h->prev_byte, nfull,
                  h->gran.p, h->fq_lines.p, hgl);
    } else if (nfull > 0 && MODE == 0 && with_comp) {         // the scan and the composition counters in one read (fx_scancomp.hpp)
        const int gpw = comp_run_granules(ngran);
        const int64_t runs = (ngran + gpw - 1) / gpw;
        if ((rc = h->comp_runs.alloc(runs * RUN_WORDS))) return rc;
        HIPCHK(hipMemsetAsync(h->comp_runs.p, 0, (size_t)runs * RUN_WORDS * 4, h->stream));
        FX_LAUNCH(h, K_SCAN_COMP, k_scan_comp, dim3(nblocks(runs, COMP_WPB)), dim3(COMP_WPB * 64), h->d_data, h->n, h->prev_byte,
                  (int)h->is_last, nfull, h->gran.p, hgl, gpw, h->comp_runs.p);
        h->comp_gpw = gpw;
        h->comp_runs_valid = true;
    } else if (nfull > 0)
        FX_LAUNCH(h, K_SPAN_SCAN, (k_span_scan<MODE>), dim3(nblocks((nfull + SCAN_GPW - 1) / SCAN_GPW * 64, SCAN_WG)), dim3(SCAN_WG), h->d_data, h->n,
                  h->prev_byte, (int)h->is_last, nfull, h->gran.p, hgl);
    if (small) {
        FX_LAUNCH(h, K_GRAN_REDUCE, (k_gran_reduce<MODE, 256>), dim3((unsigned)nchunks), dim3(256), h->d_data, h->n, h->prev_byte,
                  (int)h->is_last, hgl, h->gran.p, ngran, h->base, h->chunks.p);
        FX_LAUNCH(h, K_GRAN_PREFIX, (k_gran_prefix<256>), dim3((unsigned)nchunks), dim3(256), h->gran.p, ngran, h->base,
                  h->chunks.p, ctl_totals(h), h->nl_prefix.p, h->hdr_prefix.p, h->prevnl.p);
    } else {
        FX_LAUNCH(h, K_GRAN_REDUCE, (k_gran_reduce<MODE, 1024>), dim3((unsigned)nchunks), dim3(1024), h->d_data, h->n, h->prev_byte,
                  (int)h->is_last, hgl, h->gran.p, ngran, h->base, h->chunks.p);
        FX_LAUNCH(h, K_GRAN_PREFIX, (k_gran_prefix<1024>), dim3((unsigned)nchunks), dim3(1024), h->gran.p, ngran, h->base,
                  h->chunks.p, ctl_totals(h), h->nl_prefix.p, h->hdr_prefix.p, h->prevnl.p);
    }
    if (MODE == 1 && fq_lines && with_comp && nfull > 0) {   // every run's guess against the prefixes, the runs' counts added up
        FastqAcc init;
        memset(&init, 0, sizeof init);
        init.minqs = 104; init.maxqs = 33;                    // fastq.c:667-668
        HIPCHK(hipMemcpyAsync(h->fq_acc_build.p, &init, sizeof init, hipMemcpyHostToDevice, h->stream));
        const int64_t nruns = (nfull + FQLC_G - 1) / FQLC_G;
        HIPCHK(hipMemsetAsync(h->fq_rej.p, 0, 4, h->stream));
        FX_LAUNCH(h, K_FASTQ_COMP, k_fastq_comp_reduce, dim3((unsigned)std::min<int64_t>(nblocks(nruns, BLOCK), 1024)), dim3(BLOCK), (const FqRun *)h->fq_runs.p,
                  nruns, (const int64_t *)h->nl_prefix.p, (int64_t)0, h->d_data, h->n, nfull, h->fq_acc_build.p, h->fq_rej.p);
    }
    HIPCHK(hipGetLastError());
    return FX_OK;
}

// records: everything after the granule pass, for the current table capacity (enqueue only)
static int enqueue_records(fx_handle *h, int full_name) {
    const ScanCtx x = scan_ctx(h);
    const int64_t cap = h->hdr.cap, ngran = h->ngran;
    const FastaCols c = fasta_cols(h);
    const RecView rv{h->fa_boff.p, h->fa_llen.p, h->fa_dlen.p, h->fa_bad.p, h->hdr.p};
    const GranList hgl{h->hdr_grans.p, ctl_counter(h, 0)}, irr{h->irr_grans.p, ctl_counter(h, 1)};
    FX_LAUNCH(h, K_HDR_REC, k_hdr_rec, dim3(2048), dim3(BLOCK), x, h->prev_byte, (int)h->is_last, full_name, hgl, c, cap);
    FX_LAUNCH(h, K_GRAN_LINES, k_gran_lines, dim3(nblocks(ngran, BLOCK)), dim3(BLOCK), x, rv, cap, irr);
    FX_LAUNCH(h, K_GRAN_EXACT, k_gran_exact, dim3(2048), dim3(BLOCK), x, rv, cap, irr, (int)h->is_last, h->prev_byte);
    FX_LAUNCH(h, K_FASTA_FINALIZE, k_fasta_finalize2, dim3(nblocks(cap, BLOCK)), dim3(BLOCK), cap, c, ctl_totals(h), h->d_data, h->n, h->base);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h->pin_tot, ctl_totals(h), sizeof(Totals), hipMemcpyDeviceToHost, h->stream));
    return FX_OK;
}

// Enqueue the whole build on the handle's stream and return: granule pass, prefixes, records.  The tables are
// sized from an estimate (previous build, else 4096 records), so nothing here needs a host round trip; device-side
// consumers (fx_fasta_fetch, fx_shard_summary_dev, fx_fasta_stitch_dev) may be enqueued right behind it -- they read
// the record count from device memory.  fx_fasta_build_end (or any call that needs host-side totals) completes it.
extern "C" int fx_fasta_build_begin(fx_handle *h, int full_name) {
    if (!h) return fail(FX_EINVAL, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    if (h->n <= 0) return fail(FX_EFORMAT, "empty input");
    h->scanned = false;
    h->nm_kind = -1;
    const bool with_comp = (full_name & 2) != 0;   // bit 1: the composition counters ride on the scan (fx_fasta_comp* then reads nothing twice)
    full_name &= 1;
    h->comp_runs_valid = false;
    if ((rc = granule_pass<0>(h, false, with_comp))) return rc;      // the one pass over the stream: granule summaries + prefixes
    if (h->hdr.cap < 4096 && (rc = alloc_fasta_table(h, 4096))) return rc;
    if ((rc = enqueue_records(h, full_name))) return rc;
    h->build_pending = true;
    h->pending_full_name = full_name;
    h->fasta_built = true;
    return FX_OK;
}

// Wait for a pending build and read its totals; if more header lines turned up than the tables hold, grow them
// and redo only the (cheap) record part.
static int finish_build(fx_handle *h) {
    if (!h->build_pending) return FX_OK;
    int rc = use_device(h);
    if (rc) return rc;
    Totals tot;
    for (;;) {
        HIPCHK(hipStreamSynchronize(h->stream));
        tot = *h->pin_tot;
        if (tot.n_hdr <= h->hdr.cap) break;
        if ((rc = alloc_fasta_table(h, tot.n_hdr + tot.n_hdr / 16 + 16))) return rc;
        HIPCHK(hipMemsetAsync(&ctl_totals(h)->seq_len, 0, 8, h->stream));
        HIPCHK(hipMemsetAsync(ctl_counter(h, 1), 0, 4, h->stream));                   // the irregular list is rebuilt
        if ((rc = enqueue_records(h, h->pending_full_name))) return rc;
    }
    h->build_pending = false;
    h->n_hdr = tot.n_hdr;
    h->n_nl = tot.n_nl;
    h->fa_seqlen = tot.seq_len;
    if (tot.n_hdr <= 0 && h->base == 0 && h->is_last) { h->fasta_built = false; return fail(FX_EFORMAT, "no FASTA header line ('>') found"); }
    // (a shard that lies entirely inside one record has an empty local table; its summary is still valid)
    return FX_OK;
}

extern "C" int fx_fasta_build_end(fx_handle *h, fx_fasta_summary *out) {
    if (!h) return fail(FX_EINVAL, "null handle");
    if (!h->fasta_built) return fail(FX_ESTATE, "fx_fasta_build_begin has not run");
    int rc = finish_build(h);
    if (rc) return rc;
    if (out) { out->n_seq = h->n_hdr; out->seq_len = h->fa_seqlen; out->n_lines = h->n_nl; out->n_bytes = h->n; }
    return FX_OK;
}

extern "C" int fx_fasta_build(fx_handle *h, int full_name, fx_fasta_summary *out) {
    int rc = fx_fasta_build_begin(h, full_name);
    if (rc) return rc;
    return fx_fasta_build_end(h, out);
}

// The resident record table from an existing .fxi (pyfastx_load_index, index.c:391-429): batched fetches by
// (record id, start, stop) then work without re-scanning the file.
extern "C" int fx_fasta_set_table(fx_handle *h, int64_t n, const int64_t *boff, const int64_t *blen, const int64_t *slen,
                                  const int64_t *llen, const int32_t *elen, const int32_t *norm) {
    if (!h || n < 0 || (n > 0 && (!boff || !blen || !slen || !llen || !elen || !norm))) return fail(FX_EINVAL, "bad argument");
    int rc = use_device(h);
    if (rc) return rc;
    if ((rc = alloc_fasta_table(h, std::max<int64_t>(n, 1)))) return rc;
    auto up = [&](void *d, const void *s_, size_t bytes) { return hipMemcpyAsync(d, s_, bytes, hipMemcpyHostToDevice, h->stream); };
    if (n) {
        HIPCHK(up(h->fa_boff.p, boff, (size_t)n * 8)); HIPCHK(up(h->fa_blen.p, blen, (size_t)n * 8));
        HIPCHK(up(h->fa_slen.p, slen, (size_t)n * 8)); HIPCHK(up(h->fa_llen.p, llen, (size_t)n * 8));
        HIPCHK(up(h->fa_elen.p, elen, (size_t)n * 4)); HIPCHK(up(h->fa_norm.p, norm, (size_t)n * 4));
        // the .fxi has no column for "line-regular": decided here, once, from the rows and one byte of the stream each
        hipLaunchKernelGGL(k_line_regular, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, h->stream, h->d_data, h->n, h->base, n, h->fa_boff.p,
                           h->fa_blen.p, h->fa_slen.p, h->fa_llen.p, h->fa_elen.p, h->fa_norm.p, h->fa_reg.p);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    h->n_hdr = n;
    h->fasta_built = true;
    return FX_OK;
}

template <class T>
static int copy_out(fx_handle *h, int where, T *dst, const T *src, int64_t n) {
    if (!dst || n <= 0) return FX_OK;
    HIPCHK(hipMemcpyAsync(dst, src, (size_t)n * sizeof(T), where == FX_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
    return FX_OK;
}

extern "C" int fx_fasta_table(fx_handle *h, int where, int64_t *hoff, int64_t *boff, int64_t *blen, int64_t *slen,
                              int64_t *llen, int32_t *elen, int32_t *norm, int32_t *dlen, int32_t *name_len) {
    if (!h) return fail(FX_EINVAL, "null handle");
    if (!h->fasta_built) return fail(FX_ESTATE, "fx_fasta_build has not run");
    int rc = use_device(h);
    if (!rc) rc = finish_build(h);
    if (rc) return rc;
    const int64_t n = h->n_hdr;
    if ((rc = copy_out(h, where, hoff, h->hdr.p, n)) || (rc = copy_out(h, where, boff, h->fa_boff.p, n)) ||
        (rc = copy_out(h, where, blen, h->fa_blen.p, n)) || (rc = copy_out(h, where, slen, h->fa_slen.p, n)) ||
        (rc = copy_out(h, where, llen, h->fa_llen.p, n)) || (rc = copy_out(h, where, elen, h->fa_elen.p, n)) ||
        (rc = copy_out(h, where, norm, h->fa_norm.p, n)) || (rc = copy_out(h, where, dlen, h->fa_dlen.p, n)) ||
        (rc = copy_out(h, where, name_len, h->fa_name_len.p, n)))
        return rc;
    int stitch_err = 0;
    HIPCHK(hipMemcpyAsync(&stitch_err, h->ctl.p + 60, sizeof stitch_err, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (stitch_err) return fail(FX_ERANGE, "header line crossing a shard cut continues for more than 64 KiB");
    return FX_OK;
}

// dense composition of the records of this handle into tmp (n_hdr rows + the row of the leading bytes), enqueued
static int fasta_comp_dense(fx_handle *h, int64_t lead_from, DevBuf<unsigned long long> &tmp, DevBuf<int32_t> &edge) {
    if (!h->fasta_built) return fail(FX_ESTATE, "fx_fasta_build has not run");
    int rc = use_device(h);
    if (!rc) rc = finish_build(h);
    if (rc) return rc;
    const int64_t n = h->n_hdr * 128;
    if ((rc = tmp.alloc(n + 128))) return rc;
    unsigned long long *d = tmp.p;
    HIPCHK(hipMemsetAsync(d, 0, (size_t)(n + 128) * 8, h->stream));
    // one wave per run of granules
    const bool from_scan = h->comp_runs_valid;              // the build counted on the way (k_scan_comp): the runs only need an owner
    const int gpw = from_scan ? h->comp_gpw : comp_run_granules(h->ngran);
    const int64_t waves = (h->ngran + gpw - 1) / gpw;
    const dim3 grid((unsigned)((waves + COMP_WPB - 1) / COMP_WPB));
    if ((rc = edge.alloc(waves + 1))) return rc;            // [0]: number of runs left to the second launch, [1..]: their ids
    HIPCHK(hipMemsetAsync(edge.p, 0, 4, h->stream));
    if (from_scan)
        FX_LAUNCH(h, K_COMP_ATTRIBUTE, k_comp_attribute, dim3(nblocks(waves, ATTR_BLOCK)), dim3(ATTR_BLOCK), h->comp_runs.p, waves, gpw, h->n,
                  h->base, h->fa_boff.p, h->n_hdr, h->hdr_prefix.p, h->ngran, lead_from, edge.p, d);
    else
        FX_LAUNCH(h, K_FASTA_COMP, k_fasta_comp<true>, grid, dim3(COMP_WPB * 64), h->d_data, h->n, h->base,
                       h->hdr.p, h->fa_boff.p, h->n_hdr, h->hdr_prefix.p, h->ngran, gpw, edge.p, lead_from, d);
    FX_LAUNCH(h, K_FASTA_COMP_EDGE, k_fasta_comp<false>, grid, dim3(COMP_WPB * 64), h->d_data, h->n, h->base,
                       h->hdr.p, h->fa_boff.p, h->n_hdr, h->hdr_prefix.p, h->ngran, gpw, edge.p, lead_from, d);
    if (h->n_hdr > 0) {                                      // short records: one 16-lane group each
        static int per_cu = 0, n_cu = 256;                  // asked once
        if (!per_cu) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fasta_comp_small, BLOCK, 0) != hipSuccess || per_cu <= 0) per_cu = 4;
            (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device);
        }
        const unsigned nb = (unsigned)std::min<int64_t>(nblocks(h->n_hdr, (BLOCK / 64) * 4), (int64_t)per_cu * n_cu);
        FX_LAUNCH(h, K_FASTA_COMP_SMALL, k_fasta_comp_small, dim3(nb), dim3(BLOCK), h->d_data, h->n, h->base, h->hdr.p, h->fa_boff.p, h->n_hdr, d);
    }
    HIPCHK(hipGetLastError());
    return FX_OK;
}

// comp: n_hdr x 128 (where = FX_HOST / FX_DEVICE); lead (host, 128 words or null): counts of the bytes before the
// shard's first header line from global offset lead_from on (lead_from < 0: not counted)
static int fasta_comp_impl(fx_handle *h, int where, int64_t *comp, int64_t lead_from, int64_t *lead) {
    if (!h || !comp) return fail(FX_EINVAL, "null argument");
    DevBuf<unsigned long long> tmp;
    DevBuf<int32_t> edge;
    int rc = fasta_comp_dense(h, lead_from, tmp, edge);
    if (rc) return rc;
    const int64_t n = h->n_hdr * 128;
    if (n) HIPCHK(hipMemcpyAsync(comp, tmp.p, (size_t)n * 8, where == FX_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
    if (lead) HIPCHK(hipMemcpyAsync(lead, tmp.p + n, 128 * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return FX_OK;
}

extern "C" int fx_fasta_comp_sparse(fx_handle *h, int where, int64_t cap, int64_t *seqid, int64_t *abc, int64_t *num,
                                    int64_t *n_out, int64_t *total) {
    if (!h || !n_out || !total || cap < 0 || (cap > 0 && (!seqid || !abc || !num))) return fail(FX_EINVAL, "bad argument");
    DevBuf<unsigned long long> dense, tot;
    DevBuf<int32_t> edge, cnt;
    DevBuf<int64_t> sums, off, out;
    int rc = fasta_comp_dense(h, -1, dense, edge);
    if (rc) return rc;
    const int64_t n = h->n_hdr, nchunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if ((rc = tot.alloc(128)) || (rc = cnt.alloc(std::max<int64_t>(n, 1))) || (rc = sums.alloc(nchunks + 1)) || (rc = off.alloc(n + 1))) return rc;
    HIPCHK(hipMemsetAsync(tot.p, 0, 128 * 8, h->stream));
    HIPCHK(hipMemsetAsync(off.p, 0, (size_t)(n + 1) * 8, h->stream));
    if (n) {
        const unsigned nb = (unsigned)std::min<int64_t>(nblocks(n, BLOCK / 64), 2048);
        hipLaunchKernelGGL(k_comp_count, dim3(nb), dim3(BLOCK), 0, h->stream, dense.p, n, cnt.p, tot.p);
        hipLaunchKernelGGL(k_cnt_chunk_sums, dim3((unsigned)nchunks), dim3(BLOCK), 0, h->stream, cnt.p, n, sums.p);
        hipLaunchKernelGGL(k_cnt_chunk_bases, dim3(1), dim3(BLOCK), 0, h->stream, sums.p, nchunks);
        hipLaunchKernelGGL(k_cnt_offsets, dim3((unsigned)nchunks), dim3(BLOCK), 0, h->stream, cnt.p, n, sums.p, off.p);
        HIPCHK(hipGetLastError());
    }
    int64_t count = 0;
    HIPCHK(hipMemcpyAsync(&count, off.p + n, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(total, tot.p, 128 * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *n_out = count;
    if (count > cap) return fail(FX_ERANGE, "%lld triples, room for %lld", (long long)count, (long long)cap);
    if (count == 0) return FX_OK;
    int64_t *ds = seqid, *da = abc, *dn = num;
    if (where != FX_DEVICE) {
        if ((rc = out.alloc(3 * count))) return rc;
        ds = out.p; da = out.p + count; dn = out.p + 2 * count;
    }
    const unsigned nb = (unsigned)std::min<int64_t>(nblocks(n, BLOCK / 64), 2048);
    hipLaunchKernelGGL(k_comp_emit, dim3(nb), dim3(BLOCK), 0, h->stream, dense.p, n, off.p, ds, da, dn);
    HIPCHK(hipGetLastError());
    if (where != FX_DEVICE) {
        if ((rc = to_host(h, seqid, ds, count * 8)) || (rc = to_host(h, abc, da, count * 8)) || (rc = to_host(h, num, dn, count * 8))) return rc;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return FX_OK;
}

extern "C" int fx_fasta_comp(fx_handle *h, int where, int64_t *comp) { return fasta_comp_impl(h, where, comp, -1, nullptr); }

extern "C" int fx_fasta_comp_shard(fx_handle *h, int where, int64_t *comp, int64_t lead_from, int64_t *lead) {
    if (!lead) return fail(FX_EINVAL, "null argument");
    return fasta_comp_impl(h, where, comp, lead_from, lead);
}

// ------------------------------------------------------------- FASTQ build
// Count pass (fx_fastq.hpp): granule newline counts + prefixes, and the newlines of the shard's core
// (what the next shards need to number their lines).
static int fastq_count(fx_handle *h, int64_t *n_nl_core, int64_t *last_nl_core, bool want_comp = false) {
    int rc = use_device(h);
    if (rc) return rc;
    if (h->n <= 0) return fail(FX_EFORMAT, "empty input");
    h->fasta_built = h->fastq_built = h->comp_runs_valid = false;
    h->fq_comp_valid = false;
    // One read or two?  Line records pay when most granules fit their slot: ask three windows of the stream.
    static const int force = [] { const char *e = getenv("FX_FQ_LINES"); return e ? atoi(e) : -1; }();   // 0 / 1: experiments
    bool by_lines = force > 0;
    h->fq_crlf = false;
    memset(g_build_laps, 0, sizeof g_build_laps);
    LapClock lc;
    if (h->n >= 4 * GRAN) {
        if ((rc = h->ctl.alloc(64))) return rc;
        HIPCHK(hipMemsetAsync(h->ctl.p + 56, 0, 3 * sizeof(unsigned long long), h->stream));
        hipLaunchKernelGGL(k_nl_sample, dim3(3), dim3(BLOCK), 0, h->stream, h->d_data, h->n, h->ctl.p + 56);
        HIPCHK(hipGetLastError());
        unsigned long long smp[3];
        HIPCHK(hipMemcpyAsync(smp, h->ctl.p + 56, sizeof smp, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (force < 0) by_lines = smp[1] > 0 && (double)smp[0] / (double)smp[1] * GRAN <= 0.7 * FQL_CAP;
        h->fq_crlf = smp[2] > 0;                              // the sampled windows hold "\r\n": the stream kernels take their CRLF form
    }
    // the composition on the way (fx_fastq_build_comp): whole streams with line records only -- a shard counts the reads it OWNS
    static const bool no_fuse = [] { const char *e = getenv("FX_FQ_NO_FUSED_COMP"); return e && atoi(e) != 0; }();
    const bool with_comp = want_comp && by_lines && !no_fuse && h->base == 0 && h->halo == 0 && h->is_last && h->n >= GRAN;
    lc.lap(0);
    if ((rc = granule_pass<1>(h, by_lines, with_comp))) return rc;
    lc.lap(1);
    int64_t *res = (int64_t *)(h->ctl.p + 48);                // 3 words of the control block
    hipLaunchKernelGGL(k_core_count, dim3(1), dim3(64), 0, h->stream, scan_ctx(h), (int)h->is_last, h->n - h->halo, res);
    HIPCHK(hipGetLastError());
    int64_t host[3];
    HIPCHK(hipMemcpyAsync(h->pin_tot, ctl_totals(h), sizeof(Totals), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(host, res, sizeof host, hipMemcpyDeviceToHost, h->stream));
    uint32_t n_over = 0;
    if (by_lines) HIPCHK(hipMemcpyAsync(&n_over, ctl_counter(h, 0), sizeof n_over, hipMemcpyDeviceToHost, h->stream));
    FastqAcc built;
    memset(&built, 0, sizeof built);
    uint32_t n_rej = 0;
    if (with_comp) {
        HIPCHK(hipMemcpyAsync(&built, h->fq_acc_build.p, sizeof built, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(&n_rej, h->fq_rej.p, 4, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    lc.lap(2);
    if (with_comp) {
        const int64_t nfull = h->n / GRAN, nruns = (nfull + FQLC_G - 1) / FQLC_G;
        h->fq_comp_runs = nruns;
        h->fq_comp_rejected = n_rej;
        // runs whose guess was wrong or missing (no '+' line in the run's first granule, '+' lines that repeat the name, ...): counted
        // again, with the line numbers the prefixes give, when they are few -- else the table kernel counts everything (fx_fastq_comp)
        static const int64_t rej_div = [] { const char *e = getenv("FX_FQ_RECOUNT_DIV"); return e && atoll(e) > 0 ? atoll(e) : 8ll; }();
        if (built.qfix == 0 && n_rej > 0) {
            if ((int64_t)n_rej <= std::max<int64_t>(64, nruns / rej_div)) {
                const int64_t pieces = (int64_t)n_rej * (FQLC_G / FS_GPW);
                FX_LAUNCH(h, K_FASTQ_COMP, k_fastq_comp_stream, dim3((unsigned)std::min<int64_t>(nblocks(pieces, BLOCK / 64), 2048)), dim3(BLOCK), h->d_data, h->n, nfull,
                          (const int64_t *)h->nl_prefix.p, (int64_t)0, h->fq_acc_build.p, (const uint32_t *)(h->fq_rej.p + 1), (int64_t)n_rej, (int)FQLC_G);
                HIPCHK(hipGetLastError());
                HIPCHK(hipMemcpyAsync(&built, h->fq_acc_build.p, sizeof built, hipMemcpyDeviceToHost, h->stream));
                HIPCHK(hipStreamSynchronize(h->stream));
            } else {
                built.qfix = 1;
                h->fq_comp_rejected = -1;
            }
        }
    }
    if (with_comp && built.qfix == 0) {                       // every run counted (by its guess or from the prefixes), no byte the stream form leaves to the table kernel
        h->fq_comp_valid = true;
        h->fq_comp_base[0] = (int64_t)built.a; h->fq_comp_base[1] = (int64_t)built.c; h->fq_comp_base[2] = (int64_t)built.g;
        h->fq_comp_base[3] = (int64_t)built.t; h->fq_comp_base[4] = (int64_t)built.n;
        h->fq_comp_minqs = built.minqs; h->fq_comp_maxqs = built.maxqs;
    }
    h->fq_by_lines = by_lines;
    if (by_lines) {                                           // the partial last granule goes on the list as well
        const uint32_t tail = (uint32_t)(h->ngran - 1);
        HIPCHK(hipMemcpyAsync(h->hdr_grans.p + n_over, &tail, sizeof tail, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->fq_nlist = (int64_t)n_over + 1;
    }
    h->n_nl = h->pin_tot->n_nl;
    h->fq_c2 = host[2];
    h->scanned = true;
    if (n_nl_core) *n_nl_core = host[0];
    if (last_nl_core) *last_nl_core = host[1];
    return FX_OK;
}

// Emit pass: the read table, given where this shard's lines sit in the global numbering
// (loff = newlines in earlier shards' cores, prev_nl = offset of the last of them).
struct FqPlan { FqOwn own; int64_t n_reads, n_seq; };
// Ownership: record k's header line starts right after global newline 4k-1 (k = 0: at offset 0); the
// shard owns the records whose header line STARTS in [base, core_end).
static int fastq_plan(fx_handle *h, int64_t loff, int64_t prev_nl, FqPlan *pl) {
    const int64_t N = h->n_nl;                                 // shard newlines (virtual end-of-stream one included)
    const int64_t k0 = (loff + 3) / 4;
    const int64_t k_first = k0 + ((loff % 4 == 0 && prev_nl + 1 < h->base) ? 1 : 0);   // that record's header began in the previous shard
    const int64_t core_len = h->n - h->halo;
    const int64_t k_end = core_len > 0 ? (loff + h->fq_c2) / 4 + 1 : k_first;           // exclusive
    const int64_t complete = (loff + N) / 4;                   // records whose four lines end inside what we hold
    const int64_t nrows = std::max<int64_t>(k_end - k_first, 0);
    pl->n_reads = std::max<int64_t>(std::min(k_end, complete) - k_first, 0);
    // rows that have at least their sequence line (an incomplete trailing record still counts in stat.size, fastq.c:125)
    pl->n_seq = loff + N >= 2 ? std::max<int64_t>(std::min(k_end, (loff + N - 2) / 4 + 1) - k_first, 0) : 0;
    if (!h->is_last && k_end > complete)
        return fail(FX_ERANGE, "a FASTQ record that starts in this shard runs past its %lld-byte halo", (long long)h->halo);
    pl->own = FqOwn{loff, prev_nl, k_first, nrows};
    return FX_OK;
}
static int fastq_alloc(fx_handle *h, int64_t cap) {
    int rc;
    cap = std::max<int64_t>(cap, 1);
    if ((rc = h->fq_name_off.alloc(cap)) || (rc = h->fq_rlen.alloc(cap)) || (rc = h->fq_soff.alloc(cap)) ||
        (rc = h->fq_qoff.alloc(cap)) || (rc = h->fq_name_len.alloc(cap)) || (rc = h->fq_dlen.alloc(cap)) ||
        (rc = h->fq_qlen.alloc(cap)) || (rc = h->fq_acc.alloc(1)))
        return rc;
    return FX_OK;
}
static FqTab fastq_tab(fx_handle *h) {
    return FqTab{h->fq_name_off.p, h->fq_rlen.p, h->fq_soff.p, h->fq_qoff.p, h->fq_name_len.p, h->fq_dlen.p, h->fq_qlen.p};
}
// size / maxlen / minlen over the table, the handle's FASTQ state, the summary
static int fastq_finish(fx_handle *h, const FqPlan &pl, fx_fastq_summary *out) {
    FastqAcc init;
    memset(&init, 0, sizeof init);
    init.maxlen = 0; init.minlen = 10000000000LL; init.minqs = 104; init.maxqs = 33;   // fastq.c:667-675
    HIPCHK(hipMemcpyAsync(h->fq_acc.p, &init, sizeof init, hipMemcpyHostToDevice, h->stream));
    const FqTab t = fastq_tab(h);
    FX_LAUNCH(h, K_FASTQ_STATS, k_fastq_stats, dim3((unsigned)std::min<int64_t>(nblocks(pl.n_seq, BLOCK), 1024)), dim3(BLOCK), t,
              pl.n_seq, pl.n_reads, h->fq_acc.p);
    HIPCHK(hipGetLastError());
    FastqAcc acc;
    HIPCHK(hipMemcpyAsync(&acc, h->fq_acc.p, sizeof acc, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->n_reads = pl.n_reads;
    h->fq_seq_rows = pl.n_seq;
    h->fq_size = (int64_t)acc.size;
    h->fq_maxlen = acc.maxlen; h->fq_minlen = acc.minlen;
    h->fastq_built = true;
    if (out) { out->n_reads = h->n_reads; out->size = h->fq_size; out->n_lines = h->n_nl; out->n_bytes = h->n; out->first_id = pl.own.k_first; }
    return FX_OK;
}

// Emit pass: the read table, given where this shard's lines sit in the global numbering
// (loff = newlines in earlier shards' cores, prev_nl = offset of the last of them).
static int fastq_records(fx_handle *h, int64_t loff, int64_t prev_nl, fx_fastq_summary *out) {
    int rc = use_device(h);
    if (rc) return rc;
    FqPlan pl;
    LapClock lc;
    if ((rc = fastq_plan(h, loff, prev_nl, &pl)) || (rc = fastq_alloc(h, pl.own.nrows))) return rc;
    lc.lap(4);
    struct AtExit { LapClock &c; ~AtExit() { c.lap(5); } } at_exit{lc};
    if (h->fq_by_lines) {
        static const bool rows_wave = [] { const char *e = getenv("FX_FQ_ROWS_WAVE"); return e && atoi(e) != 0; }();   // the round-4 form, for comparison
        if (h->ngran > 1 && rows_wave)
            FX_LAUNCH(h, K_FASTQ_ROWS, k_fastq_rows, dim3(nblocks(h->ngran - 1, (BLOCK / 64) * FQR_G)), dim3(BLOCK), scan_ctx(h), pl.own, fastq_tab(h),
                      h->fq_lines.p, h->ngran - 1);
        else if (h->ngran > 1) {
            // granules per workgroup: as many as leave the lines of a workgroup under FQW_CAP staged records with room to spare
            const double lpg = (double)h->n_nl / (double)h->ngran;
            static const bool nt = [] { const char *e = getenv("FX_FQ_ROWS_NT"); return e && atoi(e) != 0; }();
#define FX_ROWS_WG(G, NT) FX_LAUNCH(h, K_FASTQ_ROWS, (k_fastq_rows_wg<G, NT>), dim3(nblocks(h->ngran - 1, G * FX_FQW_GROUPS)), dim3(BLOCK), scan_ctx(h), pl.own, fastq_tab(h), h->fq_lines.p, h->ngran - 1)
            if (lpg * 64 <= 0.75 * FQW_CAP) { if (nt) FX_ROWS_WG(64, true); else FX_ROWS_WG(64, false); }
            else if (lpg * 32 <= 0.75 * FQW_CAP) { if (nt) FX_ROWS_WG(32, true); else FX_ROWS_WG(32, false); }
            else { if (nt) FX_ROWS_WG(16, true); else FX_ROWS_WG(16, false); }
#undef FX_ROWS_WG
        }
        FX_LAUNCH(h, K_FASTQ_EMIT, k_fastq_emit, dim3(nblocks(h->fq_nlist, BLOCK / 64)), dim3(BLOCK), scan_ctx(h), h->prev_byte,
                  (int)h->is_last, pl.own, fastq_tab(h), (const uint32_t *)h->hdr_grans.p, h->fq_nlist);
    } else {
        FX_LAUNCH(h, K_FASTQ_EMIT, k_fastq_emit, dim3(nblocks(h->ngran, BLOCK / 64)), dim3(BLOCK), scan_ctx(h), h->prev_byte,
                  (int)h->is_last, pl.own, fastq_tab(h), (const uint32_t *)nullptr, (int64_t)0);
    }
    return fastq_finish(h, pl, out);
}

extern "C" int fx_set_halo(fx_handle *h, int64_t halo_bytes) {
    if (!h || halo_bytes < 0 || halo_bytes > h->n) return fail(FX_EINVAL, "bad halo");
    h->halo = halo_bytes;
    h->scanned = h->fasta_built = h->fastq_built = h->comp_runs_valid = false;
    return FX_OK;
}

extern "C" int fx_fastq_scan(fx_handle *h, int64_t *n_nl_core, int64_t *last_nl_core) {
    if (!h) return fail(FX_EINVAL, "null handle");
    return fastq_count(h, n_nl_core, last_nl_core);
}

extern "C" int fx_fastq_build_ctx(fx_handle *h, int64_t line_offset, int64_t prev_nl, fx_fastq_summary *out) {
    if (!h) return fail(FX_EINVAL, "null handle");
    if (!h->scanned) return fail(FX_ESTATE, "fx_fastq_scan has not run");
    return fastq_records(h, line_offset, prev_nl, out);
}

extern "C" int fx_fastq_build(fx_handle *h, fx_fastq_summary *out) {
    if (!h) return fail(FX_EINVAL, "null handle");
    int rc = fastq_count(h, nullptr, nullptr);
    if (rc) return rc;
    return fastq_records(h, 0, -1, out);
}

extern "C" int fx_fastq_build_comp(fx_handle *h, fx_fastq_summary *out) {
    if (!h) return fail(FX_EINVAL, "null handle");
    int rc = fastq_count(h, nullptr, nullptr, true);
    if (rc) return rc;
    return fastq_records(h, 0, -1, out);
}

extern "C" int fx_fastq_comp_info(fx_handle *h, int64_t *runs, int64_t *recounted, int *one_read) {
    if (!h) return fail(FX_EINVAL, "null handle");
    if (runs) *runs = h->fq_comp_runs;
    if (recounted) *recounted = h->fq_comp_rejected;
    if (one_read) *one_read = h->fq_comp_valid ? 1 : 0;
    return FX_OK;
}

extern "C" int fx_fastq_table(fx_handle *h, int where, int64_t *name_off, int32_t *name_len, int32_t *dlen,
                              int64_t *rlen, int64_t *soff, int64_t *qoff) {
    if (!h) return fail(FX_EINVAL, "null handle");
    if (!h->fastq_built) return fail(FX_ESTATE, "fx_fastq_build has not run");
    int rc = use_device(h);
    if (rc) return rc;
    const int64_t n = h->n_reads;
    if ((rc = copy_out(h, where, name_off, h->fq_name_off.p, n)) || (rc = copy_out(h, where, name_len, h->fq_name_len.p, n)) ||
        (rc = copy_out(h, where, dlen, h->fq_dlen.p, n)) || (rc = copy_out(h, where, rlen, h->fq_rlen.p, n)) ||
        (rc = copy_out(h, where, soff, h->fq_soff.p, n)) || (rc = copy_out(h, where, qoff, h->fq_qoff.p, n)))
        return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    return FX_OK;
}

extern "C" int fx_fastq_comp(fx_handle *h, int64_t base[5], int64_t meta[5]) {
    if (!h || !base || !meta) return fail(FX_EINVAL, "null argument");
    if (!h->fastq_built) return fail(FX_ESTATE, "fx_fastq_build has not run");
    int rc = use_device(h);
    if (rc) return rc;
    if (h->fq_comp_valid) {                                  // counted while the index was built (fx_fastq_build_comp): nothing is read again
        for (int i = 0; i < 5; ++i) base[i] = h->fq_comp_base[i];
        int phred = 0;
        if (h->fq_comp_maxqs > 74) phred = 64;                // fastq.c:768-774
        if (h->fq_comp_minqs < 59) phred = 33;
        meta[0] = h->fq_maxlen; meta[1] = h->fq_minlen; meta[2] = h->fq_comp_minqs; meta[3] = h->fq_comp_maxqs; meta[4] = phred;
        return FX_OK;
    }
    const FqTab t{h->fq_name_off.p, h->fq_rlen.p, h->fq_soff.p, h->fq_qoff.p, h->fq_name_len.p, h->fq_dlen.p, h->fq_qlen.p};
    // FX_FQ_COMP_STREAM=1: the composition as a stream over the bytes (k_fastq_comp_stream, fx_fastq_stream.hpp: coalesced loads,
    // the line of four of every byte from the build's newline prefixes).  Correct on everything the tests hold, and measured in
    // round 4: 7.6-7.8 ms for C3 against 7.2 for the gather from the read table below -- the stream form is bound by its
    // ~490 VALU instructions per granule (the exact newline mask and the wave scans alone are 150 of them), so the table
    // kernel stays the default.  The stream kernel only flags what it does not do itself (a '\r' or a byte outside '!'..127 in a
    // quality line): then, and for shards (which count the reads they OWN), the table kernel runs.
    static const bool table_only = [] { const char *e = getenv("FX_FQ_COMP_STREAM"); return !(e && atoi(e) != 0); }();
    if (!table_only && h->base == 0 && h->halo == 0 && h->is_last && h->nl_prefix.p && h->ngran > 0) {
        const int64_t nfull = h->n / GRAN, waves = nfull / FS_GPW + 1;
        static int s_per_cu = 0, s_n_cu = 256;               // as many workgroups as are resident at once
        if (!s_per_cu) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&s_per_cu, k_fastq_comp_stream, BLOCK, 0) != hipSuccess || s_per_cu <= 0) s_per_cu = 4;
            (void)hipDeviceGetAttribute(&s_n_cu, hipDeviceAttributeMultiprocessorCount, h->device);
        }
        FX_LAUNCH(h, K_FASTQ_COMP, k_fastq_comp_stream, dim3((unsigned)std::min<int64_t>(nblocks(waves, BLOCK / 64), (int64_t)s_per_cu * s_n_cu)), dim3(BLOCK), h->d_data, h->n, nfull,
                  (const int64_t *)h->nl_prefix.p, (int64_t)0, h->fq_acc.p, (const uint32_t *)nullptr, (int64_t)0, 0);
        HIPCHK(hipGetLastError());
        FastqAcc acc;
        HIPCHK(hipMemcpyAsync(&acc, h->fq_acc.p, sizeof acc, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        FastqAcc keep = acc;                                 // the counters back to their start values, whichever way this goes on
        keep.a = keep.c = keep.g = keep.t = keep.n = 0; keep.minqs = 104; keep.maxqs = 33; keep.qfix = 0;
        HIPCHK(hipMemcpyAsync(h->fq_acc.p, &keep, sizeof keep, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (!acc.qfix) {
            base[0] = (int64_t)acc.a; base[1] = (int64_t)acc.c; base[2] = (int64_t)acc.g; base[3] = (int64_t)acc.t; base[4] = (int64_t)acc.n;
            int phred = 0;
            if (acc.maxqs > 74) phred = 64;                  // fastq.c:768-774
            if (acc.minqs < 59) phred = 33;
            meta[0] = h->fq_maxlen; meta[1] = h->fq_minlen; meta[2] = acc.minqs; meta[3] = acc.maxqs; meta[4] = phred;
            return FX_OK;
        }
    }
    // a grid-stride kernel: exactly as many workgroups as are resident at once (no partial last round)
    static int per_cu = 0, n_cu = 256;                      // asked once
    if (!per_cu) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fastq_comp, BLOCK, 0) != hipSuccess || per_cu <= 0) per_cu = 3;
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device);
    }
    // lanes per record: one 16-byte piece per lane covers the mean read (FX_FQ_LPR: experiments)
    static const int lpr_force = [] { const char *e = getenv("FX_FQ_LPR"); return e ? atoi(e) : 0; }();
    const int64_t mean_len = h->fq_seq_rows > 0 ? (h->fq_size + h->fq_seq_rows - 1) / h->fq_seq_rows : 1;
    const int lpr = (int)std::clamp<int64_t>(lpr_force > 0 ? lpr_force : (mean_len + 15) / 16, 1, 64);
    const unsigned nb = (unsigned)std::min<int64_t>(nblocks(std::max<int64_t>(h->fq_seq_rows, 1), (BLOCK / 64) * (64 / lpr) * FX_FQ_U), (int64_t)per_cu * n_cu);
    FX_LAUNCH(h, K_FASTQ_COMP, k_fastq_comp, dim3(nb), dim3(BLOCK), h->d_data, h->base, h->n, t, h->fq_seq_rows, h->n_reads, h->fq_acc.p, lpr,
              h->n >= 16 ? h->d_data : (const uint8_t *)h->fq_acc.p);      // 16 bytes that can always be read
    HIPCHK(hipGetLastError());
    FastqAcc acc;
    HIPCHK(hipMemcpyAsync(&acc, h->fq_acc.p, sizeof acc, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (acc.qfix) {
        // a '\r' INSIDE a quality line somewhere (fastq.c:733-737 then shrinks line.l as it goes): the quality half once more,
        // line by line exactly as the reference's loop runs, the rows' qlen rewritten, meta.maxlen / minlen from them
        FastqAcc again = acc;
        again.minqs = 104; again.maxqs = 33; again.maxlen = 0; again.minlen = 10000000000LL; again.qfix = 0;
        HIPCHK(hipMemcpyAsync(h->fq_acc.p, &again, sizeof again, hipMemcpyHostToDevice, h->stream));
        const unsigned g = (unsigned)std::min<int64_t>(nblocks(h->n_reads, BLOCK), 4096);
        hipLaunchKernelGGL(k_fastq_qual_walk, dim3(g), dim3(BLOCK), 0, h->stream, h->d_data, h->base, h->n, t, h->n_reads, h->fq_acc.p);
        hipLaunchKernelGGL(k_fastq_qlen_range, dim3((unsigned)std::min<int64_t>(nblocks(h->n_reads, BLOCK), 1024)), dim3(BLOCK), 0, h->stream, t, h->n_reads, h->fq_acc.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&again, h->fq_acc.p, sizeof again, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        acc.minqs = again.minqs; acc.maxqs = again.maxqs;
        if (h->n_reads > 0) { h->fq_maxlen = again.maxlen; h->fq_minlen = again.minlen; }
    }
    // reset the counters so a second call does not double count
    FastqAcc keep = acc;
    keep.a = keep.c = keep.g = keep.t = keep.n = 0; keep.minqs = 104; keep.maxqs = 33; keep.qfix = 0;
    keep.maxlen = h->fq_maxlen; keep.minlen = h->fq_minlen;
    HIPCHK(hipMemcpyAsync(h->fq_acc.p, &keep, sizeof keep, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    base[0] = (int64_t)acc.a; base[1] = (int64_t)acc.c; base[2] = (int64_t)acc.g; base[3] = (int64_t)acc.t; base[4] = (int64_t)acc.n;
    int phred = 0;
    if (acc.maxqs > 74) phred = 64;                        // fastq.c:768-774
    if (acc.minqs < 59) phred = 33;
    meta[0] = h->fq_maxlen; meta[1] = h->fq_minlen; meta[2] = acc.minqs; meta[3] = acc.maxqs; meta[4] = phred;
    return FX_OK;
}


// ------------------------------------------------------------------- Fastx: the kseq walk (fx_kseq.hpp)
extern "C" int fx_kseq_scan(fx_handle *h, int64_t *n_records, int64_t *n_lines, int64_t *seq_bytes, int *end_code) {
    if (!h) return fail(FX_EINVAL, "null handle");
    if (h->base != 0 || h->halo != 0) return fail(FX_ESTATE, "fx_kseq_scan works on a whole stream, not on a shard");
    int rc = use_device(h);
    if (rc) return rc;
    h->kq_nrec = -1;
    const int64_t n = h->n;
    KqOut res{0, -1, 0, 0};
    int64_t L = 0;
    if (n > 0) {
        const int64_t ntiles = (n + KQ_TILE - 1) / KQ_TILE, nchunks = (ntiles + SCAN_CHUNK - 1) / SCAN_CHUNK;
        ScratchBuf<int32_t> cnt;
        ScratchBuf<int64_t> sums, off;
        ScratchBuf<unsigned long long> ctl;                  // [0] '>' and '@' bytes, [1] error bits, [2..5] KqOut, [6..7] k_kq_classify, [8..15] KqInit
        if ((rc = cnt.alloc(h->device, ntiles, h->stream)) || (rc = sums.alloc(h->device, nchunks + 1, h->stream)) ||
            (rc = off.alloc(h->device, ntiles + 1, h->stream)) || (rc = ctl.alloc(h->device, 16, h->stream)))
            return rc;
        HIPCHK(hipMemsetAsync(ctl.p, 0, 16 * sizeof(unsigned long long), h->stream));
        const unsigned wide = (unsigned)std::min<int64_t>(nblocks(ntiles, BLOCK / 64), 256 * 8);
        // exclusive prefix sums of a column of counts (fx_comp.hpp: three small kernels)
        auto scan = [&](const int32_t *c, int64_t m, int64_t *sums_p, int64_t *o) {
            const int64_t nch = (m + SCAN_CHUNK - 1) / SCAN_CHUNK;
            hipLaunchKernelGGL(k_cnt_chunk_sums, dim3((unsigned)nch), dim3(BLOCK), 0, h->stream, c, m, sums_p);
            hipLaunchKernelGGL(k_cnt_chunk_bases, dim3(1), dim3(BLOCK), 0, h->stream, sums_p, nch);
            hipLaunchKernelGGL(k_cnt_offsets, dim3((unsigned)nch), dim3(BLOCK), 0, h->stream, c, m, sums_p, o);
        };
        FX_LAUNCH(h, K_KQ_LINES, k_kq_count, dim3(wide), dim3(BLOCK), h->d_data, n, ntiles, cnt.p, ctl.p);
        scan(cnt.p, ntiles, sums.p, off.p);
        HIPCHK(hipGetLastError());
        int64_t n_nl = 0;
        unsigned long long hdrchars = 0;
        uint8_t tail = 0;
        HIPCHK(hipMemcpyAsync(&n_nl, off.p + ntiles, sizeof n_nl, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(&hdrchars, ctl.p, sizeof hdrchars, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(&tail, h->d_data + (n - 1), 1, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        const bool virt = tail != '\n';                      // the last line has no '\n': it ends where the stream ends
        L = n_nl + (virt ? 1 : 0);
        const int64_t cap = std::min<int64_t>(L, (int64_t)hdrchars) + 1;
        ScratchBuf<uint4> desc;
        if ((rc = h->kq_nl.alloc(L)) || (rc = h->kq_ldst.alloc(L)) || (rc = h->kq_lcon.alloc(L)) || (rc = h->kq_recs.alloc(cap)) ||
            (rc = desc.alloc(h->device, L, h->stream)))
            return rc;
        HIPCHK(hipMemsetAsync(h->kq_ldst.p, 0, (size_t)L * sizeof(int64_t), h->stream));
        HIPCHK(hipMemsetAsync(h->kq_lcon.p, 0, (size_t)L * sizeof(uint32_t), h->stream));
        const unsigned per_line = (unsigned)std::min<int64_t>(nblocks(L, BLOCK), 256 * 16);
        FX_LAUNCH(h, K_KQ_LINES, k_kq_lines, dim3(wide), dim3(BLOCK), h->d_data, n, ntiles, off.p, h->kq_nl.p, virt ? L - 1 : (int64_t)-1);
        hipLaunchKernelGGL(k_kq_desc, dim3(per_line), dim3(BLOCK), 0, h->stream, h->d_data, n, h->kq_nl.p, L, desc.p, (uint32_t *)(ctl.p + 1));
        // ---- the regular prefix in parallel (FX_KSEQ_WALK_ONLY=1: everything through the walk -- tests, experiments)
        static const bool walk_only = [] { const char *e = getenv("FX_KSEQ_WALK_ONLY"); return e && atoi(e) != 0; }();
        KqInit *init = (KqInit *)(ctl.p + 8);
        h->kq_prefix_lines = 0;
        if (!walk_only) {
            const unsigned long long preset[2] = {(unsigned long long)L, (unsigned long long)L};
            unsigned long long first[2];
            uint4 d0;
            HIPCHK(hipMemcpyAsync(ctl.p + 6, preset, sizeof preset, hipMemcpyHostToDevice, h->stream));
            FX_LAUNCH(h, K_KQ_PREFIX, k_kq_classify, dim3(per_line), dim3(BLOCK), desc.p, L, ctl.p + 6);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(first, ctl.p + 6, sizeof first, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipMemcpyAsync(&d0, desc.p, sizeof d0, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            const int64_t R = (int64_t)first[0] / 4, na = (int64_t)first[1];
            const bool hdr0 = d0.z >= 1 && ((d0.w & 0xFF) == '>' || (d0.w & 0xFF) == '@');
            if (R >= 1) {                                     // four-line FASTQ records up to line 4 R
                ScratchBuf<int32_t> c;
                ScratchBuf<int64_t> sm, o;
                if ((rc = c.alloc(h->device, R, h->stream)) || (rc = sm.alloc(h->device, R / SCAN_CHUNK + 2, h->stream)) || (rc = o.alloc(h->device, R + 1, h->stream))) return rc;
                const unsigned g = (unsigned)std::min<int64_t>(nblocks(R, BLOCK), 256 * 16);
                h->prof.begin(K_KQ_PREFIX, h->stream);
                hipLaunchKernelGGL(k_kq_fq_cnt, dim3(g), dim3(BLOCK), 0, h->stream, desc.p, R, c.p);
                scan(c.p, R, sm.p, o.p);
                hipLaunchKernelGGL(k_kq_fq_emit, dim3(g), dim3(BLOCK), 0, h->stream, desc.p, R, o.p, h->kq_recs.p, h->kq_ldst.p, h->kq_lcon.p, init);
                h->prof.end(h->stream);
                HIPCHK(hipGetLastError());
                HIPCHK(hipStreamSynchronize(h->stream));      // the scratch goes back to the pool
                h->kq_prefix_lines = 4 * R;
            } else if (hdr0 && na >= 1) {                     // FASTA header / sequence lines up to line na
                ScratchBuf<int32_t> hf, cn;
                ScratchBuf<int64_t> sm, ho, co, hp;
                if ((rc = hf.alloc(h->device, na, h->stream)) || (rc = cn.alloc(h->device, na, h->stream)) || (rc = sm.alloc(h->device, na / SCAN_CHUNK + 2, h->stream)) ||
                    (rc = ho.alloc(h->device, na + 1, h->stream)) || (rc = co.alloc(h->device, na + 1, h->stream)))
                    return rc;
                const unsigned g = (unsigned)std::min<int64_t>(nblocks(na, BLOCK), 256 * 16);
                h->prof.begin(K_KQ_PREFIX, h->stream);
                hipLaunchKernelGGL(k_kq_fa_cnt, dim3(g), dim3(BLOCK), 0, h->stream, desc.p, na, hf.p, cn.p);
                scan(hf.p, na, sm.p, ho.p);
                scan(cn.p, na, sm.p, co.p);
                h->prof.end(h->stream);
                HIPCHK(hipGetLastError());
                int64_t nh = 0;
                HIPCHK(hipMemcpyAsync(&nh, ho.p + na, sizeof nh, hipMemcpyDeviceToHost, h->stream));
                HIPCHK(hipStreamSynchronize(h->stream));
                if ((rc = hp.alloc(h->device, nh, h->stream))) return rc;
                h->prof.begin(K_KQ_PREFIX, h->stream);
                hipLaunchKernelGGL(k_kq_fa_lines, dim3(g), dim3(BLOCK), 0, h->stream, desc.p, na, ho.p, co.p, cn.p, hp.p, h->kq_ldst.p, h->kq_lcon.p);
                hipLaunchKernelGGL(k_kq_fa_recs, dim3((unsigned)std::min<int64_t>(nblocks(nh, BLOCK), 256 * 16)), dim3(BLOCK), 0, h->stream, desc.p, na, nh, hp.p,
                                   co.p, h->kq_recs.p, init);
                h->prof.end(h->stream);
                HIPCHK(hipGetLastError());
                HIPCHK(hipStreamSynchronize(h->stream));
                h->kq_prefix_lines = na;
            }
        }
        FX_LAUNCH(h, K_KQ_WALK, k_kq_walk, dim3(1), dim3(KQ_WALK_BLOCK), h->d_data, n, desc.p, L, h->kq_recs.p, cap, h->kq_ldst.p, h->kq_lcon.p,
                  init, (KqOut *)(ctl.p + 2));
        HIPCHK(hipGetLastError());
        unsigned long long back[6];
        HIPCHK(hipMemcpyAsync(back, ctl.p, sizeof back, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (back[1]) return fail(FX_ERANGE, "a line of 4 GiB or more");
        memcpy(&res, back + 2, sizeof res);
        if (res.pad) return fail(FX_EDEVICE, "kseq walk: record table overflow (%lld records, room for %lld)", res.n_rec, (long long)cap);
    }
    h->kq_nrec = res.n_rec; h->kq_lines = L; h->kq_seq_bytes = res.seq_bytes; h->kq_code = (int)res.code;
    if (n_records) *n_records = res.n_rec;
    if (n_lines) *n_lines = L;
    if (seq_bytes) *seq_bytes = res.seq_bytes;
    if (end_code) *end_code = (int)res.code;
    return FX_OK;
}

extern "C" int64_t fx_kseq_prefix_lines(const fx_handle *h) { return h ? h->kq_prefix_lines : 0; }

extern "C" int fx_kseq_records(fx_handle *h, int64_t first, int64_t count, fx_kseq_rec *out) {
    if (!h || (count > 0 && !out)) return fail(FX_EINVAL, "null argument");
    if (h->kq_nrec < 0) return fail(FX_ESTATE, "fx_kseq_scan has not run");
    if (first < 0 || count < 0 || first + count > h->kq_nrec) return fail(FX_ERANGE, "records [%lld, %lld) of %lld", (long long)first, (long long)(first + count), (long long)h->kq_nrec);
    if (!count) return FX_OK;
    int rc = use_device(h);
    if (rc) return rc;
    static_assert(sizeof(fx_kseq_rec) == sizeof(KqRec), "fx_kseq_rec mirrors KqRec");
    HIPCHK(hipMemcpyAsync(out, h->kq_recs.p + first, (size_t)count * sizeof(KqRec), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return FX_OK;
}

extern "C" int fx_kseq_fetch(fx_handle *h, int where, int64_t first, int64_t count, int flags, uint8_t *seq_dst, uint8_t *qual_dst,
                             int64_t *n_bytes) {
    if (!h) return fail(FX_EINVAL, "null handle");
    if (h->kq_nrec < 0) return fail(FX_ESTATE, "fx_kseq_scan has not run");
    if (first < 0 || count < 0 || first + count > h->kq_nrec) return fail(FX_ERANGE, "records [%lld, %lld) of %lld", (long long)first, (long long)(first + count), (long long)h->kq_nrec);
    if (n_bytes) *n_bytes = 0;
    if (!count) return FX_OK;
    int rc = use_device(h);
    if (rc) return rc;
    KqRec a, b;
    HIPCHK(hipMemcpyAsync(&a, h->kq_recs.p + first, sizeof a, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(&b, h->kq_recs.p + first + count - 1, sizeof b, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int64_t total = b.seq_cum + b.seq_len - a.seq_cum;
    if (n_bytes) *n_bytes = total;
    if (total <= 0 || (!seq_dst && !qual_dst)) return FX_OK;
    const int64_t l0 = a.hdr_line, l1 = std::min<int64_t>(b.hdr_line + b.s_n + ((b.flags & KQ_F_FASTQ) ? 1 + (int64_t)b.q_n : 0), h->kq_lines - 1);
    ScratchBuf<uint8_t> d_seq, d_qual;
    uint8_t *ds = seq_dst, *dq = qual_dst;
    if (where == FX_HOST) {
        if (seq_dst) { if ((rc = d_seq.alloc(h->device, total, h->stream))) return rc; ds = d_seq.p; }
        if (qual_dst) { if ((rc = d_qual.alloc(h->device, total, h->stream))) return rc; dq = d_qual.p; }
    }
    // a FASTA-style record has no quality: those stretches of the quality string read as zero bytes
    if (dq) HIPCHK(hipMemsetAsync(dq, 0, (size_t)total, h->stream));
    const uint32_t long_cap = (uint32_t)std::min<int64_t>(2 * (total / KQ_LONG + 2), 1 << 24);
    ScratchBuf<KqLong> longs;
    ScratchBuf<uint32_t> n_long;
    if ((rc = longs.alloc(h->device, long_cap, h->stream)) || (rc = n_long.alloc(h->device, 1, h->stream))) return rc;
    HIPCHK(hipMemsetAsync(n_long.p, 0, sizeof(uint32_t), h->stream));
    const unsigned g = (unsigned)std::min<int64_t>(nblocks(l1 - l0 + 1, BLOCK / 16), 256 * 16);
    FX_LAUNCH(h, K_KQ_GATHER, k_kq_gather, dim3(g), dim3(BLOCK), h->d_data, h->kq_nl.p, h->kq_ldst.p, h->kq_lcon.p, l0, l1, a.seq_cum, ds, dq,
              (flags & FX_UPPER) ? 1 : 0, longs.p, n_long.p, long_cap);
    HIPCHK(hipGetLastError());
    uint32_t nl_host = 0;
    HIPCHK(hipMemcpyAsync(&nl_host, n_long.p, sizeof nl_host, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (nl_host > long_cap) return fail(FX_EDEVICE, "kseq gather: long-line list overflow");
    if (nl_host) {
        hipLaunchKernelGGL(k_kq_gather_long, dim3(256 * 4), dim3(BLOCK), 0, h->stream, h->d_data, longs.p, nl_host, ds, dq, (flags & FX_UPPER) ? 1 : 0);
        HIPCHK(hipGetLastError());
    }
    if (where == FX_HOST) {
        if (seq_dst) HIPCHK(hipMemcpyAsync(seq_dst, ds, (size_t)total, hipMemcpyDeviceToHost, h->stream));
        if (qual_dst) HIPCHK(hipMemcpyAsync(qual_dst, dq, (size_t)total, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    h->prof.drain();
    return FX_OK;
}

// ------------------------------------------------------------------- fetch
// Per-call device mirrors of host query arrays come out of a grow-only arena owned by the
// handle (no hipMalloc/hipFree per call -- they dominate a single 100-byte query otherwise).
// A request that does not fit falls back to hipMalloc for this call and enlarges the arena
// for the next one.
struct Staged {
    fx_handle *h;
    std::vector<void *> spill;
    int64_t want = 0;
    explicit Staged(fx_handle *hh) : h(hh) { h->arena_used = 0; h->pin_in_used = 0; }
    // room for `bytes` of query arrays in the handle's pinned staging buffer (kept from call to call, grown by half more
    // than asked): a pageable source goes through it -- copied by a few threads, then ONE DMA -- instead of through the
    // runtime's own bounce buffer, which moves a pageable 8 MB array at a fraction of the link's rate
    void reserve_pin(int64_t bytes) {
        bytes += 4096;
        if (bytes <= h->pin_in_cap || bytes < (1 << 16)) return;
        (void)hipStreamSynchronize(h->stream);
        if (h->pin_in) (void)hipHostFree(h->pin_in);
        h->pin_in = nullptr; h->pin_in_cap = 0;
        const int64_t cap = bytes + bytes / 2;
        if (hipHostMalloc((void **)&h->pin_in, (size_t)cap, hipHostMallocDefault) == hipSuccess) h->pin_in_cap = cap;
        else h->pin_in = nullptr;
    }
    ~Staged() {
        for (void *p : spill) (void)hipFree(p);
        if (want > h->arena.cap) {                       // only reached after the call's final sync
            (void)hipStreamSynchronize(h->stream);
            (void)h->arena.alloc(want + want / 2);
        }
    }
    void *take(int64_t bytes) {
        bytes = (std::max<int64_t>(bytes, 1) + 255) & ~255ll;
        want += bytes;
        if (h->arena.p && h->arena_used + bytes <= h->arena.cap) {
            void *p = h->arena.p + h->arena_used;
            h->arena_used += bytes;
            return p;
        }
        void *d = nullptr;
        if (hipMalloc(&d, (size_t)bytes) != hipSuccess) return nullptr;
        spill.push_back(d);
        return d;
    }
    template <class T> int up(fx_handle *, const T *src, int64_t n, const T **dst) {
        *dst = nullptr;
        if (!src || n <= 0) return FX_OK;
        void *d = take(n * (int64_t)sizeof(T));
        if (!d) return fail(FX_ENOMEM, "device scratch allocation failed");
        const int64_t bytes = n * (int64_t)sizeof(T), padded = (bytes + 255) & ~255ll;
        const void *from = src;
        if (bytes >= (1 << 16) && h->pin_in && h->pin_in_used + padded <= h->pin_in_cap && !g_hostpool.holds(src, (size_t)bytes)) {
            par_memcpy(h->pin_in + h->pin_in_used, src, (size_t)bytes);
            from = h->pin_in + h->pin_in_used;
            h->pin_in_used += padded;
        }
        hipError_t e = hipMemcpyAsync(d, from, (size_t)bytes, hipMemcpyHostToDevice, h->stream);
        if (e != hipSuccess) return fail(FX_EDEVICE, "H2D: %s", hipGetErrorString(e));
        *dst = (const T *)d;
        return FX_OK;
    }
    template <class T> int scratch(int64_t n, T **dst) {
        void *d = take(n * (int64_t)sizeof(T));
        if (!d) return fail(FX_ENOMEM, "device scratch allocation failed");
        *dst = (T *)d;
        return FX_OK;
    }
};

static unsigned fetch_grid(int64_t nq) {
    // one wave per query, 4 waves per workgroup; a capped grid, grid-stride beyond it (FX_FETCH_WG: tuning override)
    static const int64_t cap = []() { const char *e = getenv("FX_FETCH_WG"); const int64_t v = e ? atoll(e) : 0; return v > 0 ? v : (int64_t)256 * 8 * 4; }();
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>((nq + 3) / 4, cap));
}

// the kernels of one batch, enqueued on the handle's stream: q's arrays and d_dst are device pointers
static int fetch_launch(fx_handle *h, Staged &st, const FetchQ &q, bool by_id, bool longq, int64_t n, int flags, uint8_t *d_dst) {
    int rc;
    FastaTab tab;
    memset(&tab, 0, sizeof tab);
    if (h->fasta_built) {
        tab.boff = h->fa_boff.p; tab.blen = h->fa_blen.p; tab.slen = h->fa_slen.p; tab.llen = h->fa_llen.p;
        tab.elen = h->fa_elen.p; tab.norm = h->fa_reg.p; tab.n_seq = h->n_hdr;     // slices go by the line-regular column
        if (h->build_pending) { tab.n_seq_dev = (const long long *)&ctl_totals(h)->n_hdr; tab.n_seq = h->hdr.cap; }
    }
    // lanes per query for intervals by record id (FX_FETCH_G: tuning).  The kernel is latency-bound -- PMC (profiles/r02_pmc_fetch.txt):
    // 57 % of the wave cycles parked in s_waitcnt, 22 % issuing; queries sorted by offset, which share DRAM pages and even lines,
    // are only 3-6 % faster -- so what counts is how many queries a wave keeps in flight: 4 lanes (16 queries per wave, a 100-base
    // query in two steps of 64 bytes) 0.117 ms per 1 M, 8 lanes 0.134, 2 lanes 0.160, 16 lanes 0.225.
    static const int fetch_g = [] { const char *e = getenv("FX_FETCH_G"); return e ? atoi(e) : 4; }();
    static const bool lean = [] { const char *e = getenv("FX_FETCH_LEAN"); return !e || atoi(e) != 0; }();
    if (!longq && by_id && fetch_g == 4 && lean && n < 0x7FFFFFFFll) {
        // the line arithmetic alone (k_fetch_lines: few registers, many waves), then the general kernel over what it left over
        int32_t *d_list = nullptr;
        int *d_cnt = nullptr;
        if ((rc = st.scratch<int32_t>(n, &d_list)) || (rc = st.scratch<int>(1, &d_cnt))) return rc;
        HIPCHK(hipMemsetAsync(d_cnt, 0, sizeof(int), h->stream));
        // (pieces of 16 bytes a lane has in flight: 1 -> 71 registers, 7 waves per SIMD, 0.104 ms per 1 M; 2 -> 85, 5 waves, 0.106;
        // the general kernel alone, 102 registers: 0.113-0.121.  FX_FETCH_NP: experiments)
        static const int np = [] { const char *e = getenv("FX_FETCH_NP"); return e ? atoi(e) : 1; }();
        static const bool coal = [] { const char *e = getenv("FX_FETCH_COAL"); return e && atoi(e) != 0; }();
        if (np == 1 && coal) FX_LAUNCH(h, K_FETCH, (k_fetch_lines<4, 1, true>), dim3(fetch_grid((n + 15) / 16)), dim3(BLOCK), h->d_data, h->base, h->n, q, tab, n, flags, d_dst, d_list, d_cnt);
        else if (np == 1) FX_LAUNCH(h, K_FETCH, (k_fetch_lines<4, 1>), dim3(fetch_grid((n + 15) / 16)), dim3(BLOCK), h->d_data, h->base, h->n, q, tab, n, flags, d_dst, d_list, d_cnt);
        else         FX_LAUNCH(h, K_FETCH, (k_fetch_lines<4, 2>), dim3(fetch_grid((n + 15) / 16)), dim3(BLOCK), h->d_data, h->base, h->n, q, tab, n, flags, d_dst, d_list, d_cnt);
        FX_LAUNCH(h, K_FETCH_REST, (k_fetch<true, 4, 16>), dim3(std::min(fetch_grid((n + 15) / 16), 1024u)), dim3(BLOCK), h->d_data, h->base, h->n, q, tab, n, flags,
                  d_dst, (const int32_t *)d_list, (const int *)d_cnt);
    } else if (!longq && by_id && fetch_g == 4) {
        FX_LAUNCH(h, K_FETCH, (k_fetch<true, 4, 16>), dim3(fetch_grid((n + 15) / 16)), dim3(BLOCK), h->d_data, h->base, h->n, q, tab, n, flags, d_dst);
    } else if (!longq && by_id && fetch_g == 2) {
        FX_LAUNCH(h, K_FETCH, (k_fetch<true, 2, 16>), dim3(fetch_grid((n + 31) / 32)), dim3(BLOCK), h->d_data, h->base, h->n, q, tab, n, flags, d_dst);
    } else if (!longq && by_id && fetch_g == 16) {
        FX_LAUNCH(h, K_FETCH, (k_fetch<true, 16, 16>), dim3(fetch_grid((n + 3) / 4)), dim3(BLOCK), h->d_data, h->base, h->n, q, tab, n, flags, d_dst);
    } else if (!longq) {     // 8 lanes x 16 B per query: 8 queries in flight per wave (measured 0.19 ms vs 0.25 ms for 16 x 8 B)
        const unsigned grid8 = fetch_grid((n + 7) / 8);
        if (by_id) FX_LAUNCH(h, K_FETCH, (k_fetch<true, 8, 16>), dim3(grid8), dim3(BLOCK), h->d_data, h->base, h->n, q, tab, n, flags, d_dst);
        else       FX_LAUNCH(h, K_FETCH, (k_fetch<false, 8, 16>), dim3(grid8), dim3(BLOCK), h->d_data, h->base, h->n, q, tab, n, flags, d_dst);
    } else {
        const unsigned grid = fetch_grid(n);
        if (by_id) FX_LAUNCH(h, K_FETCH, (k_fetch<true, 64, 16>), dim3(grid), dim3(BLOCK), h->d_data, h->base, h->n, q, tab, n, flags, d_dst);
        else       FX_LAUNCH(h, K_FETCH, (k_fetch<false, 64, 16>), dim3(grid), dim3(BLOCK), h->d_data, h->base, h->n, q, tab, n, flags, d_dst);
    }
    HIPCHK(hipGetLastError());
    return FX_OK;
}

static int fetch_common(fx_handle *h, int where, int64_t n, bool by_id, const int64_t *a0, const int64_t *a1,
                        const int64_t *a2, const int64_t *skip, int flags, const uint8_t *qflags, uint8_t *dst,
                        const int64_t *dst_off, int64_t *out_len, int64_t dst_bytes_hint) {
    if (!h) return fail(FX_EINVAL, "null handle");
    if (n < 0 || (n > 0 && (!a0 || !a1 || !a2 || !dst || !dst_off))) return fail(FX_EINVAL, "null query array");
    if (by_id && !h->fasta_built) return fail(FX_ESTATE, "fx_fasta_build has not run");
    if (n == 0) return FX_OK;
    int rc = use_device(h);
    if (rc) return rc;
    Staged st(h);
    FetchQ q;
    memset(&q, 0, sizeof q);
    const int64_t *host_blen = (where == FX_HOST && !by_id) ? a1 : nullptr;
    uint8_t *d_dst = dst;
    int64_t *d_len = out_len;
    int64_t total = dst_bytes_hint;
    if (where == FX_HOST) {
        const int64_t *d0, *d1, *d2, *d3 = nullptr, *doff;
        const uint8_t *dfl = nullptr;
        st.reserve_pin(n * 8 * (4 + (skip ? 1 : 0)) + (qflags ? n : 0) + 6 * 256);
        if ((rc = st.up(h, a0, n, &d0)) || (rc = st.up(h, a1, n, &d1)) || (rc = st.up(h, a2, n, &d2)) ||
            (rc = st.up(h, skip, n, &d3)) || (rc = st.up(h, dst_off, n, &doff)) || (rc = st.up(h, qflags, n, &dfl)))
            return rc;
        a0 = d0; a1 = d1; a2 = d2; skip = d3; qflags = dfl;
        q.dst_off = doff;
        if ((rc = st.scratch<int64_t>(n, &d_len))) return rc;
    } else {
        q.dst_off = dst_off;
    }
    if (by_id) { q.seq_id = a0; q.start = a1; q.stop = a2; }
    else       { q.off = a0; q.blen = a1; q.take = a2; q.skip = skip; }
    q.qflags = qflags;
    q.out_len = d_len;
    if (where == FX_HOST) {
        if (total <= 0) return fail(FX_EINVAL, "internal: host fetch needs dst size");
        if ((rc = st.scratch<uint8_t>(total, &d_dst))) return rc;
    }
    // lanes per query: 16 (128-byte window, 4 queries per wave) for short random access,
    // 64 (1 KiB window) when the caller says the ranges are long (FX_LONG) or host arrays show it
    bool longq = (flags & 16) != 0;
    if (where == FX_HOST && !by_id && host_blen) {
        double sum = 0;
        for (int64_t i = 0; i < n; ++i) sum += (double)host_blen[i];
        longq = longq || sum / (double)n > 512.0;
    }
    if ((rc = fetch_launch(h, st, q, by_id, longq, n, flags, d_dst))) return rc;
    if (where == FX_HOST) {
        if ((rc = to_host_any(h, dst, d_dst, total))) return rc;
        if (out_len && (rc = to_host_any(h, out_len, d_len, n * 8))) return rc;
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return FX_OK;
}

static int64_t host_extent(int64_t n, const int64_t *dst_off, const int64_t *take, const int64_t *start, const int64_t *stop) {
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t t = take ? take[i] : (stop[i] - start[i]);
        m = std::max(m, dst_off[i] + std::max<int64_t>(t, 0));
    }
    return m;
}

extern "C" int fx_fetch_ranges(fx_handle *h, int where, int64_t n, const int64_t *off, const int64_t *blen,
                               const int64_t *slen, int flags, const uint8_t *flags_per_query, uint8_t *dst,
                               const int64_t *dst_off, int64_t *out_len) {
    int64_t ext = 0;
    if (where == FX_HOST && n > 0 && dst_off && slen) ext = std::max<int64_t>(1, host_extent(n, dst_off, slen, nullptr, nullptr));
    return fetch_common(h, where, n, false, off, blen, slen, nullptr, flags, flags_per_query, dst, dst_off, out_len, ext);
}

// ------------------------------------------------------------------- routing of a query batch over byte-range shards (host only)
extern "C" int fx_shard_route(int64_t n, const int64_t *ids, const int64_t *starts, const int64_t *stops, int64_t n_rec,
                              const int64_t *boff, const int64_t *blen, const int64_t *llen, const int64_t *elen, const uint8_t *reg,
                              int n_shard, const int64_t *bases, const int64_t *ends, int flags, const uint8_t *flags_per_query,
                              int64_t *order, int64_t *shard_start, int64_t *off, int64_t *len, int64_t *skip, int64_t *take,
                              uint8_t *fl, int32_t *cnt) {
    if (n < 0 || n_shard <= 0 || !bases || !ends || !shard_start) return fail(FX_EINVAL, "fx_shard_route: bad shard list");
    if (n && (!ids || !starts || !stops || !boff || !blen || !llen || !elen || !reg || !order || !off || !len || !skip || !take || !fl || !cnt))
        return fail(FX_EINVAL, "fx_shard_route: null argument");
    const int64_t stream_end = ends[n_shard - 1];
    const int T = (int)std::clamp<int64_t>(n / 65536, 1, 16);          // threads: blocks of consecutive queries (the order stays stable)
    std::vector<int64_t> counts((size_t)T * n_shard, 0);
    std::vector<int32_t> first((size_t)std::max<int64_t>(n, 1));
    std::atomic<int64_t> bad{-1};
    auto shard_of = [&](int64_t x) {                                   // last shard whose base is <= x (0 before the first)
        const int r = (int)(std::upper_bound(bases, bases + n_shard, x) - bases) - 1;
        return r < 0 ? 0 : r;
    };
    // the reference's line arithmetic, and where the range lies
    auto range_of = [&](int64_t i, int64_t &o, int64_t &l, int64_t &sk, int64_t &tk) {
        const int64_t id = ids[i], a = starts[i], b = stops[i];
        const int64_t bpl = llen[id] - elen[id];
        if (reg[id] && bpl > 0) {                                      // sequence.c:498-510
            const int64_t bs = a / bpl, be = b / bpl;
            o = boff[id] + a + elen[id] * bs; l = (b - a) + (be - bs) * elen[id]; sk = 0;
        } else { o = boff[id]; l = blen[id]; sk = a; }                 // sequence.c:100-110
        tk = b - a;
    };
    auto pass = [&](int t, bool scatter, const int64_t *base_pos) {
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        std::vector<int64_t> pos;
        if (scatter) pos.assign(base_pos + (size_t)t * n_shard, base_pos + (size_t)(t + 1) * n_shard);
        for (int64_t i = lo; i < hi; ++i) {
            if (!scatter) {
                if (ids[i] < 0 || ids[i] >= n_rec) { bad.store(i); first[i] = 0; continue; }
                int64_t o, l, sk, tk;
                range_of(i, o, l, sk, tk);
                first[i] = shard_of(o);
                ++counts[(size_t)t * n_shard + first[i]];
            } else {
                int64_t o, l, sk, tk;
                range_of(i, o, l, sk, tk);
                const int64_t k = pos[first[i]]++;
                const int64_t stop = std::min(o + std::max<int64_t>(l, 0), stream_end);
                order[k] = i; off[k] = o; len[k] = l; skip[k] = sk; take[k] = tk;
                fl[k] = flags_per_query ? flags_per_query[i] : (uint8_t)flags;
                cnt[k] = stop > o ? (int32_t)(shard_of(stop - 1) - first[i] + 1) : 0;
            }
        }
    };
    auto run = [&](bool scatter, const int64_t *base_pos) {
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t) th.emplace_back(pass, t, scatter, base_pos);
        pass(0, scatter, base_pos);
        for (auto &x : th) x.join();
    };
    run(false, nullptr);
    if (bad.load() >= 0) return fail(FX_EINVAL, "fx_shard_route: query %lld names record %lld of %lld", (long long)bad.load(),
                                     (long long)ids[bad.load()], (long long)n_rec);
    std::vector<int64_t> base_pos((size_t)T * n_shard);
    int64_t at = 0;
    for (int r = 0; r < n_shard; ++r) {
        shard_start[r] = at;
        for (int t = 0; t < T; ++t) { base_pos[(size_t)t * n_shard + r] = at; at += counts[(size_t)t * n_shard + r]; }
    }
    shard_start[n_shard] = at;
    run(true, base_pos.data());
    return FX_OK;
}

extern "C" int fx_fetch_slices(fx_handle *h, int where, int64_t n, const int64_t *off, const int64_t *blen, const int64_t *skip,
                               const int64_t *take, int flags, const uint8_t *flags_per_query, uint8_t *dst,
                               const int64_t *dst_off, int64_t *out_len) {
    if (n > 0 && !skip) return fail(FX_EINVAL, "null query array");
    int64_t ext = 0;
    if (where == FX_HOST && n > 0 && dst_off && take) ext = std::max<int64_t>(1, host_extent(n, dst_off, take, nullptr, nullptr));
    return fetch_common(h, where, n, false, off, blen, take, skip, flags, flags_per_query, dst, dst_off, out_len, ext);
}

// One range for one caller (Sequence.seq, Read.seq, fa.fetch of a single interval): the per-object API of the
// reference makes one such call per getter.  A batch entry point pays for it with three small uploads, a download
// and two synchronisations (~65 us); here the descriptor and the result live in pinned host memory that the kernel
// reads and writes directly, so a call is one launch and one wait.
static const int64_t ONE_CAP = 1 << 20;
extern "C" int fx_fetch_one(fx_handle *h, int64_t off, int64_t blen, int64_t skip, int64_t take, int flags, uint8_t *dst,
                            int64_t *out_len) {
    if (!h || !out_len || (take > 0 && !dst)) return fail(FX_EINVAL, "null argument");
    *out_len = 0;
    if (take <= 0 || blen <= 0) return FX_OK;
    if (take > ONE_CAP) {                                    // long results: the batch path (staged copies)
        const int64_t zero = 0;
        return fetch_common(h, FX_HOST, 1, false, &off, &blen, &take, &skip, flags, nullptr, dst, &zero, out_len, take);
    }
    int rc = use_device(h);
    if (rc) return rc;
    if (!h->one_box) {
        HIPCHK(hipHostMalloc((void **)&h->one_box, sizeof(*h->one_box), hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void **)&h->one_out, (size_t)ONE_CAP + 64, hipHostMallocDefault));
    }
    // ---- the resident kernel: post the request in pinned memory, spin on the acknowledgement
    static const bool mb_env_off = [] { const char *e = getenv("FX_NO_MAILBOX"); return e && atoi(e) != 0; }();
    if (take <= MB_OUT && !h->mb_off && !mb_env_off && !h->build_pending) {
        if (!h->mb) {
            if (hipHostMalloc((void **)&h->mb, sizeof(Mailbox), hipHostMallocDefault) != hipSuccess ||
                hipStreamCreateWithFlags(&h->mb_stream, hipStreamNonBlocking) != hipSuccess) {
                if (h->mb) { (void)hipHostFree(h->mb); h->mb = nullptr; }
                h->mb_off = true;
            } else memset(h->mb, 0, sizeof(Mailbox));
        }
        if (h->mb) {
            Mailbox *mb = h->mb;
            const unsigned long long n = ++h->mb_seq;
            __atomic_store_n(&mb->tail, n, __ATOMIC_RELEASE);            // tail first, head last: see struct Mailbox
            mb->off = off; mb->blen = blen; mb->skip = skip; mb->take = take; mb->flags_quit = (long long)(unsigned)flags;
            __atomic_store_n(&mb->head, n, __ATOMIC_RELEASE);
            auto launch = [&]() {
                mb->state = 1;
                hipLaunchKernelGGL(k_mailbox, dim3(1), dim3(64), 0, h->mb_stream, h->d_data, h->base, h->n, mb, h->one_out, n - 1, 300);
                h->mb_running = hipGetLastError() == hipSuccess;
                return h->mb_running;
            };
            // (a kernel that has left wrote state = 0 as its last act: the next one queues behind it on the same stream)
            bool ok = (h->mb_running && __atomic_load_n(&mb->state, __ATOMIC_ACQUIRE) != 0) || launch();
            const auto t0 = std::chrono::steady_clock::now();
            auto answered = [&]() { return (__atomic_load_n(&mb->ack, __ATOMIC_ACQUIRE) >> 20) == n; };
            for (unsigned spins = 0; ok; ++spins) {
                if (answered()) break;
                if ((spins & 63) == 63) {
                    // the kernel may have left between two requests: its stream is idle then, and the request unanswered
                    if (__atomic_load_n(&mb->state, __ATOMIC_ACQUIRE) != 1 && hipStreamQuery(h->mb_stream) == hipSuccess) {
                        if (answered()) break;
                        ok = launch();
                    }
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2000)) ok = false;
                }
            }
            if (ok) {
                const int64_t got = (int64_t)(mb->ack & 0xFFFFFull);
                if (got < 0 || got > take) return fail(FX_ERANGE, "range outside the stream");
                memcpy(dst, h->one_out, (size_t)got);
                *out_len = got;
                return FX_OK;
            }
            // no answer in two seconds (or no launch): stop using the mailbox, send the kernel home, take the launch path
            h->mb_off = true;
            { const unsigned long long q = ++h->mb_seq; __atomic_store_n(&mb->tail, q, __ATOMIC_RELEASE); mb->flags_quit = 1ll << 32; __atomic_store_n(&mb->head, q, __ATOMIC_RELEASE); }
            (void)hipStreamSynchronize(h->mb_stream);
            h->mb_running = false;
        }
    }
    *h->one_box = {off, blen, skip, take, 0, 0};
    FetchQ q;
    memset(&q, 0, sizeof q);
    q.off = &h->one_box->off; q.blen = &h->one_box->blen; q.skip = &h->one_box->skip; q.take = &h->one_box->take;
    q.dst_off = &h->one_box->dst_off; q.out_len = &h->one_box->out_len;
    FastaTab tab;
    memset(&tab, 0, sizeof tab);
    if (blen > 512) hipLaunchKernelGGL((k_fetch<false, 64, 16>), dim3(1), dim3(BLOCK), 0, h->stream, h->d_data, h->base, h->n, q, tab, (int64_t)1, flags, h->one_out);
    else            hipLaunchKernelGGL((k_fetch<false, 8, 16>), dim3(1), dim3(BLOCK), 0, h->stream, h->d_data, h->base, h->n, q, tab, (int64_t)1, flags, h->one_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    const int64_t got = h->one_box->out_len;
    if (got < 0 || got > take) return fail(FX_ERANGE, "range outside the stream");
    memcpy(dst, h->one_out, (size_t)got);
    *out_len = got;
    return FX_OK;
}

extern "C" int fx_fasta_fetch(fx_handle *h, int where, int64_t n, const int64_t *seq_id, const int64_t *start,
                              const int64_t *stop, int flags, const uint8_t *flags_per_query, uint8_t *dst,
                              const int64_t *dst_off, int64_t *out_len) {
    int64_t ext = 0;
    if (where == FX_HOST && n > 0 && dst_off && start && stop) ext = std::max<int64_t>(1, host_extent(n, dst_off, nullptr, start, stop));
    return fetch_common(h, where, n, true, seq_id, start, stop, nullptr, flags, flags_per_query, dst, dst_off, out_len, ext);
}

// ---- batches whose answers the library lays out itself (fx_*_fetch_alloc)
// Host-side phases of the last fx_*_fetch_alloc call of this thread, in milliseconds: 0 query arrays staged and their
// copies enqueued, 1 counts + scan + offsets back (first wait), 2 pinned blocks for the answers, 3 kernels enqueued,
// 4 answers back (second wait), 5 the whole call
static thread_local double g_fetch_phase[6] = {0, 0, 0, 0, 0, 0};
extern "C" int fx_fetch_phases(double *ms, int cap) {
    if (!ms || cap <= 0) return fail(FX_EINVAL, "null argument");
    for (int i = 0; i < cap && i < 6; ++i) ms[i] = g_fetch_phase[i];
    return FX_OK;
}
struct PhaseClock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
    void lap(int i) { const auto now = std::chrono::steady_clock::now(); g_fetch_phase[i] = std::chrono::duration<double, std::milli>(now - last).count(); last = now; }
    void done() { g_fetch_phase[5] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

// cnt[i] = bytes query i will write (0 for a query that is not valid; *bad = index of the first such query)
__global__ __launch_bounds__(BLOCK) void k_q_counts_fasta(const int64_t *__restrict__ slen, int64_t n_seq, const int64_t *__restrict__ id,
                                                         const int64_t *__restrict__ a, const int64_t *__restrict__ b, int64_t n,
                                                         int32_t *__restrict__ cnt, unsigned long long *__restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const int64_t r = id[i], x = a[i], y = b[i];
    const bool ok = r >= 0 && r < n_seq && x >= 0 && y >= x && y - x <= 0x7FFFFFFFll && y <= slen[r];
    cnt[i] = ok ? (int32_t)(y - x) : 0;
    if (!ok) atomicMin(bad, (unsigned long long)i);
}
__global__ __launch_bounds__(BLOCK) void k_q_counts_fastq(const int64_t *__restrict__ rlen, int64_t n_reads, const int64_t *__restrict__ id, int64_t n,
                                                         int32_t *__restrict__ cnt, unsigned long long *__restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const int64_t r = id[i];
    const bool ok = r >= 0 && r < n_reads && rlen[r] >= 0 && rlen[r] <= 0x7FFFFFFFll;
    cnt[i] = ok ? (int32_t)rlen[r] : 0;
    if (!ok) atomicMin(bad, (unsigned long long)i);
}

// cnt[0..n) on the device -> exclusive offsets d_off[0..n] on the device, and in *offs_out a pinned host copy (fx_pinned_free)
// with the index of the first invalid query in *first_bad (-1: none); one wait
static int offsets_of_counts(fx_handle *h, Staged &st, const int32_t *d_cnt, unsigned long long *d_bad, int64_t n, int64_t **d_off_out,
                             int64_t **offs_out, int64_t *first_bad) {
    int rc;
    const int64_t nchunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    int64_t *d_sums = nullptr, *d_off = nullptr;
    if ((rc = st.scratch<int64_t>(nchunks + 1, &d_sums)) || (rc = st.scratch<int64_t>(n + 1, &d_off))) return rc;
    hipLaunchKernelGGL(k_cnt_chunk_sums, dim3((unsigned)nchunks), dim3(BLOCK), 0, h->stream, d_cnt, n, d_sums);
    hipLaunchKernelGGL(k_cnt_chunk_bases, dim3(1), dim3(BLOCK), 0, h->stream, d_sums, nchunks);
    hipLaunchKernelGGL(k_cnt_offsets, dim3((unsigned)nchunks), dim3(BLOCK), 0, h->stream, d_cnt, n, (const int64_t *)d_sums, d_off);
    HIPCHK(hipGetLastError());
    int64_t *offs = (int64_t *)fx_pinned_alloc((n + 2) * 8);            // one more word: the first invalid query
    if (!offs) return FX_ENOMEM;
    hipError_t e = hipMemcpyAsync(offs, d_off, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(offs + n + 1, d_bad, 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { fx_pinned_free(offs); return fail(FX_EDEVICE, "offsets of a batch: %s", hipGetErrorString(e)); }
    const unsigned long long bad = (unsigned long long)offs[n + 1];
    *first_bad = bad == ~0ull ? -1 : (int64_t)bad;
    *d_off_out = d_off;
    *offs_out = offs;
    return FX_OK;
}

extern "C" int fx_fasta_fetch_alloc(fx_handle *h, int64_t n, const int64_t *seq_id, const int64_t *start, const int64_t *stop, int flags,
                                    const uint8_t *flags_per_query, uint8_t **dst, int64_t **dst_off, int64_t *first_bad) {
    if (!h || !dst || !dst_off || !first_bad) return fail(FX_EINVAL, "null argument");
    *dst = nullptr; *dst_off = nullptr; *first_bad = -1;
    if (n < 0 || (n > 0 && (!seq_id || !start || !stop))) return fail(FX_EINVAL, "null query array");
    if (!h->fasta_built) return fail(FX_ESTATE, "fx_fasta_build has not run");
    int rc = use_device(h);
    if (!rc) rc = finish_build(h);
    if (rc) return rc;
    if (n >= 0x7FFFFFFFll * (int64_t)SCAN_CHUNK) return fail(FX_ERANGE, "too many queries in one batch");
    Staged st(h);
    PhaseClock pc;
    FetchQ q;
    memset(&q, 0, sizeof q);
    int64_t *offs = nullptr;
    if (n == 0) {
        if (!(offs = (int64_t *)fx_pinned_alloc(16)) || !(*dst = (uint8_t *)fx_pinned_alloc(1))) { fx_pinned_free(offs); return FX_ENOMEM; }
        offs[0] = 0; *dst_off = offs;
        return FX_OK;
    }
    st.reserve_pin(n * 8 * 3 + (flags_per_query ? n : 0) + 5 * 256);
    if ((rc = st.up(h, seq_id, n, &q.seq_id)) || (rc = st.up(h, start, n, &q.start)) || (rc = st.up(h, stop, n, &q.stop)) ||
        (rc = st.up(h, flags_per_query, n, &q.qflags)))
        return rc;
    int32_t *d_cnt = nullptr;
    unsigned long long *d_bad = nullptr;
    if ((rc = st.scratch<int32_t>(n, &d_cnt)) || (rc = st.scratch<unsigned long long>(1, &d_bad))) return rc;
    HIPCHK(hipMemsetAsync(d_bad, 0xFF, 8, h->stream));
    hipLaunchKernelGGL(k_q_counts_fasta, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, h->stream, (const int64_t *)h->fa_slen.p, h->n_hdr, q.seq_id, q.start,
                       q.stop, n, d_cnt, d_bad);
    pc.lap(0);
    int64_t *d_off = nullptr;
    if ((rc = offsets_of_counts(h, st, d_cnt, d_bad, n, &d_off, &offs, first_bad))) return rc;
    pc.lap(1);
    if (*first_bad >= 0) { fx_pinned_free(offs); return fail(FX_ERANGE, "query %lld: record id or interval outside the sequence", (long long)*first_bad); }
    const int64_t total = offs[n];
    uint8_t *out = (uint8_t *)fx_pinned_alloc(std::max<int64_t>(total, 1));
    uint8_t *d_dst = nullptr;
    if (!out) { fx_pinned_free(offs); return FX_ENOMEM; }
    // (a copy into these blocks may still be in flight when something fails: the stream is waited for before they go back to the pool)
    auto bail = [&](int code) { (void)hipStreamSynchronize(h->stream); fx_pinned_free(offs); fx_pinned_free(out); return code; };
    if ((rc = st.scratch<uint8_t>(std::max<int64_t>(total, 1), &d_dst))) return bail(rc);
    pc.lap(2);
    q.dst_off = d_off;
    if (total > 0) {
        if ((rc = fetch_launch(h, st, q, true, (flags & 16) != 0 || total / n > 512, n, flags, d_dst))) return bail(rc);
        if (hipMemcpyAsync(out, d_dst, (size_t)total, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return bail(fail(FX_EDEVICE, "D2H failed"));
    }
    pc.lap(3);
    if (hipStreamSynchronize(h->stream) != hipSuccess) return bail(fail(FX_EDEVICE, "stream synchronisation failed"));
    pc.lap(4);
    h->prof.drain();
    *dst = out; *dst_off = offs;
    pc.done();
    return FX_OK;
}

extern "C" int fx_fastq_fetch_alloc(fx_handle *h, int64_t n, const int64_t *read_id, int phred, int seq_flags, int want, uint8_t **seq,
                                    uint8_t **qual, int8_t **quali, int64_t **dst_off, int64_t *first_bad) {
    if (!h || !dst_off || !first_bad) return fail(FX_EINVAL, "null argument");
    if (seq) *seq = nullptr; if (qual) *qual = nullptr; if (quali) *quali = nullptr;
    *dst_off = nullptr; *first_bad = -1;
    if (!h->fastq_built) return fail(FX_ESTATE, "fx_fastq_build has not run");
    if (n < 0 || (n > 0 && !read_id)) return fail(FX_EINVAL, "null query array");
    int rc = use_device(h);
    if (rc) return rc;
    if (!phred) phred = 33;                                // read.c:268
    const bool w_seq = (want & 1) && seq, w_qual = (want & 2) && qual, w_qi = (want & 4) && quali;
    Staged st(h);
    PhaseClock pc;
    int64_t *offs = nullptr;
    void *outs[3] = {nullptr, nullptr, nullptr};
    auto bail = [&](int code) { (void)hipStreamSynchronize(h->stream); fx_pinned_free(offs); for (void *p : outs) fx_pinned_free(p); return code; };
    if (n == 0) {
        if (!(offs = (int64_t *)fx_pinned_alloc(16))) return FX_ENOMEM;
        offs[0] = 0;
    } else {
        st.reserve_pin(n * 8 + 512);
        const int64_t *d_ids = nullptr;
        if ((rc = st.up(h, read_id, n, &d_ids))) return rc;
        int32_t *d_cnt = nullptr;
        unsigned long long *d_bad = nullptr;
        if ((rc = st.scratch<int32_t>(n, &d_cnt)) || (rc = st.scratch<unsigned long long>(1, &d_bad))) return rc;
        HIPCHK(hipMemsetAsync(d_bad, 0xFF, 8, h->stream));
        hipLaunchKernelGGL(k_q_counts_fastq, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, h->stream, (const int64_t *)h->fq_rlen.p, h->n_reads, d_ids, n, d_cnt, d_bad);
        pc.lap(0);
        int64_t *d_off = nullptr;
        if ((rc = offsets_of_counts(h, st, d_cnt, d_bad, n, &d_off, &offs, first_bad))) return rc;
        pc.lap(1);
        if (*first_bad >= 0) return bail(fail(FX_ERANGE, "read id %lld out of range", (long long)read_id[*first_bad]));
        const int64_t total = std::max<int64_t>(offs[n], 1);
        uint8_t *d_out[3] = {nullptr, nullptr, nullptr};
        const bool w[3] = {w_seq, w_qual, w_qi};
        for (int k = 0; k < 3; ++k)
            if (w[k]) {
                if (!(outs[k] = fx_pinned_alloc(total))) return bail(FX_ENOMEM);
                if ((rc = st.scratch<uint8_t>(total, &d_out[k]))) return bail(rc);
            }
        pc.lap(2);
        FX_LAUNCH(h, K_FASTQ_FETCH, k_fastq_fetch, dim3(fetch_grid((n + 3) / 4)), dim3(BLOCK), h->d_data, h->base, h->n, h->fq_rlen.p,
                  h->fq_soff.p, h->fq_qoff.p, h->n_reads, d_ids, n, phred, seq_flags, d_out[0], d_out[1], (int8_t *)d_out[2], (const int64_t *)d_off);
        if (hipGetLastError() != hipSuccess) return bail(fail(FX_EDEVICE, "launch failed"));
        for (int k = 0; k < 3; ++k)
            if (w[k] && hipMemcpyAsync(outs[k], d_out[k], (size_t)offs[n], hipMemcpyDeviceToHost, h->stream) != hipSuccess) return bail(fail(FX_EDEVICE, "D2H failed"));
        pc.lap(3);
        if (hipStreamSynchronize(h->stream) != hipSuccess) return bail(fail(FX_EDEVICE, "stream synchronisation failed"));
        pc.lap(4);
        h->prof.drain();
    }
    if (n == 0) for (int k = 0; k < 3; ++k) { const bool w[3] = {w_seq, w_qual, w_qi}; if (w[k] && !(outs[k] = fx_pinned_alloc(1))) return bail(FX_ENOMEM); }
    if (w_seq) *seq = (uint8_t *)outs[0];
    if (w_qual) *qual = (uint8_t *)outs[1];
    if (w_qi) *quali = (int8_t *)outs[2];
    *dst_off = offs;
    pc.done();
    return FX_OK;
}

extern "C" int fx_fastq_fetch(fx_handle *h, int where, int64_t n, const int64_t *read_id, int phred, int seq_flags,
                              uint8_t *seq, uint8_t *qual, int8_t *quali, const int64_t *dst_off) {
    if (!h) return fail(FX_EINVAL, "null handle");
    if (!h->fastq_built) return fail(FX_ESTATE, "fx_fastq_build has not run");
    if (n < 0 || (n > 0 && (!read_id || !dst_off))) return fail(FX_EINVAL, "null query array");
    if (n == 0) return FX_OK;
    int rc = use_device(h);
    if (rc) return rc;
    if (!phred) phred = 33;                                // read.c:268
    Staged st(h);
    const int64_t *d_ids = read_id, *d_off = dst_off;
    uint8_t *d_seq = seq, *d_qual = qual;
    int8_t *d_qi = quali;
    int64_t total = 0;
    std::vector<int64_t> rl;
    if (where == FX_HOST) {
        // output extent needs rlen of the requested reads: gather them on the device (n values, not the whole column)
        if ((rc = st.up(h, read_id, n, &d_ids))) return rc;
        int64_t *d_rl = nullptr;
        if ((rc = st.scratch<int64_t>(n, &d_rl))) return rc;
        hipLaunchKernelGGL(k_gather_i64, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, h->stream, h->fq_rlen.p, h->n_reads, d_ids, n, d_rl);
        rl.resize((size_t)n);
        HIPCHK(hipMemcpyAsync(rl.data(), d_rl, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        for (int64_t i = 0; i < n; ++i) {
            if (rl[(size_t)i] < 0) return fail(FX_ERANGE, "read id %lld out of range", (long long)read_id[i]);
            total = std::max(total, dst_off[i] + rl[(size_t)i]);
        }
        total = std::max<int64_t>(total, 1);
        if ((rc = st.up(h, dst_off, n, &d_off))) return rc;
        if (seq && (rc = st.scratch<uint8_t>(total, &d_seq))) return rc;
        if (qual && (rc = st.scratch<uint8_t>(total, &d_qual))) return rc;
        if (quali && (rc = st.scratch<int8_t>(total, &d_qi))) return rc;
    }
    FX_LAUNCH(h, K_FASTQ_FETCH, k_fastq_fetch, dim3(fetch_grid((n + 3) / 4)), dim3(BLOCK), h->d_data, h->base, h->n, h->fq_rlen.p,
                       h->fq_soff.p, h->fq_qoff.p, h->n_reads, d_ids, n, phred, seq_flags, d_seq, d_qual, d_qi, d_off);
    HIPCHK(hipGetLastError());
    if (where == FX_HOST) {
        if (seq) HIPCHK(hipMemcpyAsync(seq, d_seq, (size_t)total, hipMemcpyDeviceToHost, h->stream));
        if (qual) HIPCHK(hipMemcpyAsync(qual, d_qual, (size_t)total, hipMemcpyDeviceToHost, h->stream));
        if (quali) HIPCHK(hipMemcpyAsync(quali, d_qi, (size_t)total, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return FX_OK;
}

extern "C" int fx_read_fetch(fx_handle *h, int where, int64_t n, const int64_t *soff, const int64_t *qoff,
                             const int64_t *rlen, int phred, int seq_flags, uint8_t *seq, uint8_t *qual, int8_t *quali,
                             const int64_t *dst_off) {
    if (!h) return fail(FX_EINVAL, "null handle");
    if (n < 0 || (n > 0 && (!soff || !qoff || !rlen || !dst_off))) return fail(FX_EINVAL, "null query array");
    if (n == 0) return FX_OK;
    int rc = use_device(h);
    if (rc) return rc;
    if (!phred) phred = 33;                                // read.c:268
    Staged st(h);
    const int64_t *d_s = soff, *d_q = qoff, *d_r = rlen, *d_off = dst_off;
    uint8_t *d_seq = seq, *d_qual = qual;
    int8_t *d_qi = quali;
    int64_t total = 0;
    if (where == FX_HOST) {
        for (int64_t i = 0; i < n; ++i) {
            if (rlen[i] < 0 || soff[i] < h->base || qoff[i] < h->base || soff[i] + rlen[i] > h->base + h->n ||
                qoff[i] + rlen[i] > h->base + h->n)
                return fail(FX_ERANGE, "read %lld lies outside the stream", (long long)i);
            total = std::max(total, dst_off[i] + rlen[i]);
        }
        total = std::max<int64_t>(total, 1);
        if ((rc = st.up(h, soff, n, &d_s)) || (rc = st.up(h, qoff, n, &d_q)) || (rc = st.up(h, rlen, n, &d_r)) ||
            (rc = st.up(h, dst_off, n, &d_off)))
            return rc;
        if (seq && (rc = st.scratch<uint8_t>(total, &d_seq))) return rc;
        if (qual && (rc = st.scratch<uint8_t>(total, &d_qual))) return rc;
        if (quali && (rc = st.scratch<int8_t>(total, &d_qi))) return rc;
    }
    FX_LAUNCH(h, K_FASTQ_FETCH, k_fastq_fetch, dim3(fetch_grid((n + 3) / 4)), dim3(BLOCK), h->d_data, h->base, h->n, d_r, d_s, d_q, n,
              (const int64_t *)nullptr, n, phred, seq_flags, d_seq, d_qual, d_qi, d_off);
    HIPCHK(hipGetLastError());
    if (where == FX_HOST) {
        if (seq) HIPCHK(hipMemcpyAsync(seq, d_seq, (size_t)total, hipMemcpyDeviceToHost, h->stream));
        if (qual) HIPCHK(hipMemcpyAsync(qual, d_qual, (size_t)total, hipMemcpyDeviceToHost, h->stream));
        if (quali) HIPCHK(hipMemcpyAsync(quali, d_qi, (size_t)total, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return FX_OK;
}

// ------------------------------------------------------------- names
__global__ void k_add_i64(const int64_t *__restrict__ a, int64_t add, int64_t n, int64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + add;
}

extern "C" int fx_names_build(fx_handle *h, int kind) {
    if (!h || (kind != 0 && kind != 1)) return fail(FX_EINVAL, "bad argument");
    if (kind == 0 ? !h->fasta_built : !h->fastq_built) return fail(FX_ESTATE, "the index has not been built");
    if (kind == 0 && !h->hdr.p) return fail(FX_ESTATE, "names need a scanned index (fx_fasta_build), not an installed table");
    int rc = use_device(h);
    if (!rc) rc = finish_build(h);
    if (rc) return rc;
    const int64_t n = kind == 0 ? h->n_hdr : h->n_reads;
    if (n >= 0xFFFFFFFFll) return fail(FX_ERANGE, "too many records for the 32-bit name table");
    int64_t cap = 1024;
    while (cap < 2 * n) cap <<= 1;
    if ((rc = h->nm_table.alloc(cap))) return rc;
    HIPCHK(hipMemsetAsync(h->nm_table.p, 0, (size_t)cap * 4, h->stream));
    const int64_t *noff = h->fq_name_off.p;
    const int32_t *nlen = h->fq_name_len.p;
    if (kind == 0) {
        if ((rc = h->nm_off.alloc(std::max<int64_t>(n, 1)))) return rc;
        hipLaunchKernelGGL(k_add_i64, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, h->stream, h->hdr.p, (int64_t)1, n, h->nm_off.p);
        noff = h->nm_off.p; nlen = h->fa_name_len.p;
    }
    if (n) hipLaunchKernelGGL(k_names_build, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, h->stream, h->d_data, h->base, noff, nlen, n,
                              h->nm_table.p, (uint64_t)(cap - 1));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    h->nm_mask = (uint64_t)(cap - 1);
    h->nm_kind = kind;
    return FX_OK;
}

extern "C" int fx_names_lookup(fx_handle *h, int where, int64_t nq, const uint8_t *qbytes, const int64_t *qoff, int64_t *out_ids) {
    if (!h || nq < 0 || (nq > 0 && (!qbytes || !qoff || !out_ids))) return fail(FX_EINVAL, "bad argument");
    if (h->nm_kind < 0) return fail(FX_ESTATE, "fx_names_build has not run");
    if (nq == 0) return FX_OK;
    int rc = use_device(h);
    if (rc) return rc;
    Staged st(h);
    const uint8_t *d_q = qbytes;
    const int64_t *d_off = qoff;
    int64_t *d_out = out_ids;
    if (where == FX_HOST) {
        const int64_t total = qoff[nq];
        if ((rc = st.up(h, qbytes, std::max<int64_t>(total, 1) + 8, &d_q))) return rc;     // + 8: whole-word reads of the last key
        if ((rc = st.up(h, qoff, nq + 1, &d_off)) || (rc = st.scratch<int64_t>(nq, &d_out))) return rc;
    }
    const int64_t *noff = h->nm_kind == 0 ? h->nm_off.p : h->fq_name_off.p;
    const int32_t *nlen = h->nm_kind == 0 ? h->fa_name_len.p : h->fq_name_len.p;
    hipLaunchKernelGGL(k_names_lookup, dim3(nblocks(nq, BLOCK)), dim3(BLOCK), 0, h->stream, h->d_data, h->base, noff, nlen,
                       h->nm_table.p, h->nm_mask, d_q, d_off, nq, d_out);
    HIPCHK(hipGetLastError());
    if (where == FX_HOST) {
        HIPCHK(hipMemcpyAsync(out_ids, d_out, (size_t)nq * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return FX_OK;
}

__global__ void k_len_clamp(const int32_t *__restrict__ len, int64_t n, int32_t *__restrict__ l32, int64_t *__restrict__ l64) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int32_t v = len[i] > 0 ? len[i] : 0; l32[i] = v; l64[i] = v; }
}

// The record names back to back, straight from the record table in HBM: no query arrays go up (fx_fetch_ranges with
// host arrays uploads 32 bytes per name and walks them twice on the host: 130 ms for 20 M names, most of it not the names)
extern "C" int fx_names_pack(fx_handle *h, int kind, uint8_t *dst, int64_t cap, int64_t *name_off, int64_t *total_out) {
    if (!h || (kind != 0 && kind != 1) || !total_out || !name_off || cap < 0 || (cap > 0 && !dst)) return fail(FX_EINVAL, "bad argument");
    if (kind == 0 ? !h->fasta_built : !h->fastq_built) return fail(FX_ESTATE, "the index has not been built");
    if (kind == 0 && !h->hdr.p) return fail(FX_ESTATE, "names need a scanned index (fx_fasta_build), not an installed table");
    int rc = use_device(h);
    if (!rc) rc = finish_build(h);
    if (rc) return rc;
    const int64_t n = kind == 0 ? h->n_hdr : h->n_reads;
    name_off[0] = 0;
    *total_out = 0;
    if (n == 0) return FX_OK;
    const int64_t *noff = h->fq_name_off.p;
    const int32_t *nlen = h->fq_name_len.p;
    if (kind == 0) {
        if ((rc = h->nm_off.alloc(n))) return rc;
        hipLaunchKernelGGL(k_add_i64, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, h->stream, h->hdr.p, (int64_t)1, n, h->nm_off.p);
        noff = h->nm_off.p; nlen = h->fa_name_len.p;
    }
    const int64_t nchunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    DevBuf<int32_t> l32;
    DevBuf<int64_t> l64, sums, off;
    DevBuf<uint8_t> out;
    if ((rc = l32.alloc(n)) || (rc = l64.alloc(n)) || (rc = sums.alloc(nchunks + 1)) || (rc = off.alloc(n + 1))) return rc;
    hipLaunchKernelGGL(k_len_clamp, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, h->stream, nlen, n, l32.p, l64.p);
    hipLaunchKernelGGL(k_cnt_chunk_sums, dim3((unsigned)nchunks), dim3(BLOCK), 0, h->stream, l32.p, n, sums.p);
    hipLaunchKernelGGL(k_cnt_chunk_bases, dim3(1), dim3(BLOCK), 0, h->stream, sums.p, nchunks);
    hipLaunchKernelGGL(k_cnt_offsets, dim3((unsigned)nchunks), dim3(BLOCK), 0, h->stream, l32.p, n, sums.p, off.p);
    HIPCHK(hipGetLastError());
    int64_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, off.p + n, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *total_out = total;
    if (total > cap) return fail(FX_ERANGE, "%lld name bytes, room for %lld", (long long)total, (long long)cap);
    if ((rc = to_host(h, name_off, off.p, (n + 1) * 8))) return rc;
    if (total) {
        if ((rc = out.alloc(total + 64))) return rc;
        if ((rc = fetch_common(h, FX_DEVICE, n, false, noff, l64.p, l64.p, nullptr, FX_RAW, nullptr, out.p, off.p, nullptr, 0))) return rc;
        if ((rc = to_host(h, dst, out.p, total))) return rc;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return FX_OK;
}

extern "C" int fx_names_sort(fx_handle *h, int kind, int where, int64_t *order, int64_t *n_dup) {
    if (!h || (kind != 0 && kind != 1) || !n_dup) return fail(FX_EINVAL, "bad argument");
    if (kind == 0 ? !h->fasta_built : !h->fastq_built) return fail(FX_ESTATE, "the index has not been built");
    if (kind == 0 && !h->hdr.p) return fail(FX_ESTATE, "names need a scanned index (fx_fasta_build), not an installed table");
    int rc = use_device(h);
    if (!rc) rc = finish_build(h);
    if (rc) return rc;
    const int64_t n = kind == 0 ? h->n_hdr : h->n_reads;
    *n_dup = 0;
    if (n == 0) return FX_OK;
    if (!order) return fail(FX_EINVAL, "null order");
    if (n >= 0xFFFFFFFFll) return fail(FX_ERANGE, "too many records for the 32-bit sort index");
    const int64_t *noff = h->fq_name_off.p;
    const int32_t *nlen = h->fq_name_len.p;
    if (kind == 0) {
        if ((rc = h->nm_off.alloc(n))) return rc;
        hipLaunchKernelGGL(k_add_i64, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, h->stream, h->hdr.p, (int64_t)1, n, h->nm_off.p);
        noff = h->nm_off.p; nlen = h->fa_name_len.p;
    }
    Staged st(h);
    int64_t *d_order = order, *d_ndup = nullptr;
    if (where == FX_HOST && (rc = st.scratch<int64_t>(n, &d_order))) return rc;
    if ((rc = st.scratch<int64_t>(1, &d_ndup))) return rc;
    const char *what = "";
    const int e = sort_names(h->d_data, h->base, noff, nlen, n, d_order, d_ndup, h->stream, &what);
    if (e) return fail(e == (int)hipErrorOutOfMemory ? FX_ENOMEM : FX_EDEVICE, "name sort, %s: %s", what, hipGetErrorString((hipError_t)e));
    HIPCHK(hipMemcpyAsync(n_dup, d_ndup, 8, hipMemcpyDeviceToHost, h->stream));
    if (where == FX_HOST && (rc = to_host(h, order, d_order, n * 8))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    return FX_OK;
}

// SQLite's BINARY order of n names given as one packed host buffer (name i = names[name_off[i], name_off[i + 1])): what
// fx_names_sort computes from a handle's own table, for names that come from SEVERAL handles -- the shards of one file,
// whose packed names rank 0 has concatenated -- so that the merged .fxi gets its index b-tree from the same GPU sort
// instead of CREATE UNIQUE INDEX's (index.c:363, fastq.c:155).  order[i] = 0-based index of the i-th smallest name.
extern "C" int fx_sort_packed_names(int device, const uint8_t *names, const int64_t *name_off, int64_t n, int64_t *order, int64_t *n_dup) {
    if (!n_dup || n < 0 || (n > 0 && (!name_off || !order || (!names && name_off[n] > 0)))) return fail(FX_EINVAL, "bad argument");
    *n_dup = 0;
    if (n == 0) return FX_OK;
    if (n >= 0xFFFFFFFFll) return fail(FX_ERANGE, "too many records for the 32-bit sort index");
    fx_handle *h = nullptr;
    int rc = new_handle(device, &h);
    if (rc) return rc;
    const int64_t total = name_off[n];
    DevBuf<uint8_t> d_names;
    DevBuf<int64_t> d_off, d_order, d_ndup;
    DevBuf<int32_t> d_len;
    std::vector<int32_t> len((size_t)n);
    for (int64_t i = 0; i < n; ++i) len[(size_t)i] = (int32_t)(name_off[i + 1] - name_off[i]);
    auto done = [&](int code) { fx_close(h); return code; };
    if ((rc = d_names.alloc(total + 64)) || (rc = d_off.alloc(n)) || (rc = d_len.alloc(n)) || (rc = d_order.alloc(n)) || (rc = d_ndup.alloc(1))) return done(rc);
    hipError_t e = hipMemsetAsync(d_names.p + total, 0, 64, h->stream);
    if (e == hipSuccess && total) e = hipMemcpyAsync(d_names.p, names, (size_t)total, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off.p, name_off, (size_t)n * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_len.p, len.data(), (size_t)n * 4, hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) return done(fail(FX_EDEVICE, "H2D: %s", hipGetErrorString(e)));
    const char *what = "";
    const int se = sort_names(d_names.p, 0, d_off.p, d_len.p, n, d_order.p, d_ndup.p, h->stream, &what);
    if (se) return done(fail(se == (int)hipErrorOutOfMemory ? FX_ENOMEM : FX_EDEVICE, "name sort, %s: %s", what, hipGetErrorString((hipError_t)se)));
    e = hipMemcpyAsync(n_dup, d_ndup.p, 8, hipMemcpyDeviceToHost, h->stream);
    if (e != hipSuccess) return done(fail(FX_EDEVICE, "D2H: %s", hipGetErrorString(e)));
    if ((rc = to_host(h, order, d_order.p, n * 8))) return done(rc);
    if (hipStreamSynchronize(h->stream) != hipSuccess) return done(fail(FX_EDEVICE, "stream sync failed"));
    return done(FX_OK);
}

extern "C" int fx_fasta_len_stats(fx_handle *h, int64_t count_min, double half, fx_len_stats *out) {
    static_assert(sizeof(fx_len_stats) == sizeof(LenStats), "statistics layout");
    if (!h || !out) return fail(FX_EINVAL, "null argument");
    if (!h->fasta_built) return fail(FX_ESTATE, "the record table is not resident (fx_fasta_build / fx_fasta_set_table)");
    int rc = use_device(h);
    if (!rc) rc = finish_build(h);
    if (rc) return rc;
    const char *what = "";
    const int e = len_stats(h->fa_slen.p, h->n_hdr, count_min, half, reinterpret_cast<LenStats *>(out), h->stream, &what);
    if (e) return fail(e == (int)hipErrorOutOfMemory ? FX_ENOMEM : FX_EDEVICE, "length statistics, %s: %s", what, hipGetErrorString((hipError_t)e));
    return FX_OK;
}

extern "C" int fx_revcomp(int device, int where, uint8_t *buf, int64_t n, int mode) {
    if (!buf && n) return fail(FX_EINVAL, "null buffer");
    if (n <= 0) return FX_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(FX_EDEVICE, "no HIP device available; libfxgpu has no CPU fallback");
    HIPCHK(hipSetDevice(device));
    uint8_t *d = buf;
    if (where == FX_HOST) {
        HIPCHK(dev_malloc((void **)&d, (size_t)n));
        hipError_t e = hipMemcpy(d, buf, (size_t)n, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(d); return fail(FX_EDEVICE, "H2D: %s", hipGetErrorString(e)); }
    }
    const unsigned nb = (unsigned)std::min<int64_t>(nblocks(n, BLOCK), 2048);
    hipLaunchKernelGGL(k_revcomp, dim3(nb), dim3(BLOCK), 0, 0, d, n, mode);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess && where == FX_HOST) e = hipMemcpy(buf, d, (size_t)n, hipMemcpyDeviceToHost);
    if (where == FX_HOST) (void)hipFree(d);
    if (e != hipSuccess) return fail(FX_EDEVICE, "revcomp: %s", hipGetErrorString(e));
    return FX_OK;
}

extern "C" int fx_gz_points(fx_handle *h, int64_t spacing, int64_t *cmp_off, int64_t *uncmp_off, int64_t cap,
                            int64_t *n_out, int64_t *compressed_size) {
    if (!h || !n_out) return fail(FX_EINVAL, "null argument");
    if (compressed_size) *compressed_size = h->gz_csize;
    int64_t n = 0, next = 0;
    if (h->bgzf) {
        if (spacing <= 0) spacing = 1048576;                 // zran spacing used by the reference (index.c:70)
        for (size_t m = 0; m < h->gz_moff.size(); ++m) {
            if (h->gz_uoff[m] < next && m != 0) continue;
            // a zran point is where a RAW inflate can start (zran_seek: inflateInit2(-15) at cmp_offset): the first byte of the
            // member's deflate data, behind its 18-byte BGZF header -- bits 0, no window (nothing before a member's first block)
            if (cmp_off && uncmp_off && n < cap) { cmp_off[n] = h->gz_coff[m]; uncmp_off[n] = h->gz_uoff[m]; }
            ++n;
            next = h->gz_uoff[m] + spacing;
        }
    }
    *n_out = n;
    return FX_OK;
}

// fx_pgzip.hpp without a device (tests, tools): the whole gzip file `in` -> `out` (cap bytes); 0 done, 1 "not a case for it"
// (too small, several members, no block start found, ...: the caller inflates serially), FX_ERANGE when out is too small.
extern "C" int fx_gunzip_parallel(const uint8_t *in, int64_t n, int threads, uint8_t *out, int64_t cap, int64_t *out_n, int64_t *n_points) {
    if (!in || n < 0 || !out_n) return fail(FX_EINVAL, "bad argument");
    std::vector<uint8_t> padded((size_t)n + 16, 0);           // (the decoder looks 8 bytes past the positions it reads)
    memcpy(padded.data(), in, (size_t)n);
    pgz::Result res;
    bool small = false;
    const bool ok = pgz::inflate_parallel(padded.data(), (uint64_t)n, threads > 1 ? threads : 2, (uint64_t)GZ_SPACING,
        [&](uint64_t total, int) { *out_n = (int64_t)total; small = (int64_t)total > cap || !out; return !small; },
        [&](int, uint64_t off, const uint8_t *d, size_t len) { memcpy(out + off, d, len); return true; }, res);
    if (small) return fail(FX_ERANGE, "the stream inflates to %lld bytes", (long long)*out_n);
    if (!ok) return 1;
    if (n_points) *n_points = (int64_t)res.pt_cin.size();
    return FX_OK;
}

extern "C" int fx_gz_open_mode(const fx_handle *h) { return h ? h->gz_mode : 0; }

extern "C" int fx_bgzf_counts(fx_handle *h, int64_t out[3]) {
    if (!h || !out) return fail(FX_EINVAL, "null argument");
    out[0] = h->bgzf_members; out[1] = h->bgzf_handed_over; out[2] = h->bgzf_reason;
    return FX_OK;
}

extern "C" int fx_gz_checkpoints(fx_handle *h, int64_t cap, int64_t *cmp_off, int64_t *uncmp_off, uint8_t *bits, uint8_t *has_data,
                                 uint8_t *windows, int64_t *n_out, int64_t *n_windows) {
    if (!h || !n_out) return fail(FX_EINVAL, "null argument");
    const int64_t n = (int64_t)h->gzp_cin.size();
    *n_out = n;
    if (n_windows) *n_windows = (int64_t)(h->gzp_win.size() / GZ_WINDOW);
    if (cap < n || !cmp_off) return FX_OK;                    // a query for the counts
    if (!uncmp_off || !bits || !has_data) return fail(FX_EINVAL, "null argument");
    for (int64_t i = 0; i < n; ++i) { cmp_off[i] = h->gzp_cin[(size_t)i]; uncmp_off[i] = h->gzp_cout[(size_t)i]; bits[i] = h->gzp_bits[(size_t)i]; has_data[i] = h->gzp_has[(size_t)i]; }
    if (windows && !h->gzp_win.empty()) memcpy(windows, h->gzp_win.data(), h->gzp_win.size());
    return FX_OK;
}

extern "C" int fx_sync(fx_handle *h) {
    if (!h) return fail(FX_EINVAL, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    h->prof.drain();
    return FX_OK;
}

extern "C" int fx_prof_enable(fx_handle *h, int on) {
    if (!h) return fail(FX_EINVAL, "null handle");
    int rc = fx_sync(h);
    if (rc) return rc;
    h->prof.on = on != 0;
    h->prof.mask = (on == 2) ? (1u << K_SPAN_SCAN) : ~0u;   // 2: only the dominant kernel (2 events per build)
    return FX_OK;
}

extern "C" int fx_prof_default(int on) { g_prof_default = on != 0; return FX_OK; }

extern "C" int fx_prof_reset(fx_handle *h) {
    if (!h) return fail(FX_EINVAL, "null handle");
    int rc = fx_sync(h);
    if (rc) return rc;
    for (int i = 0; i < K_NKERN; ++i) { h->prof.ms[i] = 0; h->prof.cnt[i] = 0; }
    return FX_OK;
}

extern "C" int fx_prof_count(void) { return K_NKERN; }
extern "C" const char *fx_prof_name(int id) { return (id >= 0 && id < K_NKERN) ? kKernelNames[id] : ""; }

extern "C" int fx_prof_read(fx_handle *h, int id, double *total_ms, int64_t *launches) {
    if (!h || id < 0 || id >= K_NKERN) return fail(FX_EINVAL, "bad argument");
    int rc = fx_sync(h);
    if (rc) return rc;
    if (total_ms) *total_ms = h->prof.ms[id];
    if (launches) *launches = h->prof.cnt[id];
    return FX_OK;
}

static int shard_summary_launch(fx_handle *h, int64_t *d_out) {
    static_assert(sizeof(fx_shard_summary) == SS_WORDS * 8, "summary layout");
    if (!h->fasta_built) return fail(FX_ESTATE, "fx_fasta_build has not run");
    int rc = use_device(h);
    if (rc) return rc;
    hipLaunchKernelGGL(k_shard_summary2, dim3(1), dim3(SUMM_BLOCK), 0, h->stream, scan_ctx(h), (int)h->is_last, ctl_totals(h),
                       h->hdr.cap, h->hdr.p, h->fa_hdr_line.p, fasta_cols(h), d_out);
    HIPCHK(hipGetLastError());
    return FX_OK;
}

extern "C" int fx_shard_summary_get(fx_handle *h, fx_shard_summary *out) {
    if (!h || !out) return fail(FX_EINVAL, "null argument");
    int64_t *S = h->ctl.p ? (int64_t *)(h->ctl.p + 16) : nullptr;   // 28 words of the control block (no allocation per call)
    int rc = shard_summary_launch(h, S);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out, S, sizeof(fx_shard_summary), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return FX_OK;
}

extern "C" int fx_shard_summary_dev(fx_handle *h, int64_t *d_out) {
    if (!h || !d_out) return fail(FX_EINVAL, "null argument");
    return shard_summary_launch(h, d_out);
}

extern "C" int fx_fasta_stitch_dev(fx_handle *h, const int64_t *d_all, int world, int rank, int full_name) {
    if (!h || !d_all || world < 1 || rank < 0 || rank >= world) return fail(FX_EINVAL, "bad argument");
    if (!h->fasta_built) return fail(FX_ESTATE, "fx_fasta_build has not run");
    int rc = use_device(h);
    if (!rc) rc = finish_build(h);
    if (rc) return rc;
    hipLaunchKernelGGL(k_stitch_tail, dim3(1), dim3(1), 0, h->stream, d_all, world, rank, full_name, fasta_cols(h), h->hdr.cap,
                       (int *)(h->ctl.p + 60));
    HIPCHK(hipGetLastError());
    return FX_OK;
}

extern "C" void *fx_stream(fx_handle *h) { return h ? (void *)h->stream : nullptr; }

extern "C" int fx_fasta_line_regular(fx_handle *h, int where, int32_t *reg) {
    if (!h || !reg) return fail(FX_EINVAL, "null argument");
    if (!h->fasta_built) return fail(FX_ESTATE, "fx_fasta_build has not run");
    int rc = use_device(h);
    if (!rc) rc = finish_build(h);
    if (rc) return rc;
    if ((rc = copy_out(h, where, reg, h->fa_reg.p, h->n_hdr))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    return FX_OK;
}

extern "C" int fx_fasta_set_row(fx_handle *h, int64_t k, int64_t boff, int64_t blen, int64_t slen, int64_t llen,
                                int32_t elen, int32_t norm, int32_t dlen, int32_t name_len) {
    if (!h) return fail(FX_EINVAL, "null handle");
    if (!h->fasta_built) return fail(FX_ESTATE, "fx_fasta_build has not run");
    int rc = use_device(h);
    if (!rc) rc = finish_build(h);
    if (rc) return rc;
    if (k < 0 || k >= h->n_hdr) return fail(FX_ERANGE, "row %lld out of range", (long long)k);
    // norm: bit 0 = index.c's norm, bit 1 = line-regular (pyfastx_amd/shard.py: stitch_tail decides both from the summaries)
    hipLaunchKernelGGL(k_set_row, dim3(1), dim3(1), 0, h->stream, fasta_cols(h), k, boff, blen, slen, llen, elen, norm, dlen, name_len,
                       (int32_t)((norm >> 1) & 1));
    HIPCHK(hipGetLastError());
    return FX_OK;
}

// ------------------------------------------------------------- .fxi bulk load (host side)
extern "C" int fx_fxi_bulk_rows(const char *path, int rootpage, int64_t n, const uint8_t *names, const int64_t *name_off,
                                int ncols, const int64_t *const *cols) {
    // names == name_off == null: a table without a TEXT column (comp)
    if (!path || n < 0 || (n > 0 && ((name_off && !names && name_off[n] > 0) || (!name_off && names) || (ncols > 0 && !cols))))
        return fail(FX_EINVAL, "bad argument");
    const fxi::Rows r{n, names, name_off, ncols, cols};
    const int rc = fxi::bulk_load_table(path, (uint32_t)rootpage, r);
    if (rc == fxi::E_ROW) return fail(FX_ERANGE, "a row does not fit a b-tree page without overflow: use the INSERT path");
    if (rc == fxi::E_IO) return fail(FX_EIO, "cannot write %s", path);
    if (rc) return fail(FX_EINVAL, "%s is not a SQLite database this loader can extend", path);
    return FX_OK;
}

extern "C" int fx_fxi_bulk_index(const char *path, int rootpage, int64_t n, const uint8_t *names, const int64_t *name_off,
                                 const int64_t *order) {
    if (!path || n < 0 || (n > 0 && (!name_off || !order || (!names && name_off[n] > 0)))) return fail(FX_EINVAL, "bad argument");
    const fxi::Entries e{n, names, name_off, nullptr, order};
    const int rc = fxi::bulk_load_index(path, (uint32_t)rootpage, e);
    if (rc == fxi::E_ROW) return fail(FX_ERANGE, "an index entry does not fit a b-tree page without overflow: use CREATE INDEX");
    if (rc == fxi::E_IO) return fail(FX_EIO, "cannot write %s", path);
    if (rc) return fail(FX_EINVAL, "%s is not a SQLite database this loader can extend", path);
    return FX_OK;
}

extern "C" int fx_fxi_bulk_index_int(const char *path, int rootpage, int64_t n, const int64_t *key, const int64_t *order) {
    if (!path || n < 0 || (n > 0 && (!key || !order))) return fail(FX_EINVAL, "bad argument");
    const fxi::Entries e{n, nullptr, nullptr, key, order};
    const int rc = fxi::bulk_load_index(path, (uint32_t)rootpage, e);
    if (rc == fxi::E_IO) return fail(FX_EIO, "cannot write %s", path);
    if (rc) return fail(FX_EINVAL, "%s is not a SQLite database this loader can extend", path);
    return FX_OK;
}

// ------------------------------------------------------------------ the two big b-trees of a .fxi from the device (round 5)
// fx_fxi_dev_sort + fx_fxi_dev_write: what fx_names_pack + fx_names_sort + fx_fxi_bulk_rows + fx_fxi_bulk_index do through
// host arrays, with the pages formatted where the table and the names are (fx_fxi_dev.hpp).  Only finished pages cross
// PCIe: pinned 8 MiB pieces, several threads, each piece copied into the mapping of the file -- grown to its final size
// and allocated (fallocate) before the first store -- while the next one travels.
static int fxi_cols(fx_handle *h, int kind, FxiCols *c) {
    memset(c, 0, sizeof *c);
    c->gbase = h->base;
    if (kind == 1) {                                         // read: name, dlen, rlen, soff, qoff (fastq.c:29-36)
        c->p[0] = h->fq_dlen.p; c->w[0] = 4;
        c->p[1] = h->fq_rlen.p; c->w[1] = 8;
        c->p[2] = h->fq_soff.p; c->w[2] = 8;
        c->p[3] = h->fq_qoff.p; c->w[3] = 8;
        c->ncols = 4;
        c->name_off = h->fq_name_off.p; c->name_add = 0; c->name_len = h->fq_name_len.p;
    } else {                                                 // seq: chrom, boff, blen, slen, llen, elen, norm, dlen (index.c:178-189)
        c->p[0] = h->fa_boff.p; c->w[0] = 8;
        c->p[1] = h->fa_blen.p; c->w[1] = 8;
        c->p[2] = h->fa_slen.p; c->w[2] = 8;
        c->p[3] = h->fa_llen.p; c->w[3] = 8;
        c->p[4] = h->fa_elen.p; c->w[4] = 4;
        c->p[5] = h->fa_norm.p; c->w[5] = 4;
        c->p[6] = h->fa_dlen.p; c->w[6] = 4;
        c->ncols = 7;
        c->name_off = h->hdr.p; c->name_add = 1; c->name_len = h->fa_name_len.p;
    }
    return FX_OK;
}

static int fxi_copy_threads() {
    static const int n = [] {
        if (const char *e = getenv("FX_FXI_COPY_THREADS")) { const int v = atoi(e); if (v > 0) return std::min(v, 64); }
        const unsigned hw = std::thread::hardware_concurrency();
        return (int)std::min<unsigned>(16u, std::max<unsigned>(4u, hw / 4));
    }();
    return n;
}

// logical pages koff + [k0, k1) of a page sequence, FXI_PAGE bytes each and back to back at d_img, to their places in the file
static int fxi_image_out(int device, const uint8_t *d_img, int64_t k0, int64_t k1, const fxi::PageSeq &seq, int64_t koff, int fd, const fxi::FileMap &map) {
    const int64_t ppp = PIECE_BYTES / FXI_PAGE, npieces = (k1 - k0 + ppp - 1) / ppp;
    const int T = (int)std::min<int64_t>(fxi_copy_threads(), std::max<int64_t>(1, npieces));
    std::atomic<int> err(0);                                 // 1: device, 2: file
    std::vector<std::thread> th;
    auto pg = [&](int64_t k) { return (int64_t)seq.at((uint64_t)(koff + k)); };
    auto put = [&](const uint8_t *src, int64_t a, int64_t b) {          // pages [a, b) -- adjacent in the file unless the skipped page lies between
        int64_t cut = b;
        if (pg(b - 1) - pg(a) != b - 1 - a)
            for (cut = a + 1; cut < b && pg(cut) == pg(cut - 1) + 1;) ++cut;
        for (int part = 0; part < 2; ++part) {
            const int64_t x = part ? cut : a, y = part ? b : cut;
            if (x >= y) continue;
            const size_t off = (size_t)(pg(x) - 1) * FXI_PAGE, len = (size_t)(y - x) * FXI_PAGE;
            if (map.p && off + len <= map.len) map.put(off, src + (size_t)(x - a) * FXI_PAGE, len);
            else if (!fxi::pwrite_all(fd, src + (size_t)(x - a) * FXI_PAGE, len, (off_t)off)) err.store(2);
        }
    };
    cpu_set_t near_cpus;
    static const bool no_bind = [] { const char *e = getenv("FX_FXI_NO_BIND"); return e && atoi(e) != 0; }();
    static const bool trace_copy = [] { const char *e = getenv("FX_TRACE_FXI_COPY"); return e && atoi(e) != 0; }();   // per call: what the lanes waited for the device, what they spent storing into the file
    std::atomic<long long> wait_us(0), put_us(0), first_us(1ll << 60), last_us(0);
    const auto C0 = std::chrono::steady_clock::now();
    const bool bind = !no_bind && device_cpus(device, &near_cpus);     // the copy threads on the CPUs next to the device, as the staging threads are
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t]() {
            if (bind) (void)pthread_setaffinity_np(pthread_self(), sizeof near_cpus, &near_cpus);
            if (hipSetDevice(device) != hipSuccess) { err.store(1); return; }
            uint8_t *pin[2] = {g_pins.get(), g_pins.get()};
            hipStream_t st = nullptr;
            hipEvent_t ev[2] = {nullptr, nullptr};
            bool ok = pin[0] && pin[1] && (st = g_lane_streams.get(device)) != nullptr &&
                      hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) == hipSuccess;
            if (!ok) err.store(1);
            int64_t pa[2] = {-1, -1}, pb[2] = {0, 0};
            auto drain = [&](int sl) {
                if (pa[sl] < 0) return;
                const auto w0 = std::chrono::steady_clock::now();
                if (hipEventSynchronize(ev[sl]) != hipSuccess) { err.store(1); pa[sl] = -1; return; }
                const auto w1 = std::chrono::steady_clock::now();
                put(pin[sl], pa[sl], pb[sl]);
                if (trace_copy) {
                    const long long landed = (long long)std::chrono::duration<double, std::micro>(w1 - C0).count();
                    long long cur = first_us.load();
                    while (landed < cur && !first_us.compare_exchange_weak(cur, landed)) {}
                    cur = last_us.load();
                    while (landed > cur && !last_us.compare_exchange_weak(cur, landed)) {}
                    wait_us.fetch_add((long long)std::chrono::duration<double, std::micro>(w1 - w0).count());
                    put_us.fetch_add((long long)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w1).count());
                }
                pa[sl] = -1;
            };
            int slot = 0;
            for (int64_t pc = t; ok && pc < npieces && !err.load(); pc += T, slot ^= 1) {
                const int64_t a = k0 + pc * ppp, b = std::min(k1, a + ppp);
                drain(slot);
                if (hipMemcpyAsync(pin[slot], d_img + (size_t)(a - k0) * FXI_PAGE, (size_t)(b - a) * FXI_PAGE, hipMemcpyDeviceToHost, st) != hipSuccess ||
                    hipEventRecord(ev[slot], st) != hipSuccess) { err.store(1); break; }
                pa[slot] = a; pb[slot] = b;
                drain(slot ^ 1);
            }
            drain(0); drain(1);
            if (st) (void)hipStreamSynchronize(st);
            for (int i = 0; i < 2; ++i) { if (ev[i]) (void)hipEventDestroy(ev[i]); if (pin[i]) g_pins.put(pin[i]); }
            if (st) g_lane_streams.put(device, st);
        });
    for (auto &x : th) x.join();
    if (trace_copy)
        fprintf(stderr, "[fxgpu] fxi_image_out %.2f GB, %d lanes: %.1f ms; per lane waiting for the device %.1f ms, storing into the file %.1f ms; first piece seen on the host after %.1f ms, last after %.1f ms\n", (double)(k1 - k0) * FXI_PAGE / 1e9, T,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - C0).count(), wait_us.load() / 1e3 / T, put_us.load() / 1e3 / T, first_us.load() / 1e3, last_us.load() / 1e3);
    if (err.load() == 1) return fail(FX_EDEVICE, "device to host copy of index pages failed");
    if (err.load() == 2) return fail(FX_EIO, "cannot write the index file");
    return FX_OK;
}

static int64_t fxi_slab_pages() {                          // pages formatted per kernel launch (HBM the image takes): FX_FXI_SLAB_MB, default 4096
    const char *e = getenv("FX_FXI_SLAB_MB");
    const int64_t mb = e ? atoll(e) : 0;
    return ((mb > 0 ? mb : 4096) << 20) / FXI_PAGE;
}

static int fxi_check_kind(fx_handle *h, int kind) {
    if (!h || (kind != 0 && kind != 1)) return fail(FX_EINVAL, "bad argument");
    if (kind == 0 ? !h->fasta_built : !h->fastq_built) return fail(FX_ESTATE, "the index has not been built");
    if (kind == 0 && !h->hdr.p) return fail(FX_ESTATE, "names need a scanned index (fx_fasta_build), not an installed table");
    int rc = use_device(h);
    if (!rc) rc = finish_build(h);
    return rc;
}

// What the kernels of fx_fxi_dev.hpp need, whoever owns the rows and the names: one handle's table and stream
// (fx_fxi_dev_write), one part of a table that several handles share (fx_fxi_part_*), or the packed names of all parts on
// the device that writes the index (fx_fxi_join_*).
struct FxiJob {
    int device = 0;
    hipStream_t stream = nullptr;
    FxiCols c;
    const uint8_t *data = nullptr;                         // what c.name_off points into
    int64_t n = 0;                                           // rows / entries
    const int64_t *order = nullptr;                          // the index: e-th smallest name = row order[e]
    // (all out of the scratch pool: the hipFree of these blocks at the end of the call -- 1 GB -- cost 0.12-0.23 s of waiting for the device)
    ScratchBuf<uint16_t> sz;
    ScratchBuf<int32_t> pages, bad;
    ScratchBuf<int64_t> sums, pbase;
    ScratchBuf<uint8_t> slab;
    int64_t nchunks = 0, nsc = 0;
    int init() {
        nchunks = (n + FXI_R - 1) / FXI_R;
        nsc = (nchunks + SCAN_CHUNK - 1) / SCAN_CHUNK;
        int rc;
        if ((rc = sz.alloc(device, n, stream)) || (rc = pages.alloc(device, nchunks, stream)) || (rc = bad.alloc(device, 1, stream)) ||
            (rc = sums.alloc(device, nsc + 1, stream)) || (rc = pbase.alloc(device, nchunks + 1, stream))) return rc;
        HIPCHK(hipMemsetAsync(bad.p, 0, 4, stream));
        return FX_OK;
    }
    void table_sizes() { hipLaunchKernelGGL(k_fxi_cell_sizes, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, stream, c, n, sz.p, bad.p); }
    void index_sizes() { hipLaunchKernelGGL(k_fxi_entry_sizes, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, stream, c, order, n, sz.p, bad.p); }
    // shape of one tree's leaf level: sizes are in sz -> nleaf, first[0 .. nleaf]
    int leaf_level(bool idx, ScratchBuf<int64_t> &first, int64_t *nleaf_out) {
        if (idx) hipLaunchKernelGGL((k_fxi_fill<true, false>), dim3((unsigned)nchunks), dim3(64), 0, stream, sz.p, n, FXI_PAGE - 8, pages.p, (const int64_t *)nullptr, (int64_t *)nullptr);
        else hipLaunchKernelGGL((k_fxi_fill<false, false>), dim3((unsigned)nchunks), dim3(64), 0, stream, sz.p, n, FXI_PAGE - 8, pages.p, (const int64_t *)nullptr, (int64_t *)nullptr);
        hipLaunchKernelGGL(k_cnt_chunk_sums, dim3((unsigned)nsc), dim3(BLOCK), 0, stream, pages.p, nchunks, sums.p);
        hipLaunchKernelGGL(k_cnt_chunk_bases, dim3(1), dim3(BLOCK), 0, stream, sums.p, nsc);
        hipLaunchKernelGGL(k_cnt_offsets, dim3((unsigned)nsc), dim3(BLOCK), 0, stream, pages.p, nchunks, sums.p, pbase.p);
        HIPCHK(hipGetLastError());
        int64_t nleaf = 0;
        int isbad = 0;
        HIPCHK(hipMemcpyAsync(&nleaf, pbase.p + nchunks, 8, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(&isbad, bad.p, 4, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (isbad) return fail(FX_ERANGE, idx ? "an index entry does not fit a b-tree page without overflow: use CREATE INDEX"
                                              : "a row does not fit a b-tree page without overflow: use the INSERT path");
        int r2 = first.alloc(device, nleaf + 1, stream);
        if (r2) return r2;
        if (idx) hipLaunchKernelGGL((k_fxi_fill<true, true>), dim3((unsigned)nchunks), dim3(64), 0, stream, sz.p, n, FXI_PAGE - 8, (int32_t *)nullptr, pbase.p, first.p);
        else hipLaunchKernelGGL((k_fxi_fill<false, true>), dim3((unsigned)nchunks), dim3(64), 0, stream, sz.p, n, FXI_PAGE - 8, (int32_t *)nullptr, pbase.p, first.p);
        HIPCHK(hipGetLastError());
        *nleaf_out = nleaf;
        return FX_OK;
    }
    // the leaves [0, nleaf) of one tree to the pages koff + [0, nleaf) of `seq`, slab by slab
    int leaves_out(bool idx, int64_t nleaf, const int64_t *first, const fxi::PageSeq &seq, int64_t koff, int fd, const fxi::FileMap &map, double *t_kern, double *t_copy) {
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        const int64_t S = std::min(nleaf, fxi_slab_pages());
        int r2 = slab.alloc(device, S * FXI_PAGE, stream);
        if (r2) return r2;
        for (int64_t k0 = 0; k0 < nleaf; k0 += S) {
            const int64_t k1 = std::min(nleaf, k0 + S);
            const auto t0 = now();
            const unsigned grid = (unsigned)std::min<int64_t>((k1 - k0 + 3) / 4, 16384);
            if (idx) hipLaunchKernelGGL(k_fxi_index_leaves, dim3(grid), dim3(BLOCK), 0, stream, c, data, order, first, nleaf, n, k0, k1, slab.p);
            else hipLaunchKernelGGL(k_fxi_table_leaves, dim3(grid), dim3(BLOCK), 0, stream, c, data, first, k0, k1, slab.p);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(stream));
            const auto t1 = now();
            if ((r2 = fxi_image_out(device, slab.p, k0, k1, seq, koff, fd, map))) return r2;
            *t_kern += secs(t0, t1); *t_copy += secs(t1, now());
        }
        return FX_OK;
    }
    // a tree of ONE leaf lives in its root page
    int root_leaf(bool idx, const int64_t *first, int fd, int rootpage, const char *path) {
        int r2 = slab.alloc(device, FXI_PAGE, stream);
        if (r2) return r2;
        if (idx) hipLaunchKernelGGL(k_fxi_index_leaves, dim3(1), dim3(BLOCK), 0, stream, c, data, order, first, (int64_t)1, n, (int64_t)0, (int64_t)1, slab.p);
        else hipLaunchKernelGGL(k_fxi_table_leaves, dim3(1), dim3(BLOCK), 0, stream, c, data, first, (int64_t)0, (int64_t)1, slab.p);
        std::vector<uint8_t> pg(FXI_PAGE);
        HIPCHK(hipMemcpyAsync(pg.data(), slab.p, FXI_PAGE, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (!fxi::pwrite_all(fd, pg.data(), FXI_PAGE, (off_t)(rootpage - 1) * FXI_PAGE)) return fail(FX_EIO, "cannot write %s", path);
        return FX_OK;
    }
    // the dividers of the index -- (name, rowid) of the entry between leaf d and leaf d + 1 -- for the levels the host writes
    int dividers(const int64_t *first_i, int64_t nd, std::vector<int64_t> &d_rowid, std::vector<int64_t> &d_off, std::vector<uint8_t> &d_names) {
        d_rowid.assign((size_t)std::max<int64_t>(nd, 1), 0);
        d_off.assign((size_t)nd + 1, 0);
        d_names.assign(1, 0);
        if (nd <= 0) return FX_OK;
        int rc;
        ScratchBuf<int64_t> drow, doff, dsum;
        ScratchBuf<int32_t> dlen;
        ScratchBuf<uint8_t> dnm;
        const int64_t ndc = (nd + SCAN_CHUNK - 1) / SCAN_CHUNK;
        if ((rc = drow.alloc(device, nd, stream)) || (rc = dlen.alloc(device, nd, stream)) || (rc = doff.alloc(device, nd + 1, stream)) ||
            (rc = dsum.alloc(device, ndc + 1, stream))) return rc;
        hipLaunchKernelGGL(k_fxi_divider_rows, dim3(nblocks(nd, BLOCK)), dim3(BLOCK), 0, stream, c, order, first_i, nd, drow.p, dlen.p);
        hipLaunchKernelGGL(k_cnt_chunk_sums, dim3((unsigned)ndc), dim3(BLOCK), 0, stream, dlen.p, nd, dsum.p);
        hipLaunchKernelGGL(k_cnt_chunk_bases, dim3(1), dim3(BLOCK), 0, stream, dsum.p, ndc);
        hipLaunchKernelGGL(k_cnt_offsets, dim3((unsigned)ndc), dim3(BLOCK), 0, stream, dlen.p, nd, dsum.p, doff.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(d_off.data(), doff.p, (size_t)(nd + 1) * 8, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(d_rowid.data(), drow.p, (size_t)nd * 8, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        const int64_t tot = d_off[(size_t)nd];
        d_names.resize((size_t)std::max<int64_t>(tot, 1));
        if (tot) {
            if ((rc = dnm.alloc(device, tot, stream))) return rc;
            hipLaunchKernelGGL(k_fxi_divider_names, dim3(nblocks(nd, BLOCK)), dim3(BLOCK), 0, stream, c, data, (const int64_t *)drow.p, (const int64_t *)doff.p, nd, dnm.p);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(d_names.data(), dnm.p, (size_t)tot, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
        }
        for (auto &r : d_rowid) r += c.row_base + 1;         // row -> rowid
        return FX_OK;
    }
};

// The pages of a database that exist + what is to come, mapped; grown and allocated first (see below).  -> map.p null: pwrite.
static void fxi_grow_and_map(fxi::DbFile &db, uint32_t new_npages, int device, fxi::FileMap &map) {
    // All new pages are allocated BEFORE anything is stored into them: on tmpfs fallocate alone runs at 16-18 GB/s and
    // sixteen threads then copy into the mapping at 15 GB/s, while page allocation and faults running side by side
    // (a fallocate thread ahead of the writers, or first touches through the mapping) reach 3-4 GB/s together
    // (tools/filewrite_probe2.c: 10 GB in 1.3 s against 2.6-3.8 s).  A file that already has the room (pre-sized by the
    // caller while the stream was staged) skips this.
    const off_t end = (off_t)new_npages * FXI_PAGE, from = (off_t)db.npages * FXI_PAGE;
    struct stat st;
    const off_t size_now = fstat(db.fd, &st) == 0 ? st.st_size : db.size0;     // (other parts may have grown the file since it was opened)
    const bool presized = size_now >= end;
    static const bool by_pwrite = [] { const char *e = getenv("FX_FXI_PWRITE"); return e && atoi(e) != 0; }();
    if (by_pwrite) {
        // (experiment, round 6) no mapping at all: the pages go into the file with pwrite -- the kernel copies into the page cache,
        // no page-table entries are made for 2.5 M pages and none have to be taken down again (the munmap of a 10 GB mapping holds
        // the process's mm lock for ~0.2 s, whoever needs it next waits: measured as a 214 ms hipStreamDestroy).  The room is
        // allocated here all the same.
        const off_t have = std::max(from, size_now & ~(off_t)(FXI_PAGE - 1));
        if (!presized && have < end && !getenv("FX_FXI_NO_FALLOCATE")) (void)fallocate(db.fd, 0, have, end - have);
        return;
    }
    if (presized || map.open(db.fd, (size_t)end)) {
        if (presized) (void)map.map_existing(db.fd, (size_t)end);
        // (only what a pre-sized file lacks: fallocate over pages that exist still visits every one of them, 0.2 us each)
        const off_t have = std::max(from, size_now & ~(off_t)(FXI_PAGE - 1));
        if (map.p && !presized && have < end && !getenv("FX_FXI_NO_FALLOCATE")) {
            // (in a thread on the CPUs next to the device: the pages then lie in the memory its copy threads are next to)
            cpu_set_t near_cpus;
            const bool bind = !getenv("FX_FXI_NO_BIND") && device_cpus(device, &near_cpus);
            const int fd_ = db.fd;
            int fa_errno = 0;
            std::thread([fd_, have, end, bind, near_cpus, &fa_errno]() {
                if (bind) (void)pthread_setaffinity_np(pthread_self(), sizeof near_cpus, &near_cpus);
                if (fallocate(fd_, 0, have, end - have) != 0) fa_errno = errno;
            }).join();
            // no room after all (a quota, a race with another writer): a store into the mapping would be a SIGBUS where a
            // pwrite returns an error -- the pages go through pwrite then.  (EOPNOTSUPP and the like: the mapping stays.)
            if (fa_errno == ENOSPC || fa_errno == EDQUOT || fa_errno == EFBIG) map.close();
        }
    }
}
// Taking a mapping down walks every page table entry of it (0.25 s for 10 GB of dirty shared pages) and nothing waits
// for the result: the pages are in the file's page cache either way.  It is left to a thread of its own -- which takes the
// mapping down PIECE BY PIECE: a munmap holds the process's mm lock for as long as it runs, and behind one call for the whole
// 10 GB everything else that needs the lock waited up to 0.2 s -- the free() of a large vector in this function's own epilogue,
// SQLite's next open, a hipStreamDestroy (FX_TRACE: the call returned 0-130 ms after its last lap).
static void fxi_unmap_later(fxi::FileMap &map) {
    if (map.p && !getenv("FX_FXI_SYNC_UNMAP")) {
        uint8_t *mp = map.p;
        const size_t area = map.area, len = map.len, C = map.one ? fxi::FileMap::MIN_CHUNK : map.chunk, S = map.one ? C : map.stride();
        const bool one = map.one;
        map.p = nullptr;
        std::thread([mp, area, len, one, C, S]() {
            // (every piece with the guard behind it, each address once: a range that has been unmapped may belong to somebody
            // else a moment later -- a second munmap over the whole area took a numpy array of the caller with it)
            static const int pause_us = [] { const char *e = getenv("FX_FXI_UNMAP_PAUSE_US"); return e ? atoi(e) : 0; }();   // (tests: a caller that maps memory meanwhile)
            if (one) { for (size_t o = 0; o < len; o += C) { (void)munmap(mp + o, std::min(C, len - o)); if (pause_us > 0) usleep((useconds_t)pause_us); } }
            else for (size_t o = 0; o < area; o += S) { (void)munmap(mp + o, std::min(S, area - o)); if (pause_us > 0) usleep((useconds_t)pause_us); }
        }).detach();
    }
    map.close();
}

extern "C" int fx_fxi_dev_sort(fx_handle *h, int kind, int64_t *n_dup) {
    int rc = fxi_check_kind(h, kind);
    if (rc) return rc;
    if (!n_dup) return fail(FX_EINVAL, "null n_dup");
    *n_dup = 0;
    h->fxi_order_kind = -1;
    const int64_t n = kind == 0 ? h->n_hdr : h->n_reads;
    if (n == 0) { h->fxi_order_kind = kind; h->fxi_order_n = 0; return FX_OK; }
    if (n >= 0xFFFFFFFFll) return fail(FX_ERANGE, "too many records for the 32-bit sort index");
    FxiCols c;
    fxi_cols(h, kind, &c);
    const int64_t *noff = c.name_off;
    if (kind == 0) {
        if ((rc = h->nm_off.alloc(n))) return rc;
        hipLaunchKernelGGL(k_add_i64, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, h->stream, h->hdr.p, (int64_t)1, n, h->nm_off.p);
        noff = h->nm_off.p;
    }
    if ((rc = h->fxi_order.alloc(h->device, n, h->stream)) || (rc = h->fxi_soff.alloc(h->device, n, h->stream)) || (rc = h->fxi_slen.alloc(h->device, n, h->stream))) return rc;
    DevBuf<int64_t> d_ndup;
    if ((rc = d_ndup.alloc(1))) return rc;
    const char *what = "";
    const int e = sort_names(h->d_data, h->base, noff, c.name_len, n, h->fxi_order.p, d_ndup.p, h->stream, &what, h->fxi_soff.p, h->fxi_slen.p);
    if (e) return fail(e == (int)hipErrorOutOfMemory ? FX_ENOMEM : FX_EDEVICE, "name sort, %s: %s", what, hipGetErrorString((hipError_t)e));
    HIPCHK(hipMemcpyAsync(n_dup, d_ndup.p, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->fxi_order_kind = kind;
    h->fxi_order_n = n;
    return FX_OK;
}

// laps[8]: [0] table shape, [1] table leaf kernels, [2] table leaves to the file, [3] file grown and allocated (fallocate),
// [4] index shape + dividers, [5] index leaf kernels, [6] index leaves to the file, [7] rest of the host levels + header
extern "C" int fx_fxi_dev_write(fx_handle *h, int kind, const char *path, int root_table, int root_index, double *laps) {
    int rc = fxi_check_kind(h, kind);
    if (rc) return rc;
    if (!path || root_table < 2 || (root_index != 0 && root_index < 2)) return fail(FX_EINVAL, "bad argument");
    const int64_t n = kind == 0 ? h->n_hdr : h->n_reads;
    if (root_index && (h->fxi_order_kind != kind || h->fxi_order_n != n)) return fail(FX_ESTATE, "fx_fxi_dev_sort has not been called for this table");
    if (n >= 0xFFFFFFFFll) return fail(FX_ERANGE, "too many records");
    double lap_buf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    static const bool trace = [] { const char *e = getenv("FX_TRACE"); return e && atoi(e) != 0; }();
    FxiJob J;
    J.device = h->device; J.stream = h->stream; J.data = h->d_data; J.n = n; J.order = h->fxi_order.p;
    fxi_cols(h, kind, &J.c);
    J.c.s_off = h->fxi_soff.p; J.c.s_len = h->fxi_slen.p;   // (offsets as the sort saw them: the FASTA '+ 1' is in them)
    ScratchBuf<int64_t> first_t, first_i;
    auto done = [&](int code) {                               // every way out: the order goes back to the pool, the laps to the caller
        (void)hipStreamSynchronize(h->stream);
        h->fxi_order.release(); h->fxi_soff.release(); h->fxi_slen.release(); h->fxi_order_kind = -1;
        if (laps) memcpy(laps, lap_buf, sizeof lap_buf);
        return code;
    };
#define FXI_CHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return done(fail(FX_EDEVICE, "%s: %s", #expr, hipGetErrorString(e_))); } while (0)
    if (n == 0) return done(FX_OK);
    if ((rc = J.init())) return done(rc);

    const auto t0 = now();
    fxi::DbFile db;
    {
        const int e = db.open_rw(path, (uint32_t)root_table);
        if (e == fxi::E_IO) return done(fail(FX_EIO, "cannot open %s", path));
        if (e || db.pagesize != FXI_PAGE || db.usable != FXI_PAGE || (root_index && (uint32_t)root_index > db.npages))
            return done(fail(FX_EINVAL, "%s is not a SQLite database this loader can extend (4 KiB pages, no reserved bytes)", path));
    }
    // ================================================================ shapes: which row on which leaf of the table, which entry on which leaf of the index
    int64_t nleaf_t = 0, nleaf_i = 0;
    J.table_sizes();
    if ((rc = J.leaf_level(false, first_t, &nleaf_t))) return done(rc);
    std::vector<int64_t> lf((size_t)nleaf_t + 1);
    FXI_CHK(hipMemcpyAsync(lf.data(), first_t.p, (size_t)(nleaf_t + 1) * 8, hipMemcpyDeviceToHost, h->stream));
    FXI_CHK(hipStreamSynchronize(h->stream));
    const auto t1 = now();
    lap_buf[0] = secs(t0, t1);
    std::vector<int64_t> d_rowid(1), d_off(1, 0);
    std::vector<uint8_t> d_names(1);
    fxi::IndexUpper up;
    int64_t nd = 0;
    if (root_index) {
        J.index_sizes();
        if ((rc = J.leaf_level(true, first_i, &nleaf_i))) return done(rc);
        nd = nleaf_i - 1;
        if ((rc = J.dividers(first_i.p, nd, d_rowid, d_off, d_names))) return done(rc);
    }
    const fxi::Entries dv{nd, d_names.data(), d_off.data(), nullptr, nullptr, d_rowid.data()};
    if (root_index && !up.plan((size_t)nleaf_i, dv, FXI_PAGE)) return done(fail(FX_ERANGE, "an index entry does not fit an interior page: use CREATE INDEX"));
    const auto t2 = now();
    lap_buf[4] = secs(t1, t2);

    // ================================================================ the page sequence: table leaves, table interior levels, index leaves, index upper levels
    const uint64_t tot_t = nleaf_t > 1 ? fxi::table_new_pages((size_t)nleaf_t, fxi::table_fan(FXI_PAGE)) : 0;
    const uint64_t tot_i = nleaf_i > 1 ? (uint64_t)nleaf_i + up.pages : 0;
    const uint64_t total = tot_t + tot_i;
    const fxi::PageSeq seq(db.npages + 1, FXI_PAGE);
    if (total && (uint64_t)seq.at(total - 1) >= 0xFFFFFFF0ull) return done(fail(FX_ERANGE, "the index file would exceed 2^32 pages"));
    bool ok = true;
    fxi::FileMap map;
    uint32_t new_npages = db.npages;
    if (total) {
        new_npages = seq.at(total - 1);
        fxi_grow_and_map(db, new_npages, h->device, map);
    }
    const auto t3 = now();
    lap_buf[3] = secs(t2, t3);                               // file grown and allocated
    // the host's share -- interior levels of the table, upper levels of the index -- in two threads beside the copy-out
    const fxi::PageSeq seq_i(total && tot_t ? seq.at(tot_t) : db.npages + 1, FXI_PAGE);     // (a sequence that starts behind the table's pages skips the same page)
    std::atomic<int> host_bad(0);
    std::thread th_t, th_i;
    if (nleaf_t > 1)
        th_t = std::thread([&]() { if (!fxi::table_interior(db.fd, map, FXI_PAGE, FXI_PAGE, (uint32_t)root_table, seq, lf.data(), (size_t)nleaf_t, tot_t)) host_bad.store(1); });
    if (nleaf_i > 1)
        th_i = std::thread([&]() { if (!up.write(db.fd, map, FXI_PAGE, FXI_PAGE, (uint32_t)root_index, seq_i, (size_t)nleaf_i, dv)) host_bad.store(1); });
    if (nleaf_t == 1) rc = J.root_leaf(false, first_t.p, db.fd, root_table, path);
    else rc = J.leaves_out(false, nleaf_t, first_t.p, seq, 0, db.fd, map, &lap_buf[1], &lap_buf[2]);
    if (!rc && root_index) {
        if (nleaf_i == 1) rc = J.root_leaf(true, first_i.p, db.fd, root_index, path);
        else rc = J.leaves_out(true, nleaf_i, first_i.p, seq_i, 0, db.fd, map, &lap_buf[5], &lap_buf[6]);
    }
    const auto t4 = now();
    if (th_t.joinable()) th_t.join();
    if (th_i.joinable()) th_i.join();
    if (host_bad.load()) ok = false;
    const auto t5 = now();
    if (!rc && ok) ok = db.finish(new_npages);
    if (!rc && ok && db.size0 > (off_t)new_npages * FXI_PAGE) ok = ftruncate(db.fd, (off_t)new_npages * FXI_PAGE) == 0;     // a pre-sized file: cut to what was used
    fxi_unmap_later(map);                                    // (behind the cut: an ftruncate waits for a munmap of the file's pages that is under way)
    const auto t6 = now();
    if (trace) fprintf(stderr, "[fxgpu] fxi: host levels joined after %.1f ms, header + cut + mapping handed off in %.1f ms\n", secs(t4, t5) * 1e3, secs(t5, t6) * 1e3);
    if (rc || !ok) db.give_back();
    lap_buf[7] = secs(t4, now());
    if (rc) return done(rc);
    if (!ok) return done(fail(FX_EIO, "cannot write %s", path));
    if (trace) fprintf(stderr, "[fxgpu] fxi pages from the device: %lld rows, %lld + %lld leaves, %llu pages: table shape %.1f ms, index shape + dividers %.1f ms, file grown %.1f ms, "
                               "table kernels %.1f + copy-out %.1f ms, index kernels %.1f + copy-out %.1f ms, rest of the host levels + header %.1f ms\n",
                       (long long)n, (long long)nleaf_t, (long long)nleaf_i, (unsigned long long)total, lap_buf[0] * 1e3, lap_buf[4] * 1e3, lap_buf[3] * 1e3,
                       lap_buf[1] * 1e3, lap_buf[2] * 1e3, lap_buf[5] * 1e3, lap_buf[6] * 1e3, lap_buf[7] * 1e3);
    return done(FX_OK);
#undef FXI_CHK
}

// fx_fxi_dev_sort + fx_fxi_dev_write in ONE call, with the sort and the shape of the index computed BESIDE the copy-out of the
// table's leaves (round 6): the name sort (40 ms for 10^8 reads), the entry sizes, the fill and the dividers (17 ms) run on a
// second stream in a second thread while the table's 6 GB of pages cross PCIe -- the device is idle then, the link is not.
// The schema must hold the empty UNIQUE INDEX already (root_index): whether the names are distinct is only known when the
// sort is done -- *n_dup > 0: no index was written, the caller drops the empty one (fastq.c:152-156 / index.c:363-366 ignore the
// failure of CREATE UNIQUE INDEX).  laps[8] as fx_fxi_dev_write, [4] = what of sort + index shape was NOT hidden.
extern "C" int fx_fxi_dev_build(fx_handle *h, int kind, const char *path, int root_table, int root_index, int64_t *n_dup, double *laps) {
    int rc = fxi_check_kind(h, kind);
    if (rc) return rc;
    if (!path || !n_dup || root_table < 2 || root_index < 2) return fail(FX_EINVAL, "bad argument");
    *n_dup = 0;
    const int64_t n = kind == 0 ? h->n_hdr : h->n_reads;
    if (n >= 0xFFFFFFFFll) return fail(FX_ERANGE, "too many records");
    double lap_buf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    auto done = [&](int code) { (void)hipStreamSynchronize(h->stream); if (laps) memcpy(laps, lap_buf, sizeof lap_buf); return code; };
    static const bool trace = [] { const char *e = getenv("FX_TRACE"); return e && atoi(e) != 0; }();
    const auto T00 = now();
    auto mark = [&](const char *what) { if (trace) fprintf(stderr, "[fxgpu] fxi build %-34s %8.1f ms\n", what, secs(T00, now()) * 1e3); };
    if (n == 0) return done(FX_OK);
    FxiJob T;                                                // the table
    T.device = h->device; T.stream = h->stream; T.data = h->d_data; T.n = n;
    fxi_cols(h, kind, &T.c);
    if ((rc = T.init())) return done(rc);
    const auto t0 = now();
    fxi::DbFile db;
    {
        const int e = db.open_rw(path, (uint32_t)root_table);
        if (e == fxi::E_IO) return done(fail(FX_EIO, "cannot open %s", path));
        if (e || db.pagesize != FXI_PAGE || db.usable != FXI_PAGE || (uint32_t)root_index > db.npages)
            return done(fail(FX_EINVAL, "%s is not a SQLite database this loader can extend (4 KiB pages, no reserved bytes)", path));
    }
    // ---- the other thread: order of the names, shape of the index, dividers
    FxiJob I;
    I.device = h->device; I.data = h->d_data; I.n = n;
    fxi_cols(h, kind, &I.c);
    ScratchBuf<int64_t> order, soff, first_i, noff_fa;
    ScratchBuf<int32_t> slen;
    int64_t ndup = 0, nleaf_i = 0, nd = 0;
    std::vector<int64_t> d_rowid(1), d_off(1, 0);
    std::vector<uint8_t> d_names(1);
    fxi::IndexUpper up;
    int rc_i = FX_OK;
    if (!h->stream2) HIPCHK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    const hipStream_t s2 = h->stream2;
    I.stream = s2;
    double t_side = 0;
    fxi::FileMap map, map2;                                  // (the table's pages; the index's, mapped anew once their number is known)
    std::atomic<long long> tot_t_pub(-1);                    // pages of the table for the other thread: -1 not known yet, -2 never will be
    std::thread side([&]() {
        const auto a = now();
        auto work = [&]() -> int {
            if (hipSetDevice(h->device) != hipSuccess) return fail(FX_EDEVICE, "hipSetDevice failed");
            int r;
            const int64_t *noff = I.c.name_off;
            if (kind == 0) {                                 // FASTA: the names begin one byte behind the header offsets
                if ((r = noff_fa.alloc(h->device, n, s2))) return r;
                hipLaunchKernelGGL(k_add_i64, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, s2, h->hdr.p, (int64_t)1, n, noff_fa.p);
                noff = noff_fa.p;
            }
            ScratchBuf<int64_t> d_nd;
            if ((r = order.alloc(h->device, n, s2)) || (r = soff.alloc(h->device, n, s2)) || (r = slen.alloc(h->device, n, s2)) || (r = d_nd.alloc(h->device, 1, s2))) return r;
            const char *what = "";
            const int e = sort_names(h->d_data, h->base, noff, I.c.name_len, n, order.p, d_nd.p, s2, &what, soff.p, slen.p);
            if (e) return fail(e == (int)hipErrorOutOfMemory ? FX_ENOMEM : FX_EDEVICE, "name sort, %s: %s", what, hipGetErrorString((hipError_t)e));
            if (hipMemcpyAsync(&ndup, d_nd.p, 8, hipMemcpyDeviceToHost, s2) != hipSuccess || hipStreamSynchronize(s2) != hipSuccess) return fail(FX_EDEVICE, "the name sort failed");
            if (ndup) return FX_OK;
            I.order = order.p; I.c.s_off = soff.p; I.c.s_len = slen.p;
            if ((r = I.init())) return r;
            I.index_sizes();
            if ((r = I.leaf_level(true, first_i, &nleaf_i))) return r;
            nd = nleaf_i - 1;
            if ((r = I.dividers(first_i.p, nd, d_rowid, d_off, d_names))) return r;
            const fxi::Entries dv{nd, d_names.data(), d_off.data(), nullptr, nullptr, d_rowid.data()};
            if (!up.plan((size_t)nleaf_i, dv, FXI_PAGE)) return fail(FX_ERANGE, "an index entry does not fit an interior page: use CREATE INDEX");
            // the index's pages mapped HERE, beside the table's copy-out, when the file has them already (room set aside while the
            // input was staged: nothing to allocate -- an allocation beside the stores of the copy lanes would slow both): 12 ms
            // of C3's constructor that stood between the table and the index
            if (nleaf_i > 1) {
                long long tt;
                while ((tt = tot_t_pub.load()) == -1) usleep(50);
                if (tt >= 0) {
                    const fxi::PageSeq sq(db.npages + 1, FXI_PAGE);
                    const uint64_t last = (uint64_t)tt + (uint64_t)nleaf_i + up.pages - 1;
                    struct stat st;
                    if ((uint64_t)sq.at(last) < 0xFFFFFFF0ull && fstat(db.fd, &st) == 0 && st.st_size >= (off_t)sq.at(last) * FXI_PAGE)
                        (void)map2.map_existing(db.fd, (size_t)sq.at(last) * FXI_PAGE);
                    if (trace) fprintf(stderr, "[fxgpu] fxi build (other thread) index room %s: file %lld bytes, needed %lld, after %.1f ms\n", map2.p ? "mapped" : "left to the first thread",
                                       (long long)st.st_size, (long long)sq.at(last) * FXI_PAGE, secs(T00, now()) * 1e3);
                }
            }
            return FX_OK;
        };
        rc_i = work();
        t_side = secs(a, now());
    });
    auto join_side = [&]() { long long e = -1; (void)tot_t_pub.compare_exchange_strong(e, -2); if (side.joinable()) side.join(); };
    auto bail = [&](int code) { join_side(); (void)hipStreamSynchronize(s2); return done(code); };
    // ---- this thread: the table
    ScratchBuf<int64_t> first_t;
    int64_t nleaf_t = 0;
    T.table_sizes();
    if ((rc = T.leaf_level(false, first_t, &nleaf_t))) return bail(rc);
    std::vector<int64_t> lf((size_t)nleaf_t + 1);
    if (hipMemcpyAsync(lf.data(), first_t.p, (size_t)(nleaf_t + 1) * 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
        return bail(fail(FX_EDEVICE, "the first rows of the table's leaves"));
    const auto t1 = now();
    lap_buf[0] = secs(t0, t1);
    mark("table shape");
    const uint64_t tot_t = nleaf_t > 1 ? fxi::table_new_pages((size_t)nleaf_t, fxi::table_fan(FXI_PAGE)) : 0;
    const fxi::PageSeq seq(db.npages + 1, FXI_PAGE);
    if (tot_t && (uint64_t)seq.at(tot_t - 1) >= 0xFFFFFFF0ull) return bail(fail(FX_ERANGE, "the index file would exceed 2^32 pages"));
    tot_t_pub.store((long long)tot_t);
    uint32_t new_npages = db.npages;
    if (tot_t) { new_npages = seq.at(tot_t - 1); fxi_grow_and_map(db, new_npages, h->device, map); }
    const auto t2 = now();
    lap_buf[3] = secs(t1, t2);
    mark("room for the table");
    std::atomic<int> host_bad(0);
    std::thread th_t;
    if (nleaf_t > 1)
        th_t = std::thread([&]() { if (!fxi::table_interior(db.fd, map, FXI_PAGE, FXI_PAGE, (uint32_t)root_table, seq, lf.data(), (size_t)nleaf_t, tot_t)) host_bad.store(1); });
    if (nleaf_t == 1) rc = T.root_leaf(false, first_t.p, db.fd, root_table, path);
    else rc = T.leaves_out(false, nleaf_t, first_t.p, seq, 0, db.fd, map, &lap_buf[1], &lap_buf[2]);
    mark("table leaves out");
    if (th_t.joinable()) th_t.join();
    mark("table interior joined");
    const auto t3 = now();
    join_side();
    lap_buf[4] = secs(t3, now());                            // what of the sort and the index shape the table's copy-out did not hide
    if (!rc) rc = rc_i;
    bool ok = !host_bad.load();
    // ---- the index, if the names are distinct
    *n_dup = ndup;
    if (!rc && ok && !ndup) {
        const uint64_t tot_i = nleaf_i > 1 ? (uint64_t)nleaf_i + up.pages : 0;
        const fxi::Entries dv{nd, d_names.data(), d_off.data(), nullptr, nullptr, d_rowid.data()};
        const fxi::PageSeq seq_i(tot_t ? seq.at(tot_t) : db.npages + 1, FXI_PAGE);
        if (tot_i) {
            if ((uint64_t)seq.at(tot_t + tot_i - 1) >= 0xFFFFFFF0ull) rc = fail(FX_ERANGE, "the index file would exceed 2^32 pages");
            else {
                const auto g0 = now();
                new_npages = seq.at(tot_t + tot_i - 1);
                if (!(map2.p && map2.len == (size_t)new_npages * FXI_PAGE)) { map2.close(); fxi_grow_and_map(db, new_npages, h->device, map2); }
                lap_buf[3] += secs(g0, now());
                std::thread th_i([&]() { if (!up.write(db.fd, map2, FXI_PAGE, FXI_PAGE, (uint32_t)root_index, seq_i, (size_t)nleaf_i, dv)) host_bad.store(1); });
                std::swap(I.slab.p, T.slab.p); std::swap(I.slab.cap_bytes, T.slab.cap_bytes);      // (the table's slab is free: one 4 GiB block for both trees)
                std::swap(I.slab.dev, T.slab.dev); I.slab.stream = s2;
                mark("room for the index");
                rc = I.leaves_out(true, nleaf_i, first_i.p, seq_i, 0, db.fd, map2, &lap_buf[5], &lap_buf[6]);
                mark("index leaves out");
                th_i.join();
                mark("index upper levels joined");
            }
        } else if (nleaf_i == 1)
            rc = I.root_leaf(true, first_i.p, db.fd, root_index, path);
        if (host_bad.load()) ok = false;
    }
    const auto t4 = now();
    // (header and cut FIRST, the mappings taken down afterwards: an ftruncate behind the munmap of 10 GB of touched pages waits for
    // it -- 0.2 s of "unaccounted" time in one bench run; the pages beyond the new end were never touched through the mappings)
    if (!rc && ok) ok = db.finish(new_npages);
    if (!rc && ok && db.size0 > (off_t)new_npages * FXI_PAGE) ok = ftruncate(db.fd, (off_t)new_npages * FXI_PAGE) == 0;     // a pre-sized file: cut to what was used
    fxi_unmap_later(map2);
    fxi_unmap_later(map);
    if (rc || !ok) db.give_back();
    lap_buf[7] = secs(t4, now());
    mark("header, file cut");
    (void)hipStreamSynchronize(s2);
    mark("second stream idle");
    order.release(); soff.release(); slen.release(); first_i.release(); noff_fa.release();
    I.sz.release(); I.pages.release(); I.bad.release(); I.sums.release(); I.pbase.release(); I.slab.release();
    mark("buffers back in the pool");
    if (trace) fprintf(stderr, "[fxgpu] fxi build: %lld rows, %lld + %lld leaves; sort + index shape %.1f ms beside the table's copy-out, %.1f ms of it not hidden\n",
                       (long long)n, (long long)nleaf_t, (long long)nleaf_i, t_side * 1e3, lap_buf[4] * 1e3);
    if (rc) return done(rc);
    if (!ok) return done(fail(FX_EIO, "cannot write %s", path));
    return done(FX_OK);
}

// ------------------------------------------------------------------ ONE .fxi from SEVERAL handles (round 6)
// A file that is indexed by byte range -- one process per GPU (shard.ShardedFastq), the devices of one process, or windows
// of one device that take turns (windows.WindowedFastq) -- has its rows in several handles: part r holds the rows
// [row_base_r, row_base_r + n_r) of the table, in order.  Round 5 sent every part's table and names to one host and let
// the host page loader (fx_fxi.hpp) format them: 15 M rows/s against the 220 M rows/s of fx_fxi_dev_write.  Here
//   * every part formats ITS table leaves where its rows and names are (rowids are global; a part starts a fresh page,
//     as a fill chunk does) and copies them into its own range of the ONE file: fx_fxi_part_shape -> the caller adds up
//     the leaf counts (the build's all-gather carries them) -> fx_fxi_part_leaves(first page of the new pages, leaves
//     of the parts before);
//   * the index needs all names in one order: every part hands its names -- back to back, with their lengths -- to
//     device memory of the caller's choosing (fx_fxi_part_names: a buffer the process group gathers on the writing
//     rank, RCCL over xGMI; windows: one buffer that outlives the windows); the writer sorts them once
//     (fx_fxi_join_begin), formats the index leaves from that buffer, and writes what the host writes anyway: the
//     interior levels of both trees -- the table's from the first rows of all parts' leaves (fx_fxi_part_firsts, 8 bytes
//     per leaf through the process group) -- and the header (fx_fxi_join_write).
// Why gather + one sort and not a splitter exchange: the keys are the names themselves (3 GB for 10^8 reads), a gather
// moves them once over seven xGMI links into one device that sorts 10^8 keys in 50 ms; a splitter exchange moves the
// same bytes all-to-all, needs a second round for the splitters, and leaves eight ranks to write a third of the file
// (the index leaves) that one PCIe link writes while the other seven are still writing table leaves.
struct fx_fxi_join {
    int device = 0;
    hipStream_t stream = nullptr;
    const uint8_t *d_names = nullptr;
    const int32_t *d_lens = nullptr;
    int64_t n = 0, n_dup = 0;
    ScratchBuf<int64_t> noff, order, soff;
    ScratchBuf<int32_t> slen;
};

extern "C" int fx_fxi_part_shape(fx_handle *h, int kind, int64_t row_base, int64_t *out3) {
    int rc = fxi_check_kind(h, kind);
    if (rc) return rc;
    if (!out3 || row_base < 0) return fail(FX_EINVAL, "bad argument");
    const int64_t n = kind == 0 ? h->n_hdr : h->n_reads;
    out3[0] = n; out3[1] = 0; out3[2] = 0;
    h->fxi_part_first.release(); h->fxi_part_nleaf = 0; h->fxi_part_kind = -1;
    if (n == 0) { h->fxi_part_kind = kind; h->fxi_part_row_base = row_base; return FX_OK; }
    if (n + row_base >= 0xFFFFFFFFll) return fail(FX_ERANGE, "too many records");
    FxiJob J;
    J.device = h->device; J.stream = h->stream; J.data = h->d_data; J.n = n;
    fxi_cols(h, kind, &J.c);
    J.c.row_base = row_base;
    if ((rc = J.init())) return rc;
    J.table_sizes();
    int64_t nleaf = 0;
    if ((rc = J.leaf_level(false, h->fxi_part_first, &nleaf))) return rc;
    // bytes of the names (what fx_fxi_part_names will write)
    const int64_t nchunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    ScratchBuf<int32_t> l32;
    ScratchBuf<int64_t> l64, sums, off;
    if ((rc = l32.alloc(h->device, n, h->stream)) || (rc = l64.alloc(h->device, n, h->stream)) || (rc = sums.alloc(h->device, nchunks + 1, h->stream)) ||
        (rc = off.alloc(h->device, n + 1, h->stream))) return rc;
    hipLaunchKernelGGL(k_len_clamp, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, h->stream, J.c.name_len, n, l32.p, l64.p);
    hipLaunchKernelGGL(k_cnt_chunk_sums, dim3((unsigned)nchunks), dim3(BLOCK), 0, h->stream, l32.p, n, sums.p);
    hipLaunchKernelGGL(k_cnt_chunk_bases, dim3(1), dim3(BLOCK), 0, h->stream, sums.p, nchunks);
    hipLaunchKernelGGL(k_cnt_offsets, dim3((unsigned)nchunks), dim3(BLOCK), 0, h->stream, l32.p, n, sums.p, off.p);
    HIPCHK(hipGetLastError());
    int64_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, off.p + n, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->fxi_part_nleaf = nleaf; h->fxi_part_row_base = row_base; h->fxi_part_kind = kind;
    out3[1] = nleaf; out3[2] = total;
    return FX_OK;
}

extern "C" int fx_fxi_part_firsts(fx_handle *h, int64_t *first_rows) {
    if (!h || h->fxi_part_kind < 0) return fail(FX_ESTATE, "fx_fxi_part_shape has not been called");
    if (h->fxi_part_nleaf == 0) return FX_OK;
    if (!first_rows) return fail(FX_EINVAL, "null first_rows");
    int rc = use_device(h);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(first_rows, h->fxi_part_first.p, (size_t)h->fxi_part_nleaf * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int64_t k = 0; k < h->fxi_part_nleaf; ++k) first_rows[k] += h->fxi_part_row_base;
    return FX_OK;
}

extern "C" int fx_fxi_part_names(fx_handle *h, int kind, uint8_t *d_names, int32_t *d_lens) {
    int rc = fxi_check_kind(h, kind);
    if (rc) return rc;
    const int64_t n = kind == 0 ? h->n_hdr : h->n_reads;
    if (n == 0) return FX_OK;
    if (!d_names || !d_lens) return fail(FX_EINVAL, "null destination");
    FxiCols c;
    fxi_cols(h, kind, &c);
    const int64_t *noff = c.name_off;
    if (kind == 0) {
        if ((rc = h->nm_off.alloc(n))) return rc;
        hipLaunchKernelGGL(k_add_i64, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, h->stream, h->hdr.p, (int64_t)1, n, h->nm_off.p);
        noff = h->nm_off.p;
    }
    const int64_t nchunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    ScratchBuf<int64_t> l64, sums, off;
    if ((rc = l64.alloc(h->device, n, h->stream)) || (rc = sums.alloc(h->device, nchunks + 1, h->stream)) || (rc = off.alloc(h->device, n + 1, h->stream))) return rc;
    hipLaunchKernelGGL(k_len_clamp, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, h->stream, c.name_len, n, d_lens, l64.p);
    hipLaunchKernelGGL(k_cnt_chunk_sums, dim3((unsigned)nchunks), dim3(BLOCK), 0, h->stream, (const int32_t *)d_lens, n, sums.p);
    hipLaunchKernelGGL(k_cnt_chunk_bases, dim3(1), dim3(BLOCK), 0, h->stream, sums.p, nchunks);
    hipLaunchKernelGGL(k_cnt_offsets, dim3((unsigned)nchunks), dim3(BLOCK), 0, h->stream, (const int32_t *)d_lens, n, sums.p, off.p);
    HIPCHK(hipGetLastError());
    if ((rc = fetch_common(h, FX_DEVICE, n, false, noff, l64.p, l64.p, nullptr, FX_RAW, nullptr, d_names, off.p, nullptr, 0))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    return FX_OK;
}

// laps[2]: leaf kernels, leaves to the file
extern "C" int fx_fxi_part_leaves(fx_handle *h, int kind, const char *path, int64_t first_new_page, int64_t leaf_base, double *laps) {
    int rc = fxi_check_kind(h, kind);
    if (rc) return rc;
    if (!path || first_new_page < 2) return fail(FX_EINVAL, "bad argument");
    if (h->fxi_part_kind != kind) return fail(FX_ESTATE, "fx_fxi_part_shape has not been called for this table");
    double lap_buf[2] = {0, 0};
    const int64_t n = kind == 0 ? h->n_hdr : h->n_reads, nleaf = h->fxi_part_nleaf;
    if (n == 0 || nleaf == 0) { if (laps) memcpy(laps, lap_buf, sizeof lap_buf); return FX_OK; }
    FxiJob J;
    J.device = h->device; J.stream = h->stream; J.data = h->d_data; J.n = n;
    fxi_cols(h, kind, &J.c);
    J.c.row_base = h->fxi_part_row_base;
    const int fd = open(path, O_RDWR);
    if (fd < 0) return fail(FX_EIO, "cannot open %s", path);
    if (leaf_base < 0) {                                     // the table's only leaf: it lives in the root page, number -leaf_base
        rc = nleaf == 1 ? J.root_leaf(false, h->fxi_part_first.p, fd, (int)-leaf_base, path) : fail(FX_EINVAL, "a root page takes one leaf, this part has %lld", (long long)nleaf);
        close(fd);
        return rc;
    }
    const fxi::PageSeq seq((uint32_t)first_new_page, FXI_PAGE);
    if ((uint64_t)seq.at((uint64_t)(leaf_base + nleaf - 1)) >= 0xFFFFFFF0ull) { close(fd); return fail(FX_ERANGE, "the index file would exceed 2^32 pages"); }
    // the writer has grown the file (fx_fxi_join_grow) or it has not: what of this part's range exists is written through a
    // mapping, the rest with pwrite (which grows the file by itself; never ftruncate here -- the parts run side by side)
    fxi::FileMap map;
    struct stat st;
    if (!getenv("FX_FXI_NO_MMAP") && fstat(fd, &st) == 0 && st.st_size >= (off_t)seq.at((uint64_t)leaf_base) * FXI_PAGE) {
        const size_t len = (size_t)std::min<off_t>(st.st_size, (off_t)seq.at((uint64_t)(leaf_base + nleaf - 1)) * FXI_PAGE);
        (void)map.map_existing(fd, len);
    }
    rc = J.leaves_out(false, nleaf, h->fxi_part_first.p, seq, leaf_base, fd, map, &lap_buf[0], &lap_buf[1]);
    fxi_unmap_later(map);
    close(fd);
    if (laps) memcpy(laps, lap_buf, sizeof lap_buf);
    return rc;
}

// Room for the table's pages before the parts write them (they map what exists): pages [db pages + 1, ... + all of the
// table's new pages) of `path` and extra_bytes behind them (the index to come); first_new_page <- the page the first
// part's first leaf goes to.  Best effort, like
// fx_fxi_presize_begin: parts that find no room use pwrite.
extern "C" int fx_fxi_join_grow(const char *path, int root_table, int64_t nleaf_table, int64_t extra_bytes, int device, int64_t *first_new_page) {
    if (!path || root_table < 2 || nleaf_table < 0 || extra_bytes < 0 || !first_new_page) return fail(FX_EINVAL, "bad argument");
    fxi::DbFile db;
    const int e = db.open_rw(path, (uint32_t)root_table);
    if (e == fxi::E_IO) return fail(FX_EIO, "cannot open %s", path);
    if (e || db.pagesize != FXI_PAGE || db.usable != FXI_PAGE) return fail(FX_EINVAL, "%s is not a SQLite database this loader can extend (4 KiB pages, no reserved bytes)", path);
    *first_new_page = (int64_t)db.npages + 1;
    const uint64_t tot_t = nleaf_table > 1 ? fxi::table_new_pages((size_t)nleaf_table, fxi::table_fan(FXI_PAGE)) : 0;
    if (tot_t) {
        const fxi::PageSeq seq(db.npages + 1, FXI_PAGE);
        const uint64_t more = (uint64_t)(extra_bytes / FXI_PAGE);          // room for the index as well (an estimate: fx_fxi_join_write allocates what is missing, cuts what is left over)
        if ((uint64_t)seq.at(tot_t - 1) + more >= 0xFFFFFFF0ull) return fail(FX_ERANGE, "the index file would exceed 2^32 pages");
        fxi::FileMap map;
        fxi_grow_and_map(db, seq.at(tot_t - 1) + (uint32_t)more, device, map);
        fxi_unmap_later(map);
    }
    return FX_OK;                                            // (the header still says db.npages: fx_fxi_join_write finishes it)
}

extern "C" int fx_fxi_join_begin(int device, const uint8_t *d_names, const int32_t *d_lens, int64_t n, fx_fxi_join **out, int64_t *n_dup) {
    if (!out || !n_dup || n < 0 || (n > 0 && (!d_names || !d_lens))) return fail(FX_EINVAL, "bad argument");
    *out = nullptr; *n_dup = 0;
    if (n >= 0xFFFFFFFFll) return fail(FX_ERANGE, "too many records for the 32-bit sort index");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(FX_EDEVICE, "device %d is not available; libfxgpu has no CPU fallback", device);
    HIPCHK(hipSetDevice(device));
    std::unique_ptr<fx_fxi_join> j(new fx_fxi_join());
    j->device = device; j->d_names = d_names; j->d_lens = d_lens; j->n = n;
    HIPCHK(hipStreamCreateWithFlags(&j->stream, hipStreamNonBlocking));
    auto bail = [&](int code) { (void)hipStreamSynchronize(j->stream); j->noff.release(); j->order.release(); j->soff.release(); j->slen.release(); (void)hipStreamDestroy(j->stream); return code; };
    if (n) {
        int rc;
        const int64_t nchunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
        ScratchBuf<int64_t> sums, ndup;
        if ((rc = j->noff.alloc(device, n + 1, j->stream)) || (rc = j->order.alloc(device, n, j->stream)) || (rc = j->soff.alloc(device, n, j->stream)) ||
            (rc = j->slen.alloc(device, n, j->stream)) || (rc = sums.alloc(device, nchunks + 1, j->stream)) || (rc = ndup.alloc(device, 1, j->stream))) return bail(rc);
        hipLaunchKernelGGL(k_cnt_chunk_sums, dim3((unsigned)nchunks), dim3(BLOCK), 0, j->stream, d_lens, n, sums.p);
        hipLaunchKernelGGL(k_cnt_chunk_bases, dim3(1), dim3(BLOCK), 0, j->stream, sums.p, nchunks);
        hipLaunchKernelGGL(k_cnt_offsets, dim3((unsigned)nchunks), dim3(BLOCK), 0, j->stream, d_lens, n, sums.p, j->noff.p);
        if (hipGetLastError() != hipSuccess) return bail(fail(FX_EDEVICE, "offsets of the gathered names"));
        const char *what = "";
        const int e = sort_names(d_names, 0, j->noff.p, d_lens, n, j->order.p, ndup.p, j->stream, &what, j->soff.p, j->slen.p);
        if (e) return bail(fail(e == (int)hipErrorOutOfMemory ? FX_ENOMEM : FX_EDEVICE, "name sort, %s: %s", what, hipGetErrorString((hipError_t)e)));
        if (hipMemcpyAsync(&j->n_dup, ndup.p, 8, hipMemcpyDeviceToHost, j->stream) != hipSuccess || hipStreamSynchronize(j->stream) != hipSuccess)
            return bail(fail(FX_EDEVICE, "the name sort failed"));
    }
    *n_dup = j->n_dup;
    *out = j.release();
    return FX_OK;
}

extern "C" void fx_fxi_join_end(fx_fxi_join *j) {
    if (!j) return;
    (void)hipSetDevice(j->device);
    (void)hipStreamSynchronize(j->stream);
    j->noff.release(); j->order.release(); j->soff.release(); j->slen.release();
    (void)hipStreamDestroy(j->stream);
    delete j;
}

// The writer's share once the parts' leaves are in the file (or on their way: nothing here touches their pages): the
// interior levels of the table from first_rows[0 .. nleaf_table) (0-based first row of every leaf, parts in order), the
// index from the joined names (j; root_index 0 or j null: none -- the caller lets SQLite build it, or not), the header.
// laps[6]: [0] file grown, [1] index shape + dividers, [2] index leaf kernels, [3] index leaves to the file, [4] host levels + header, [5] table interior
extern "C" int fx_fxi_join_write(fx_fxi_join *j, const char *path, int root_table, int root_index, int64_t n_rows, int64_t nleaf_table,
                                 const int64_t *first_rows, int64_t first_new_page, double *laps) {
    if (!path || root_table < 2 || (root_index != 0 && root_index < 2) || n_rows < 0 || nleaf_table < 0 || (nleaf_table > 0 && !first_rows))
        return fail(FX_EINVAL, "bad argument");
    if (root_index && (!j || j->n != n_rows || j->n_dup)) return fail(FX_ESTATE, "the index needs the joined names of all %lld rows, distinct", (long long)n_rows);
    double lap_buf[6] = {0, 0, 0, 0, 0, 0};
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    auto done = [&](int code) { if (laps) memcpy(laps, lap_buf, sizeof lap_buf); return code; };
    if (n_rows == 0) return done(FX_OK);
    fxi::DbFile db;
    {
        const int e = db.open_rw(path, (uint32_t)root_table);
        if (e == fxi::E_IO) return done(fail(FX_EIO, "cannot open %s", path));
        if (e || db.pagesize != FXI_PAGE || db.usable != FXI_PAGE || (root_index && (uint32_t)root_index > db.npages))
            return done(fail(FX_EINVAL, "%s is not a SQLite database this loader can extend (4 KiB pages, no reserved bytes)", path));
    }
    if ((int64_t)db.npages + 1 != first_new_page) return done(fail(FX_ESTATE, "%s has %u pages, the parts wrote their leaves from page %lld on", path, db.npages, (long long)first_new_page));
    const auto t0 = now();
    // ---- the index: shape and dividers
    FxiJob J;
    ScratchBuf<int64_t> first_i;
    int64_t nleaf_i = 0, nd = 0;
    std::vector<int64_t> d_rowid(1), d_off(1, 0);
    std::vector<uint8_t> d_names(1);
    fxi::IndexUpper up;
    int rc = FX_OK;
    if (root_index) {
        HIPCHK(hipSetDevice(j->device));
        J.device = j->device; J.stream = j->stream; J.data = j->d_names; J.n = n_rows; J.order = j->order.p;
        memset(&J.c, 0, sizeof J.c);
        J.c.name_off = j->noff.p; J.c.name_len = j->d_lens;
        J.c.s_off = j->soff.p; J.c.s_len = j->slen.p;
        if ((rc = J.init())) return done(rc);
        J.index_sizes();
        if ((rc = J.leaf_level(true, first_i, &nleaf_i))) return done(rc);
        nd = nleaf_i - 1;
        if ((rc = J.dividers(first_i.p, nd, d_rowid, d_off, d_names))) return done(rc);
    }
    const fxi::Entries dv{nd, d_names.data(), d_off.data(), nullptr, nullptr, d_rowid.data()};
    if (root_index && !up.plan((size_t)nleaf_i, dv, FXI_PAGE)) return done(fail(FX_ERANGE, "an index entry does not fit an interior page: use CREATE INDEX"));
    const auto t1 = now();
    lap_buf[1] = secs(t0, t1);
    // ---- the page sequence: table leaves (the parts'), table interior levels, index leaves, index upper levels
    const uint64_t tot_t = nleaf_table > 1 ? fxi::table_new_pages((size_t)nleaf_table, fxi::table_fan(FXI_PAGE)) : 0;
    const uint64_t tot_i = nleaf_i > 1 ? (uint64_t)nleaf_i + up.pages : 0;
    const uint64_t total = tot_t + tot_i;
    const fxi::PageSeq seq(db.npages + 1, FXI_PAGE);
    if (total && (uint64_t)seq.at(total - 1) >= 0xFFFFFFF0ull) return done(fail(FX_ERANGE, "the index file would exceed 2^32 pages"));
    fxi::FileMap map;
    uint32_t new_npages = db.npages;
    if (total) {
        new_npages = seq.at(total - 1);
        fxi_grow_and_map(db, new_npages, j ? j->device : 0, map);
    }
    const auto t2 = now();
    lap_buf[0] = secs(t1, t2);
    const fxi::PageSeq seq_i(total && tot_t ? seq.at(tot_t) : db.npages + 1, FXI_PAGE);
    std::vector<int64_t> lf((size_t)nleaf_table + 1);
    if (nleaf_table) memcpy(lf.data(), first_rows, (size_t)nleaf_table * 8);
    lf[(size_t)nleaf_table] = n_rows;
    std::atomic<int> host_bad(0);
    double t_interior = 0;
    std::thread th_t, th_i;
    if (nleaf_table > 1)
        th_t = std::thread([&]() {
            const auto a = now();
            if (!fxi::table_interior(db.fd, map, FXI_PAGE, FXI_PAGE, (uint32_t)root_table, seq, lf.data(), (size_t)nleaf_table, tot_t)) host_bad.store(1);
            t_interior = secs(a, now());
        });
    if (nleaf_i > 1)
        th_i = std::thread([&]() { if (!up.write(db.fd, map, FXI_PAGE, FXI_PAGE, (uint32_t)root_index, seq_i, (size_t)nleaf_i, dv)) host_bad.store(1); });
    if (root_index) {
        if (nleaf_i == 1) rc = J.root_leaf(true, first_i.p, db.fd, root_index, path);
        else rc = J.leaves_out(true, nleaf_i, first_i.p, seq_i, 0, db.fd, map, &lap_buf[2], &lap_buf[3]);
    }
    const auto t3 = now();
    if (th_t.joinable()) th_t.join();
    if (th_i.joinable()) th_i.join();
    lap_buf[5] = t_interior;
    bool ok = !host_bad.load();
    if (!rc && ok) ok = db.finish(new_npages);
    if (!rc && ok) {
        struct stat st;
        if (fstat(db.fd, &st) == 0 && st.st_size > (off_t)new_npages * FXI_PAGE) ok = ftruncate(db.fd, (off_t)new_npages * FXI_PAGE) == 0;     // a pre-sized file: cut to what was used
    }
    fxi_unmap_later(map);                                    // (behind the cut, see fx_fxi_dev_build)
    if (rc || !ok) db.give_back();
    lap_buf[4] = secs(t3, now());
    if (rc) return done(rc);
    if (!ok) return done(fail(FX_EIO, "cannot write %s", path));
    return done(FX_OK);
}

// Room for the pages to come, set aside WHILE THE STREAM IS STAGED: the caller that knows roughly how large the index file
// will be (an estimate from the head of the input) has SQLite create the database, then lets a thread of this library
// fallocate the file to that size -- 0.6 s for 10 GB that fx_fxi_dev_write would otherwise spend between the kernels and
// the first page it stores.  The database header still says where the database ends (fx_fxi.hpp: DbFile); fx_fxi_dev_write
// cuts the file to what it used.  Best effort: a file system without fallocate, or no space, just leaves the file as it is.
struct FxiPresize {
    std::thread th;
    std::atomic<bool> stop{false};
};
extern "C" int fx_fxi_presize_begin(const char *path, int64_t bytes, int device, void **token) {
    if (!path || !token || bytes < 0) return fail(FX_EINVAL, "bad argument");
    *token = nullptr;
    const int fd = open(path, O_RDWR);
    if (fd < 0) return fail(FX_EIO, "cannot open %s", path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return fail(FX_EIO, "cannot stat %s", path); }
    FxiPresize *p = new FxiPresize();
    const off_t from = st.st_size, to = (off_t)bytes;
    cpu_set_t near_cpus;
    static const bool no_bind = [] { const char *e = getenv("FX_FXI_NO_BIND"); return e && atoi(e) != 0; }();
    const bool bind = !no_bind && device >= 0 && device_cpus(device, &near_cpus);     // the pages on the memory next to the device that will fill them
    p->th = std::thread([p, fd, from, to, bind, near_cpus]() {
        if (bind) (void)pthread_setaffinity_np(pthread_self(), sizeof near_cpus, &near_cpus);
        const off_t step = 256ll << 20;
        for (off_t o = from; o < to && !p->stop.load(); o += step)
            if (fallocate(fd, 0, o, std::min(step, to - o)) != 0) break;
        close(fd);
    });
    *token = p;
    return FX_OK;
}
extern "C" int fx_fxi_presize_end(void *token, int cancel) {
    FxiPresize *p = (FxiPresize *)token;
    if (!p) return FX_OK;
    if (cancel) p->stop.store(true);
    if (p->th.joinable()) p->th.join();
    delete p;
    return FX_OK;
}

// ------------------------------------------------------------------ the collective under the C ABI
// One process per GPU; the sharded build needs ONE all-gather of 28 words per rank.  RCCL is bound at run time
// (dlopen: a single-GPU user never loads it, and a process that already carries an RCCL -- torch's -- shares that one
// instead of getting a second copy of the library), through the function types of <rccl/rccl.h>.
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;
    bool ok() const { return lib != nullptr; }
};
Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *env = getenv("FX_RCCL_LIB");
        const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void *lib = nullptr;
        for (const char *n : {"librccl.so.1", "librccl.so"})           // already in the process (torch)?
            if (!lib && !env) lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        for (const char *n : names)
            if (!lib && n) lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!lib) { const char *e = dlerror(); r.why = e ? e : "librccl.so.1 not found"; return; }
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(lib, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(lib, "ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))dlsym(lib, "ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) { r.why = "RCCL symbols missing"; return; }
        r.lib = lib;
    });
    return r;
}
}  // namespace

struct fx_comm {
    int rank = 0, world = 1, device = 0;
    ncclComm_t comm = nullptr;
    int64_t *d_send = nullptr, *d_recv = nullptr;          // 28 words; world x 28 words
    uint8_t *d_buf = nullptr;                              // staging of fx_comm_allgather: (world + 1) x COMM_SLOT bytes
    hipStream_t stream = nullptr;                          // for collectives that belong to no handle
};
static const int64_t COMM_SLOT = 64 << 10;
#define RCCLCHK(expr)                                                                                          \
    do {                                                                                                       \
        ncclResult_t r__ = (expr);                                                                             \
        if (r__ != ncclSuccess) return fail(FX_EDEVICE, "%s: %s", #expr, rccl().GetErrorString(r__));          \
    } while (0)

extern "C" int fx_comm_unique_id(uint8_t id[128]) {
    if (!id) return fail(FX_EINVAL, "null argument");
    if (!rccl().ok()) return fail(FX_EDEVICE, "RCCL is not available: %s", rccl().why.c_str());
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    RCCLCHK(rccl().GetUniqueId(&u));
    memcpy(id, &u, 128);
    return FX_OK;
}

extern "C" int fx_comm_destroy(fx_comm *c) {
    if (!c) return FX_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); }
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->d_buf) (void)hipFree(c->d_buf);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return FX_OK;
}

extern "C" int fx_comm_init(int rank, int world, const uint8_t id[128], int device, fx_comm **out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return fail(FX_EINVAL, "bad argument");
    if (!rccl().ok()) return fail(FX_EDEVICE, "RCCL is not available: %s", rccl().why.c_str());
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(FX_EDEVICE, "device %d out of range (have %d)", device, ndev);
    HIPCHK(hipSetDevice(device));
    fx_comm *c = new fx_comm();
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId u;
    memcpy(&u, id, 128);
    ncclResult_t r = rccl().CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) { c->comm = nullptr; fx_comm_destroy(c); return fail(FX_EDEVICE, "ncclCommInitRank: %s", rccl().GetErrorString(r)); }
    hipError_t e = hipMalloc((void **)&c->d_send, sizeof(fx_shard_summary));
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_recv, sizeof(fx_shard_summary) * (size_t)world);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_buf, (size_t)COMM_SLOT * (size_t)(world + 1));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { fx_comm_destroy(c); return fail(FX_ENOMEM, "fx_comm_init: %s", hipGetErrorString(e)); }
    *out = c;
    return FX_OK;
}

extern "C" int fx_comm_rank(const fx_comm *c) { return c ? c->rank : -1; }
extern "C" int fx_comm_world(const fx_comm *c) { return c ? c->world : 0; }

// nbytes (<= 64 KiB) of every rank to every rank, host buffers: the small exchanges around the build that have no kernel of
// their own (FASTQ: newline count and last newline of every shard's core; composition: the 128 counts of a shard's lead).
extern "C" int fx_comm_allgather(fx_comm *c, const void *send, void *recv, int64_t nbytes) {
    if (!c || !send || !recv || nbytes <= 0 || nbytes > COMM_SLOT) return fail(FX_EINVAL, "bad argument (at most %lld bytes per rank)", (long long)COMM_SLOT);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->d_buf, send, (size_t)nbytes, hipMemcpyHostToDevice, c->stream));
    RCCLCHK(rccl().AllGather(c->d_buf, c->d_buf + COMM_SLOT, (size_t)nbytes, ncclInt8, c->comm, c->stream));
    HIPCHK(hipMemcpyAsync(recv, c->d_buf + COMM_SLOT, (size_t)nbytes * (size_t)c->world, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return FX_OK;
}

// pyfastx_create_index (index.c:109-388) for ONE rank's byte range of the stream: everything enqueued on the handle's
// stream -- scan + tables (fx_fasta_build_begin), this shard's boundary summary into the send buffer, THE all-gather
// (ncclAllGather, RCCL over xGMI: 28 x int64 per rank), completion of the record that crosses the cut (k_stitch_tail).
// fx_fasta_build_end (or fx_fasta_build_sharded) reads the totals.  Device-side consumers may follow at once.
extern "C" int fx_fasta_build_sharded_begin(fx_handle *h, fx_comm *c, int full_name) {
    if (!h || !c) return fail(FX_EINVAL, "null argument");
    if (h->device != c->device) return fail(FX_EINVAL, "handle on device %d, communicator on device %d", h->device, c->device);
    int rc = fx_fasta_build_begin(h, full_name);
    if (rc) return rc;
    if ((rc = fx_shard_summary_dev(h, c->d_send))) return rc;
    RCCLCHK(rccl().AllGather(c->d_send, c->d_recv, sizeof(fx_shard_summary) / 8, ncclInt64, c->comm, h->stream));
    return fx_fasta_stitch_dev(h, c->d_recv, c->world, c->rank, full_name & 1);
}

extern "C" int fx_fasta_build_sharded(fx_handle *h, fx_comm *c, int full_name, fx_fasta_summary *out) {
    int rc = fx_fasta_build_sharded_begin(h, c, full_name);
    if (rc) return rc;
    return fx_fasta_build_end(h, out);
}

// every rank's summary as the all-gather delivered it (host copy; world x 28 words), e.g. for a caller that merges tables
extern "C" int fx_comm_summaries(fx_comm *c, fx_handle *h, fx_shard_summary *out) {
    if (!c || !h || !out) return fail(FX_EINVAL, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(out, c->d_recv, sizeof(fx_shard_summary) * (size_t)c->world, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return FX_OK;
}

// pyfastx_fastq_create_index (fastq.c:8-182) for one rank's byte range (fx_set_shard + fx_set_halo done): count pass, ONE
// all-gather of two integers per rank (newlines of the core, offset of the last one), then the rows with the global
// line numbering.
extern "C" int fx_fastq_build_sharded(fx_handle *h, fx_comm *c, fx_fastq_summary *out) {
    if (!h || !c) return fail(FX_EINVAL, "null argument");
    int64_t mine[2] = {0, -1};
    int rc = fx_fastq_scan(h, &mine[0], &mine[1]);
    if (rc) return rc;
    std::vector<int64_t> all((size_t)c->world * 2);
    if ((rc = fx_comm_allgather(c, mine, all.data(), sizeof mine))) return rc;
    int64_t loff = 0, prev = -1;
    for (int r = 0; r < c->rank; ++r) { loff += all[(size_t)r * 2]; if (all[(size_t)r * 2] > 0) prev = all[(size_t)r * 2 + 1]; }
    return fx_fastq_build_ctx(h, loff, prev, out);
}
