// gathercal.hip -- calibration of rocprofv3's FETCH_SIZE for the access patterns of the gather kernels (not part of the product).
// MI355X_MICROARCH.md: FETCH_SIZE on gfx950 reports half of a wide coalesced stream; "other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern".  Every kernel here reads a byte count known in advance, out
// of a 3 GiB buffer (beyond the 256 MiB Infinity Cache), in the shapes k_fetch_lines / k_fastq_fetch use:
//   cal_stream      1 GiB, 16 B per lane, coalesced                                   (the guide's case: expect raw = 1/2)
//   cal_line128     10^6 random 128-byte ALIGNED lines, 8 lanes x 16 B each             known: 128 MB
//   cal_half64      10^6 random 64-byte aligned half lines, 4 lanes x 16 B each         known: 64 MB (if memory is fetched 64 B at a time)
//   cal_unal100     10^6 random UNALIGNED 100-byte spans, 8 lanes x 16 B (7 used)       known: the distinct 64-B / 128-B blocks the spans touch (counted on the host)
//   cal_unal300     10^6 random unaligned 302-byte spans (a FASTQ read: sequence + quality), 19 lanes x 16 B
// The offsets are read too (8 MB, coalesced).  Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and, separately, the raw
// request counters; tools/gpu_r06.sh step `gathercal` writes profiles/r06_gathercal.txt with the factors.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/gathercal tools/gathercal.hip && /tmp/gathercal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <unordered_set>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint4 __attribute__((aligned(1))) uint4_u;

__global__ __launch_bounds__(256) void cal_stream(const uint4 *__restrict__ d, int64_t n16, uint32_t *out) {
    uint32_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(d + i));
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) out[0] = acc;
}
__global__ __launch_bounds__(256) void cal_line128(const uint8_t *__restrict__ data, const int64_t *__restrict__ off, int64_t nq, uint32_t *out) {
    const int lane = threadIdx.x & 63, sub = lane & 7, grp = lane >> 3;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
    uint32_t acc = 0;
    for (int64_t i0 = wave * 8; i0 < nq; i0 += nwaves * 8) {
        const int64_t i = i0 + grp;
        if (i >= nq) continue;
        const uint4 v = *reinterpret_cast<const uint4 *>(data + off[i] + 16 * sub);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) out[0] = acc;
}
__global__ __launch_bounds__(256) void cal_half64(const uint8_t *__restrict__ data, const int64_t *__restrict__ off, int64_t nq, uint32_t *out) {
    const int lane = threadIdx.x & 63, sub = lane & 3, grp = lane >> 2;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
    uint32_t acc = 0;
    for (int64_t i0 = wave * 16; i0 < nq; i0 += nwaves * 16) {
        const int64_t i = i0 + grp;
        if (i >= nq) continue;
        const uint4 v = *reinterpret_cast<const uint4 *>(data + off[i] + 16 * sub);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) out[0] = acc;
}
__global__ __launch_bounds__(256) void cal_unal100(const uint8_t *__restrict__ data, const int64_t *__restrict__ off, int64_t nq, uint32_t *out) {
    const int lane = threadIdx.x & 63, sub = lane & 7, grp = lane >> 3;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
    uint32_t acc = 0;
    for (int64_t i0 = wave * 8; i0 < nq; i0 += nwaves * 8) {
        const int64_t i = i0 + grp;
        if (i >= nq || sub >= 7) continue;                             // 7 x 16 = 112 >= 100 bytes
        const uint4 v = *reinterpret_cast<const uint4_u *>(data + off[i] + 16 * sub);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) out[0] = acc;
}
__global__ __launch_bounds__(256) void cal_unal300(const uint8_t *__restrict__ data, const int64_t *__restrict__ off, int64_t nq, uint32_t *out) {
    const int lane = threadIdx.x & 63, sub = lane & 31, grp = lane >> 5;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
    uint32_t acc = 0;
    for (int64_t i0 = wave * 2; i0 < nq; i0 += nwaves * 2) {
        const int64_t i = i0 + grp;
        if (i >= nq || sub >= 19) continue;                            // 19 x 16 = 304 >= 302 bytes
        const uint4 v = *reinterpret_cast<const uint4_u *>(data + off[i] + 16 * sub);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) out[0] = acc;
}

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rng() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static void blocks_touched(const std::vector<int64_t> &off, int span_loaded, int64_t &b64, int64_t &b128) {
    std::unordered_set<int64_t> s64, s128;
    s64.reserve(off.size() * 3); s128.reserve(off.size() * 2);
    for (int64_t o : off)
        for (int64_t a = o >> 6; a <= (o + span_loaded - 1) >> 6; ++a) { s64.insert(a); s128.insert(a >> 1); }
    b64 = (int64_t)s64.size() * 64; b128 = (int64_t)s128.size() * 128;
}

int main() {
    const int64_t n = 3ll << 30, nq = 1000000;
    uint8_t *d; uint32_t *out; int64_t *off;
    CK(hipMalloc((void **)&d, n + 4096)); CK(hipMalloc((void **)&out, 256)); CK(hipMalloc((void **)&off, nq * 8));
    CK(hipMemset(d, 65, n + 4096));
    std::vector<int64_t> h(nq);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char *name, auto launch, int64_t k64, int64_t k128) -> int {
        float tot = 0;
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(a, 0)); launch(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r) tot += ms;
        }
        printf("{\"kernel\": \"%s\", \"launches\": 4, \"ms_avg\": %.4f, \"known_bytes_64B_blocks\": %lld, \"known_bytes_128B_blocks\": %lld, \"offsets_bytes\": %lld}\n",
               name, tot / 3, (long long)k64, (long long)k128, (long long)(nq * 8));
        return 0;
    };
    // 1. the stream
    timeit("cal_stream", [&] { hipLaunchKernelGGL(cal_stream, dim3(256 * 16), dim3(256), 0, 0, (const uint4 *)d, (int64_t)((1ll << 30) / 16), out); }, 1ll << 30, 1ll << 30);
    // 2. aligned 128-byte lines, distinct
    {
        std::unordered_set<int64_t> seen;
        for (auto &x : h) { int64_t l; do { l = (int64_t)(rng() % (uint64_t)(n >> 7)); } while (!seen.insert(l).second); x = l << 7; }
        CK(hipMemcpy(off, h.data(), nq * 8, hipMemcpyHostToDevice));
        timeit("cal_line128", [&] { hipLaunchKernelGGL(cal_line128, dim3(8192), dim3(256), 0, 0, d, off, nq, out); }, nq * 128, nq * 128);
    }
    // 3. aligned 64-byte half lines, no two in one 128-byte line
    {
        std::unordered_set<int64_t> seen;
        for (auto &x : h) { int64_t l; do { l = (int64_t)(rng() % (uint64_t)(n >> 6)); } while (!seen.insert(l >> 1).second); x = l << 6; }
        CK(hipMemcpy(off, h.data(), nq * 8, hipMemcpyHostToDevice));
        timeit("cal_half64", [&] { hipLaunchKernelGGL(cal_half64, dim3(8192), dim3(256), 0, 0, d, off, nq, out); }, nq * 64, nq * 128);
    }
    // 4. unaligned 100-byte spans (112 loaded)
    {
        for (auto &x : h) x = (int64_t)(rng() % (uint64_t)(n - 4096));
        CK(hipMemcpy(off, h.data(), nq * 8, hipMemcpyHostToDevice));
        int64_t b64, b128; blocks_touched(h, 112, b64, b128);
        timeit("cal_unal100", [&] { hipLaunchKernelGGL(cal_unal100, dim3(8192), dim3(256), 0, 0, d, off, nq, out); }, b64, b128);
    }
    // 5. unaligned 302-byte spans (304 loaded)
    {
        for (auto &x : h) x = (int64_t)(rng() % (uint64_t)(n - 4096));
        CK(hipMemcpy(off, h.data(), nq * 8, hipMemcpyHostToDevice));
        int64_t b64, b128; blocks_touched(h, 304, b64, b128);
        timeit("cal_unal300", [&] { hipLaunchKernelGGL(cal_unal300, dim3(8192), dim3(256), 0, 0, d, off, nq, out); }, b64, b128);
    }
    return 0;
}
