// gatherbench.hip -- floor of the random 100-byte gather on MI355X (not part of the product):
// 1 M queries x 8 lanes x one unaligned 16-byte load + one unaligned 16-byte store, offsets uniform over 3 GB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint4 __attribute__((aligned(1))) uint4_u;

template <int PRE>
__global__ __launch_bounds__(256) void gather(const uint8_t *__restrict__ data, const int64_t *__restrict__ off, int64_t nq, uint8_t *__restrict__ out) {
    const int lane = threadIdx.x & 63, sub = lane & 7, grp = lane >> 3;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
    if (PRE == 0) {
        for (int64_t i0 = wave * 8; i0 < nq; i0 += nwaves * 8) {
            const int64_t i = i0 + grp;
            if (i >= nq) continue;
            const int64_t o = off[i];
            const uint4 v = *reinterpret_cast<const uint4_u *>(data + o + 16 * sub);
            if (sub < 6) *reinterpret_cast<uint4_u *>(out + i * 100 + 16 * sub) = v;
            else if (sub == 6) *reinterpret_cast<uint32_t *>(out + i * 100 + 96) = v.x;
        }
    } else {            // offsets AND data one iteration ahead
        int64_t i = wave * 8 + grp;
        int64_t o = i < nq ? off[i] : 0;
        uint4 v = *reinterpret_cast<const uint4_u *>(data + o + 16 * sub);
        int64_t o2 = (i + nwaves * 8) < nq ? off[i + nwaves * 8] : 0;
        for (; i < nq; i += nwaves * 8) {
            const uint4 cur = v;
            const int64_t i3 = i + 2 * nwaves * 8;
            v = *reinterpret_cast<const uint4_u *>(data + o2 + 16 * sub);
            o2 = i3 < nq ? off[i3] : 0;
            if (sub < 6) *reinterpret_cast<uint4_u *>(out + i * 100 + 16 * sub) = cur;
            else if (sub == 6) *reinterpret_cast<uint32_t *>(out + i * 100 + 96) = cur.x;
        }
    }
}

int main() {
    const int64_t n = 3050ll << 20, nq = 1000000;
    uint8_t *d, *out; int64_t *off;
    CK(hipMalloc((void **)&d, n + 4096)); CK(hipMalloc((void **)&out, nq * 100 + 256)); CK(hipMalloc((void **)&off, nq * 8));
    CK(hipMemset(d, 65, n));
    std::vector<int64_t> h(nq);
    uint64_t s = 88172645463325252ull;
    for (auto &x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (int64_t)(s % (uint64_t)(n - 200)); }
    CK(hipMemcpy(off, h.data(), nq * 8, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int pre = 0; pre < 2; ++pre) for (int grid : {2048, 4096, 8192, 16384, 31250}) {
        float tot = 0;
        for (int r = 0; r < 12; ++r) {
            CK(hipEventRecord(a, 0));
            if (pre) hipLaunchKernelGGL(gather<1>, dim3(grid), dim3(256), 0, 0, d, off, nq, out);
            else     hipLaunchKernelGGL(gather<0>, dim3(grid), dim3(256), 0, 0, d, off, nq, out);
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (r >= 2) tot += ms;
        }
        printf("pre %d grid %6d  %.4f ms\n", pre, grid, tot / 10);
    }
    return 0;
}
