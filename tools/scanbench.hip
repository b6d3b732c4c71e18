// scanbench.hip -- stand-alone tuning harness for the delimiter-scan kernel (K1).
// Not part of the product: compiles variants of k_scan's memory-access shape and
// times them on a 3 GB buffer shaped like a 60-column FASTA.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scanbench tools/scanbench.hip && ./scanbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t zero_bytes(uint32_t x) { return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu); }
__device__ __forceinline__ uint32_t flags4(uint32_t t) { return (((t >> 7) & 0x01010101u) * 0x00204081u >> 21) & 0xFu; }
__device__ __forceinline__ uint32_t eq_mask16(const uint4 &v, uint32_t pat) {
    return flags4(zero_bytes(v.x ^ pat)) | (flags4(zero_bytes(v.y ^ pat)) << 4) | (flags4(zero_bytes(v.z ^ pat)) << 8) |
           (flags4(zero_bytes(v.w ^ pat)) << 12);
}
__device__ __forceinline__ uint32_t any_eq16(const uint4 &v, uint32_t pat) {
    return zero_bytes(v.x ^ pat) | zero_bytes(v.y ^ pat) | zero_bytes(v.z ^ pat) | zero_bytes(v.w ^ pat);
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// MODE 0: one workgroup per tile.  MODE 1: persistent grid-stride over tiles.
// NT: non-temporal loads/stores.  MASK: 0 = no mask store (read-only ceiling), 1 = u16 per lane.
template <int BLOCK, int UNROLL, int MODE, int NT, int MASK>
__global__ __launch_bounds__(BLOCK) void scan(const uint8_t *__restrict__ data, int64_t n, uint16_t *__restrict__ nlmask,
                                             uint32_t *__restrict__ tile_nl, uint32_t *__restrict__ tile_hdr, int64_t ntiles) {
    __shared__ uint32_t lds[2][BLOCK / 64];
    __shared__ __attribute__((aligned(16))) uint16_t mlds[MASK == 2 ? BLOCK * UNROLL : 8];
    constexpr int TILE = BLOCK * 16 * UNROLL;
    const int tid = threadIdx.x;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += (MODE ? gridDim.x : ntiles)) {
        const int64_t tbase = tile * (int64_t)TILE;
        uint4 v[UNROLL];
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const uint4 *p = reinterpret_cast<const uint4 *>(data + tbase + (int64_t)(j * BLOCK + tid) * 16);
            if (NT) { v[j].x = __builtin_nontemporal_load(&p->x); v[j].y = __builtin_nontemporal_load(&p->y);
                      v[j].z = __builtin_nontemporal_load(&p->z); v[j].w = __builtin_nontemporal_load(&p->w); }
            else v[j] = *p;
        }
        uint32_t cnt = 0, hcnt = 0;
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const uint32_t m = eq_mask16(v[j], 0x0A0A0A0Au);
            if (MASK == 1) {
                uint16_t *q = nlmask + tile * (BLOCK * UNROLL) + j * BLOCK + tid;
                if (NT) __builtin_nontemporal_store((uint16_t)m, q); else *q = (uint16_t)m;
            }
            if (MASK == 2) mlds[j * BLOCK + tid] = (uint16_t)m;
            cnt += __popc(m);
            if (any_eq16(v[j], 0x3E3E3E3Eu)) {
                uint32_t g = eq_mask16(v[j], 0x3E3E3E3Eu);
                const int64_t p = tbase + (int64_t)(j * BLOCK + tid) * 16;
                while (g) { const int k = __ffs(g) - 1; g &= g - 1; const int64_t pos = p + k; hcnt += ((pos ? data[pos - 1] : 10) == '\n'); }
            }
        }
        cnt = wave_sum(cnt); hcnt = wave_sum(hcnt);
        __syncthreads();
        if (MASK == 2) {        // transposed through LDS: every lane stores 16 contiguous bytes (UNROLL*2 B per thread)
            static_assert(UNROLL == 8 || UNROLL == 4 || MASK != 2, "unroll");
            if (UNROLL == 8) {
                const uint4 mv = *reinterpret_cast<const uint4 *>(&mlds[tid * 8]);
                uint4 *q = reinterpret_cast<uint4 *>(nlmask + tile * (BLOCK * UNROLL) + tid * 8);
                if (NT) { __builtin_nontemporal_store(mv.x, &q->x); __builtin_nontemporal_store(mv.y, &q->y);
                          __builtin_nontemporal_store(mv.z, &q->z); __builtin_nontemporal_store(mv.w, &q->w); }
                else *q = mv;
            } else {
                const uint2 mv = *reinterpret_cast<const uint2 *>(&mlds[tid * 4]);
                uint2 *q = reinterpret_cast<uint2 *>(nlmask + tile * (BLOCK * UNROLL) + tid * 4);
                *q = mv;
            }
        }
        if ((tid & 63) == 0) { lds[0][tid >> 6] = cnt; lds[1][tid >> 6] = hcnt; }
        __syncthreads();
        if (tid == 0) {
            uint32_t a = 0, b = 0;
            for (int i = 0; i < BLOCK / 64; ++i) { a += lds[0][i]; b += lds[1][i]; }
            tile_nl[tile] = a; tile_hdr[tile] = b;
        }
    }
}

__global__ void fill(uint8_t *d, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t h = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 13);
        h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
        const char b[8] = {'A', 'C', 'G', 'T', 'a', 'c', 'g', 't'};
        d[i] = (i % 61 == 60) ? '\n' : (uint8_t)b[h & 7];
    }
}

template <int BLOCK, int UNROLL, int MODE, int NT, int MASK>
int run(const char *name, const uint8_t *d, int64_t n, uint16_t *mask, uint32_t *t1, uint32_t *t2, int gridcap) {
    constexpr int TILE = BLOCK * 16 * UNROLL;
    const int64_t ntiles = n / TILE;
    const unsigned grid = MODE ? (unsigned)gridcap : (unsigned)ntiles;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((scan<BLOCK, UNROLL, MODE, NT, MASK>), dim3(grid), dim3(BLOCK), 0, 0, d, n, mask, t1, t2, ntiles);
    CK(hipDeviceSynchronize());
    const int R = 10;
    float best = 1e9f, tot = 0;
    for (int r = 0; r < R; ++r) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((scan<BLOCK, UNROLL, MODE, NT, MASK>), dim3(grid), dim3(BLOCK), 0, 0, d, n, mask, t1, t2, ntiles);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best; tot += ms;
    }
    const double bytes = (double)ntiles * TILE;
    printf("%-44s grid %7u  avg %.4f ms  best %.4f ms  -> %.0f GB/s (best %.0f)\n", name, grid, tot / R, best,
           bytes / (tot / R * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9);
    return 0;
}

int main() {
    const int64_t n = 3050ll << 20;
    uint8_t *d; uint16_t *mask; uint32_t *t1, *t2;
    CK(hipMalloc((void **)&d, n + (1 << 20)));
    CK(hipMalloc((void **)&mask, n / 8 + (1 << 20)));
    CK(hipMalloc((void **)&t1, 4 << 20)); CK(hipMalloc((void **)&t2, 4 << 20));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, d, n + (1 << 20));
    CK(hipDeviceSynchronize());
#define RUN(B, U, M, NT, MK, cap) if (run<B, U, M, NT, MK>("block " #B " unroll " #U " mode " #M " nt " #NT " mask " #MK, d, n, mask, t1, t2, cap)) return 1;
    RUN(256, 8, 0, 0, 1, 0)      // round-1 v0 product shape
    RUN(256, 8, 0, 0, 0, 0)      // no mask store: read-only ceiling of this shape
    RUN(256, 8, 0, 1, 0, 0)
    RUN(256, 8, 0, 1, 1, 0)
    RUN(256, 8, 0, 0, 2, 0)      // LDS-transposed 16 B/lane mask store
    RUN(256, 8, 0, 1, 2, 0)
    RUN(256, 4, 0, 0, 2, 0)
    RUN(512, 8, 0, 0, 2, 0)
    RUN(512, 8, 0, 1, 2, 0)
    RUN(1024, 4, 0, 0, 2, 0)
    RUN(1024, 4, 0, 1, 1, 0)
    RUN(1024, 8, 0, 0, 2, 0)
    RUN(1024, 8, 0, 1, 2, 0)
    return 0;
}
