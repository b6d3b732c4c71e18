// readpat.hip -- which order of 4 KiB granules over the resident waves streams a 3 GB buffer fastest (not part of
// the product).  Every wave reads `gpw` granules (4 x global_load_dwordx4 nt per granule, double-buffered) and folds
// them into a checksum; wave t of a team of `team` waves takes granules t, t + team, ... of the team's window.
//   team 1 = each wave walks its own contiguous run;  team >= waves = plain grid-stride.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/readpat tools/readpat.hip && /tmp/readpat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void load_granule(uint4 (&v)[4], const uint8_t *__restrict__ data, int64_t gs, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint4 *q = reinterpret_cast<const uint4 *>(data + gs + (j * 64 + lane) * 16);
        v[j].x = __builtin_nontemporal_load(&q->x); v[j].y = __builtin_nontemporal_load(&q->y);
        v[j].z = __builtin_nontemporal_load(&q->z); v[j].w = __builtin_nontemporal_load(&q->w);
    }
}

template <int WPB>
__global__ __launch_bounds__(WPB * 64) void k_read(const uint8_t *__restrict__ data, int64_t ngran, int gpw, int64_t team, uint32_t *out) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t wave = (int64_t)blockIdx.x * WPB + wv;
    const int64_t gfirst = (wave / team) * team * gpw + wave % team;
    if (gfirst >= ngran) return;
    int64_t c = (ngran - gfirst + team - 1) / team;
    const int cnt = (int)(c < gpw ? c : gpw);
    uint4 va[4], vb[4];
    uint32_t acc = 0;
    load_granule(va, data, gfirst * 4096, lane);
    for (int i = 0; i < cnt; i += 2) {
        const int64_t g = gfirst + (int64_t)i * team;
        const int64_t gb = i + 1 < cnt ? g + team : g, ga = i + 2 < cnt ? g + 2 * team : gb;
        load_granule(vb, data, gb * 4096, lane);
        for (int j = 0; j < 4; ++j) acc ^= va[j].x ^ va[j].y ^ va[j].z ^ va[j].w;
        load_granule(va, data, ga * 4096, lane);
        if (i + 1 < cnt) for (int j = 0; j < 4; ++j) acc += vb[j].x ^ vb[j].y ^ vb[j].z ^ vb[j].w;
    }
    if (acc == 0x12345678u) out[lane] = acc;
}

template <int WPB> static int run(const uint8_t *d, int64_t ngran, int gpw, int64_t team, uint32_t *out) {
    int64_t waves = (ngran + gpw - 1) / gpw;
    if (team > waves) team = waves;
    waves = (waves + team - 1) / team * team;
    const unsigned nb = (unsigned)((waves + WPB - 1) / WPB);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_read<WPB>, dim3(nb), dim3(WPB * 64), 0, 0, d, ngran, gpw, team, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("waves/block %d  gpw %3d  team %8lld : %.3f ms  %.2f TB/s\n", WPB, gpw, (long long)team, best, ngran * 4096.0 / (best * 1e-3) / 1e12);
    return 0;
}

int main() {
    const int64_t ngran = 745000;              // 3.05 GB
    uint8_t *d; uint32_t *out;
    CK(hipMalloc((void **)&d, (size_t)ngran * 4096)); CK(hipMalloc((void **)&out, 1024));
    CK(hipMemset(d, 65, (size_t)ngran * 4096));
    for (int gpw : {1, 2, 4, 16, 32}) {
        run<4>(d, ngran, gpw, 1, out);
        if (gpw > 1) { run<4>(d, ngran, gpw, 1024, out); run<4>(d, ngran, gpw, 4096, out); run<4>(d, ngran, gpw, 8192, out); run<4>(d, ngran, gpw, 1 << 30, out); }
    }
    run<8>(d, ngran, 1, 1, out);
    run<8>(d, ngran, 16, 1, out);
    run<8>(d, ngran, 16, 8192, out);
    run<2>(d, ngran, 16, 1, out);
    run<1>(d, ngran, 16, 1, out);
    return 0;
}
