// valuprobe.hip -- issue rate of the VALU instructions k_fasta_comp is made of (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/valuprobe tools/valuprobe.hip && /tmp/valuprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k_probe(uint32_t *out, uint32_t seed, int iters) {
    uint32_t a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = seed * (k + 1) + threadIdx.x;
    uint32_t c1 = seed | 0x04080201u, c2 = seed ^ 0x10200000u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (OP == 0) a[k] = a[k] & (c1 + r);                                        // v_and_b32
                if (OP == 1) a[k] = __builtin_amdgcn_perm(c2, c1, a[k]);                    // v_perm_b32
                if (OP == 2) a[k] = a[k] ^ a[(k + 1) & 7] ^ c1;                             // v_xor3 / v_bitop3
                if (OP == 3) a[k] = (a[k] & a[(k + 1) & 7]) | ((a[k] ^ a[(k + 1) & 7]) & c1);   // majority: v_bitop3
                if (OP == 4) a[k] = (a[k] >> 1) + r;                                        // shift + add
                if (OP == 5) a[k] = a[k] - a[(k + 3) & 7];                                  // v_sub_u32
                if (OP == 6) a[k] = __builtin_amdgcn_udot4(a[k], c1, a[(k + 1) & 7], false);   // v_dot4_u32_u8
                if (OP == 7) a[k] = a[k] & a[(k + 1) & 7] & 0x20202020u;                    // bitop3 with literal
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s ^= a[k];
    if (s == 0x12345678u) out[threadIdx.x] = s;
}

template <int OP> static int run(const char *name) {
    uint32_t *d; CK(hipMalloc((void **)&d, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000, blocks = 256 * 8;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_probe<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u + rep, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double ops = (double)blocks * 256 * iters * 64.0;
    printf("%-28s %8.3f ms  %6.2f T lane-ops/s\n", name, best, ops / (best * 1e-3) / 1e12);
    hipFree(d);
    return 0;
}

int main() {
    run<0>("v_and_b32");
    run<1>("v_perm_b32");
    run<2>("xor3 (bitop3)");
    run<3>("majority (bitop3)");
    run<4>("lshr + add");
    run<5>("v_sub_u32");
    run<6>("v_dot4_u32_u8");
    run<7>("and3 with literal (bitop3)");
    return 0;
}
