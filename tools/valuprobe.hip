// valuprobe.hip -- issue rate of the VALU instructions the scan / composition / FASTQ kernels are made of (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/valuprobe tools/valuprobe.hip && /tmp/valuprobe
// Two measurements per instruction:
//   (a) the device full (8 waves per SIMD, 64 independent ops in flight per wave): lane-ops per second, and from the
//       shader clock (s_memtime ticks over wall time, measured in the same run) the SIMD cycles one wave64 instruction takes;
//   (b) ONE wave per workgroup and one workgroup per CU, s_memtime around the loop: cycles per instruction of a lone wave
//       (the issue interval of a wave that has nobody to share its SIMD with).
// The `roofline_issue` block of bench.py reads its cycles-per-op from the committed output of this tool (profiles/r06_valuprobe.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP> __device__ __forceinline__ void body(uint32_t (&a)[8], uint32_t c1, uint32_t c2, int r) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (OP == 0) a[k] = (a[k] + a[(k + 1) & 7]) ;                                // v_add_u32
        if (OP == 1) a[k] = __builtin_amdgcn_perm(c2, c1, a[k]);                    // v_perm_b32
        if (OP == 2) a[k] = a[k] ^ a[(k + 1) & 7] ^ c1;                             // v_xor3 / v_bitop3
        if (OP == 3) a[k] = (a[k] & a[(k + 1) & 7]) | ((a[k] ^ a[(k + 1) & 7]) & c1);   // majority: v_bitop3
        if (OP == 4) a[k] = (a[k] >> 1) + r;                                        // shift + add (two instructions, or v_lshl_add)
        if (OP == 5) a[k] = a[k] - a[(k + 3) & 7];                                  // v_sub_u32
        if (OP == 6) a[k] = __builtin_amdgcn_udot4(a[k], c1, a[(k + 1) & 7], false);   // v_dot4_u32_u8
        if (OP == 7) a[k] = a[k] & a[(k + 1) & 7] & 0x20202020u;                    // bitop3 with literal
        if (OP == 8) a[k] = __builtin_amdgcn_mbcnt_hi(a[k], __builtin_amdgcn_mbcnt_lo(a[(k + 1) & 7], c1));   // v_mbcnt_lo + v_mbcnt_hi (two instructions)
        if (OP == 9) a[k] = a[k] + (uint32_t)__builtin_amdgcn_readlane((int)a[(k + 1) & 7], 5);   // v_readlane_b32 + v_add with an SGPR
        if (OP == 10) a[k] = __builtin_amdgcn_alignbit(a[k], a[(k + 1) & 7], c1 & 31u);   // v_alignbit_b32
        if (OP == 11) a[k] = __builtin_amdgcn_ubfe(a[k], a[(k + 1) & 7] & 31u, 5u);       // v_bfe_u32 (+ v_and)
        if (OP == 12) a[k] = a[k] * a[(k + 1) & 7];                                     // v_mul_lo_u32
        if (OP == 13) a[k] = a[k] > a[(k + 1) & 7] ? a[k] - c1 : a[k] + c2;          // v_cmp + v_cndmask (+ add, sub)
        if (OP == 14) a[k] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(a[(k + 1) & 7] << 2), (int)a[k]);   // ds_bpermute_b32 (LDS crossbar, no memory)
        if (OP == 15) a[k] = __builtin_popcount(a[k]) + a[(k + 1) & 7];              // v_bcnt_u32_b32 (popcount + add in one)
    }
}
template <int OP> __device__ __forceinline__ void body64(uint64_t (&b)[4], uint32_t c1, int r) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (OP == 100) b[k] = (b[k] >> (c1 & 63u)) + b[(k + 1) & 3];                // v_lshrrev_b64 + 64-bit add (v_add_co + v_addc)
        if (OP == 101) b[k] = b[k] + b[(k + 1) & 3] + (uint64_t)r;                  // 64-bit adds only
        if (OP == 102) b[k] = (b[k] << (c1 & 63u)) | b[(k + 1) & 3];                // v_lshlrev_b64 + two v_or
    }
}

template <int OP>
__global__ __launch_bounds__(256) void k_probe(uint32_t *out, uint32_t seed, int iters, unsigned long long *ticks) {
    uint32_t a[8];
    uint64_t b[4];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = seed * (k + 1) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) b[k] = ((uint64_t)a[k] << 32) | a[k + 4];
    uint32_t c1 = seed | 0x04080201u, c2 = seed ^ 0x10200000u;
    const unsigned long long w0 = wall_clock64();                 // the constant 100 MHz clock (hipDeviceAttributeWallClockRate)
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (OP < 100) body<OP>(a, c1, c2, r);
            else body64<OP>(b, c1, r);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s ^= a[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) s ^= (uint32_t)b[k] ^ (uint32_t)(b[k] >> 32);
    if (s == 0x12345678u) out[threadIdx.x] = s;
    if (ticks && threadIdx.x == 0) { ticks[blockIdx.x] = t1 - t0; ticks[4096 + blockIdx.x] = w1 - w0; }
}

static double g_mhz = 0;          // shader clock under load: s_memtime ticks per microsecond, from the full-device run of v_add_u32

template <int OP> static int run(const char *name, int per_iter /* wave64 instructions the body compiles to, per a[k] */) {
    uint32_t *d; CK(hipMalloc((void **)&d, 4096));
    unsigned long long *dt; CK(hipMalloc((void **)&dt, 8 * 8192));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000, blocks = 256 * 8;
    const int n_inner = OP < 100 ? 64 : 32;                   // a[k] updates per iteration
    float best = 1e9f;
    std::vector<unsigned long long> h(blocks), hw(blocks);
    double ticks_full = 0, wall_full = 0, wall_lone = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_probe<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u + rep, iters, dt);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) {
            best = ms;
            CK(hipMemcpy(h.data(), dt, 8 * blocks, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hw.data(), dt + 4096, 8 * blocks, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end()); std::sort(hw.begin(), hw.end());
            ticks_full = (double)h[blocks / 2]; wall_full = (double)hw[blocks / 2];
        }
    }
    // a lone wave: 256 workgroups of 64 threads
    float lone_ms = 1e9f;
    double ticks_lone = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_probe<OP>, dim3(256), dim3(64), 0, 0, d, 777u + rep, iters, dt);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < lone_ms) {
            lone_ms = ms;
            CK(hipMemcpy(h.data(), dt, 8 * 256, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hw.data(), dt + 4096, 8 * 256, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.begin() + 256); std::sort(hw.begin(), hw.begin() + 256);
            ticks_lone = (double)h[128]; wall_lone = (double)hw[128];
        }
    }
    const double upd = (double)iters * n_inner;                           // updates per wave
    const double lane_ops = (double)blocks * 256 * upd;
    // full device: 8 waves per SIMD; SIMD cycles per wave64 update = ticks of one wave / (updates x 8 waves sharing the SIMD) x (shader clock / memtime clock)
    // wall_* are ticks of the 100 MHz clock over one wave's loop: 10 ns each
    printf("%-34s full: %7.3f ms %6.2f T lane-upd/s, %5.2f ns/upd/SIMD (%.0f memtime ticks per us) | lone wave: %6.2f ns/upd, %5.2f ticks/upd (%.0f per us) (%d instr/upd)\n", name, best,
           lane_ops / (best * 1e-3) / 1e12, wall_full * 10.0 / (upd * 8.0), ticks_full / (wall_full * 0.01), wall_lone * 10.0 / upd, ticks_lone / upd,
           ticks_lone / (wall_lone * 0.01), per_iter);
    if (OP == 0) g_mhz = ticks_full / (best * 1e3);
    (void)hipFree(d); (void)hipFree(dt);
    return 0;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    int wall_khz = 0; (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("device %s, %d CUs, clockRate %d kHz, wallClockRate %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, wall_khz);
    printf("columns: full device = 2048 workgroups x 256 threads (8 waves per SIMD); 'ticks' are s_memtime ticks; an 'upd' is one a[k] = f(a[k], ...) of the listed body\n");
    run<0>("v_add_u32", 1);
    printf("s_memtime ticks per microsecond under load: %.1f (the clock the tick columns are in)\n", g_mhz);
    run<1>("v_perm_b32", 1);
    run<2>("xor3 (bitop3)", 1);
    run<3>("majority (bitop3)", 1);
    run<4>("lshr + add", 2);
    run<5>("v_sub_u32", 1);
    run<6>("v_dot4_u32_u8", 1);
    run<7>("and3 with literal (bitop3)", 1);
    run<8>("v_mbcnt_lo + v_mbcnt_hi", 2);
    run<9>("v_readlane + v_add(sgpr)", 2);
    run<10>("v_alignbit_b32", 1);
    run<11>("v_bfe_u32 (+and)", 2);
    run<12>("v_mul_lo_u32", 1);
    run<13>("cmp + cndmask + add + sub", 4);
    run<14>("ds_bpermute_b32 (+shift)", 2);
    run<15>("v_bcnt_u32_b32", 1);
    run<100>("lshr_b64 + add64", 3);
    run<101>("add64 + add64", 4);
    run<102>("lshl_b64 + or64", 3);
    return 0;
}
