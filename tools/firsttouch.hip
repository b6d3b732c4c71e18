// firsttouch.hip -- what the FIRST host-to-device copy into freshly allocated device memory costs, and whether touching the
// block from a kernel first takes that cost away (the first Fastq(path) of a process stages at a third of the rate).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/firsttouch tools/firsttouch.hip && /tmp/firsttouch [GB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_touch(uint4 *p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0, 0, 0, 0);
}
__global__ void k_touch_sparse(unsigned char *p, size_t n, size_t step) {     // one byte per `step`
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * step; i < n; i += (size_t)gridDim.x * 256 * step) p[i] = 0;
}
static int copy_all(unsigned char *d, unsigned char *pin[4], size_t piece, size_t n, hipStream_t st[4], const char *what) {
    const double t0 = now();
    for (size_t off = 0, k = 0; off < n; off += piece, ++k) CK(hipMemcpyAsync(d + off, pin[k & 3], piece < n - off ? piece : n - off, hipMemcpyHostToDevice, st[k & 3]));
    for (int i = 0; i < 4; ++i) CK(hipStreamSynchronize(st[i]));
    const double t = now() - t0;
    printf("  %-52s %7.1f ms  %5.1f GB/s\n", what, t * 1e3, n / t / 1e9);
    return 0;
}
int main(int argc, char **argv) {
    const size_t n = (size_t)(argc > 1 ? atof(argv[1]) : 16.0) << 30, piece = 64u << 20;
    unsigned char *pin[4]; hipStream_t st[4];
    for (int i = 0; i < 4; ++i) { CK(hipHostMalloc((void **)&pin[i], piece)); memset(pin[i], 65 + i, piece); CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); }
    unsigned char *d;
    for (int mode = 0; mode < 5; ++mode) {
        double t0 = now();
        CK(hipMalloc((void **)&d, n));
        printf("mode %d: hipMalloc(%zu GiB) %.1f ms\n", mode, n >> 30, (now() - t0) * 1e3);
        t0 = now();
        if (mode == 1) { hipLaunchKernelGGL(k_touch, dim3(4096), dim3(256), 0, st[0], (uint4 *)d, n / 16); CK(hipStreamSynchronize(st[0])); printf("  kernel writes every byte                             %7.1f ms\n", (now() - t0) * 1e3); }
        if (mode == 2) { hipLaunchKernelGGL(k_touch_sparse, dim3(1024), dim3(256), 0, st[0], d, n, (size_t)4096); CK(hipStreamSynchronize(st[0])); printf("  kernel writes one byte per 4 KiB                     %7.1f ms\n", (now() - t0) * 1e3); }
        if (mode == 3) { CK(hipMemsetAsync(d, 0, n, st[0])); CK(hipStreamSynchronize(st[0])); printf("  hipMemsetAsync of the block                          %7.1f ms\n", (now() - t0) * 1e3); }
        if (mode == 4) { hipLaunchKernelGGL(k_touch_sparse, dim3(1024), dim3(256), 0, st[0], d, n, (size_t)(2u << 20)); CK(hipStreamSynchronize(st[0])); printf("  kernel writes one byte per 2 MiB                     %7.1f ms\n", (now() - t0) * 1e3); }
        if (copy_all(d, pin, piece, n, st, "first copy of the whole block (4 streams, 64 MiB pieces)")) return 1;
        if (copy_all(d, pin, piece, n, st, "second copy")) return 1;
        t0 = now();
        CK(hipFree(d));
        printf("  hipFree %.1f ms\n", (now() - t0) * 1e3);
    }
    return 0;
}
