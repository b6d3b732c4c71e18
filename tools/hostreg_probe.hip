// hostreg_probe.hip -- can the pages of an index file leave the device STRAIGHT into the file's page cache (the mapping of a pre-sized
// tmpfs file registered with the runtime), without the pinned piece + host memcpy of fxi_image_out?  Times: fallocate, mmap,
// hipHostRegister, device-to-host copy into the mapping, hipHostUnregister, munmap -- for a file of GB gigabytes.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hostreg tools/hostreg_probe.hip && /tmp/hostreg [GB] [path]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const size_t n = (size_t)(argc > 1 ? atof(argv[1]) : 4.0) << 30;
    const char *path = argc > 2 ? argv[2] : "/dev/shm/fx_hostreg_probe.bin";
    unsigned char *d;
    CK(hipMalloc((void **)&d, n));
    CK(hipMemset(d, 7, n));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep) {
        unlink(path);
        const int fd = open(path, O_RDWR | O_CREAT, 0600);
        double t = now();
        if (fallocate(fd, 0, 0, (off_t)n) != 0) { perror("fallocate"); return 1; }
        const double t_fa = now() - t; t = now();
        void *m = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) { perror("mmap"); return 1; }
        const double t_map = now() - t; t = now();
        hipError_t e = hipHostRegister(m, n, hipHostRegisterDefault);
        const double t_reg = now() - t;
        if (e != hipSuccess) { printf("hipHostRegister: %s\n", hipGetErrorString(e)); return 1; }
        t = now();
        CK(hipMemcpy(m, d, n, hipMemcpyDeviceToHost));
        const double t_copy = now() - t; t = now();
        CK(hipHostUnregister(m));
        const double t_unreg = now() - t; t = now();
        munmap(m, n);
        const double t_unmap = now() - t;
        unsigned char b[8] = {0};
        (void)!pread(fd, b, 8, (off_t)(n / 2));
        close(fd);
        printf("%zu GiB: fallocate %.1f ms, mmap %.1f ms, hipHostRegister %.1f ms, D2H into the mapping %.1f ms (%.1f GB/s), hipHostUnregister %.1f ms, munmap %.1f ms; byte in the file %d\n",
               n >> 30, t_fa * 1e3, t_map * 1e3, t_reg * 1e3, t_copy * 1e3, n / t_copy / 1e9, t_unreg * 1e3, t_unmap * 1e3, (int)b[0]);
    }
    unlink(path);
    return 0;
}
