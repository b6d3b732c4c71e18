// Probe: can the host post a request line straight into DEVICE memory (fine-grained allocation, CPU stores through the
// PCIe BAR) so that a resident kernel polls its own memory instead of pinned host memory across the bus?  Round trip of
// a doorbell: host writes a sequence number, a one-wave kernel sees it and writes it back to pinned host memory.
// build: hipcc --offload-arch=gfx950 -O2 -o build/bar_probe tools/bar_probe.hip ; run: build/bar_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>

__global__ void k_echo(const unsigned long long *req, unsigned long long *ack, unsigned long long last) {
    unsigned long long seen = 0;
    for (;;) {
        const unsigned long long r = __hip_atomic_load(req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (r != seen) {
            seen = r;
            __hip_atomic_store(ack, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (r >= last) break;
        } else __builtin_amdgcn_s_sleep(1);
    }
}

static sigjmp_buf g_jb;
static void on_segv(int) { siglongjmp(g_jb, 1); }

static double run(volatile unsigned long long *req_host_view, const unsigned long long *req_dev, unsigned long long *ack, int n) {
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    *ack = 0;
    *req_host_view = 0;
    hipLaunchKernelGGL(k_echo, dim3(1), dim3(1), 0, s, req_dev, ack, (unsigned long long)n);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 1; i <= n; ++i) {
        __atomic_store_n(req_host_view, (unsigned long long)i, __ATOMIC_RELEASE);
        while (__atomic_load_n(ack, __ATOMIC_ACQUIRE) != (unsigned long long)i) {}
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
    hipStreamSynchronize(s);
    hipStreamDestroy(s);
    return us;
}

int main() {
    const int n = 20000;
    unsigned long long *ack = nullptr, *req_pin = nullptr, *req_dev = nullptr;
    hipHostMalloc((void **)&ack, 64, hipHostMallocDefault);
    hipHostMalloc((void **)&req_pin, 64, hipHostMallocDefault);
    printf("request in pinned host memory : %.2f us per round trip\n", run(req_pin, req_pin, ack, n));
    for (int flavour = 0; flavour < 2; ++flavour) {
        hipError_t e = flavour == 0 ? hipExtMallocWithFlags((void **)&req_dev, 4096, hipDeviceMallocFinegrained) : hipMalloc((void **)&req_dev, 4096);
        if (e != hipSuccess) { printf("allocation %d failed: %s\n", flavour, hipGetErrorString(e)); continue; }
        hipMemset(req_dev, 0, 4096);
        hipDeviceSynchronize();
        struct sigaction sa, old;
        memset(&sa, 0, sizeof sa);
        sa.sa_handler = on_segv;
        sigaction(SIGSEGV, &sa, &old);
        struct sigaction oldb;
        sigaction(SIGBUS, &sa, &oldb);
        bool ok = false;
        if (sigsetjmp(g_jb, 1) == 0) { *(volatile unsigned long long *)req_dev = 0; ok = true; }
        sigaction(SIGSEGV, &old, nullptr);
        sigaction(SIGBUS, &oldb, nullptr);
        if (!ok) { printf("%s device memory: the host cannot store to it (SIGSEGV)\n", flavour == 0 ? "fine-grained" : "plain"); hipFree(req_dev); continue; }
        printf("request in %s device memory (host stores through the BAR): %.2f us per round trip\n", flavour == 0 ? "fine-grained" : "plain", run(req_dev, req_dev, ack, n));
        hipFree(req_dev);
    }
    return 0;
}
