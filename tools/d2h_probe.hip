// d2h_probe.hip -- what the link gives device -> host when T host threads each keep two pinned pieces in flight on a stream of their own
// (the shape of the .fxi copy-out, fxi_image_out, without the stores into the file).  Not part of the product.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/d2h_probe tools/d2h_probe.hip -lpthread && /tmp/d2h_probe [GiB = 4]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
int main(int argc, char **argv) {
    const size_t total = (size_t)(argc > 1 ? atof(argv[1]) : 4.0) << 30;
    uint8_t *d = nullptr;
    if (hipMalloc((void **)&d, total) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(d, 1, total);
    (void)hipDeviceSynchronize();
    for (size_t piece_mb : {8, 32}) for (int T : {2, 4, 8, 16, 24}) for (int dir = 0; dir < 2; ++dir) {
        const size_t piece = piece_mb << 20, npieces = total / piece;
        std::vector<std::thread> th;
        std::vector<uint8_t *> pins((size_t)T * 2, nullptr);
        for (auto &p : pins) if (hipHostMalloc((void **)&p, piece, hipHostMallocDefault) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; }
        std::vector<hipStream_t> st((size_t)T);
        for (auto &s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        const auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t]() {
                hipEvent_t ev[2];
                (void)hipEventCreateWithFlags(&ev[0], hipEventDisableTiming); (void)hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
                bool used[2] = {false, false};
                int slot = 0;
                for (size_t k = (size_t)t; k < npieces; k += (size_t)T, slot ^= 1) {
                    if (used[slot]) (void)hipEventSynchronize(ev[slot]);
                    if (dir == 0) (void)hipMemcpyAsync(pins[(size_t)t * 2 + slot], d + k * piece, piece, hipMemcpyDeviceToHost, st[(size_t)t]);
                    else (void)hipMemcpyAsync(d + k * piece, pins[(size_t)t * 2 + slot], piece, hipMemcpyHostToDevice, st[(size_t)t]);
                    (void)hipEventRecord(ev[slot], st[(size_t)t]);
                    used[slot] = true;
                }
                (void)hipStreamSynchronize(st[(size_t)t]);
                (void)hipEventDestroy(ev[0]); (void)hipEventDestroy(ev[1]);
            });
        for (auto &x : th) x.join();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%s  %2d lanes, pieces of %2zu MiB: %.1f GB/s\n", dir == 0 ? "device -> host" : "host -> device", T, piece_mb, (double)(npieces * piece) / 1e9 / s);
        for (auto &s2 : st) (void)hipStreamDestroy(s2);
        for (auto p : pins) (void)hipHostFree(p);
    }
    return 0;
}
