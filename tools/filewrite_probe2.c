// tools/filewrite_probe2.c -- schedules for getting N bytes of finished pages into ONE tmpfs file (the .fxi copy-out):
//   A  fallocate thread running ahead + T mmap writers, nobody waits            (what fx_fxi_dev_write did first)
//   B  fallocate everything, then T mmap writers
//   E  one pwrite thread from the front + T mmap writers from the back, no fallocate
//   F  fallocate thread + T mmap writers that wait until their piece has been allocated
//   G  K fallocate'd... (none)   P  T pwrite threads, each on its own contiguous share, big calls
//   Q  (round 6) fallocate everything, then the page-table entries made ahead of the writers: madvise(MADV_POPULATE_WRITE) by
//      K threads on K shares of the mapping, then T mmap writers -- each stage timed (the copy alone = what is left of the
//      copy-out when the entries were made while the input was still being staged)
//   R  as Q without the fallocate: the populate allocates the pages as it maps them (one pass over the file instead of two)
//   V  as B, but the file is mapped in SEPARATE mappings of `step` MiB (a page between them, so that the kernel keeps them apart) and a
//      writer takes a whole mapping at a time: no two threads fault on the same VMA (its lock is one cache line for all of them)
//   W  V without the fallocate: the writers' first touches allocate the pages (in separate mappings: does the allocation spread over threads then?)
// usage: filewrite_probe2 <path> <MiB> <schedule A|B|E|F|P|Q|R|V|W> <T> [piece KiB = 8192] [falloc step MiB = 128] [K = 4] [first cpu, cpus: bind every thread to that range]
#define _GNU_SOURCE
#include <sched.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
static size_t N, PIECE = 8u << 20, STEP = 128u << 20, SRC = 512u << 20;
static int T, fd; static char sched; static char *src, *map;
static atomic_size_t cursor, back_cursor, allocated;
static void *falloc_thread(void *a) {
    for (size_t o = 0; o < N; o += STEP) { size_t len = N - o < STEP ? N - o : STEP; if (fallocate(fd, 0, o, len)) { perror("fallocate"); break; } atomic_store(&allocated, o + len); }
    atomic_store(&allocated, N); return 0; }
static void *mmap_writer(void *a) {
    for (;;) { size_t off = atomic_fetch_add(&cursor, PIECE); if (off >= N) return 0; size_t len = N - off < PIECE ? N - off : PIECE;
        if (sched == 'F') while (atomic_load(&allocated) < off + len) sched_yield();
        memcpy(map + off, src + off % (SRC - PIECE), len); } }
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static int K = 4;
static void *populate_share(void *a) { long t = (long)a; size_t lo = (N / K * t) & ~4095ul, hi = t == K - 1 ? N : (N / K * (t + 1)) & ~4095ul;
    if (madvise(map + lo, hi - lo, MADV_POPULATE_WRITE)) perror("madvise(MADV_POPULATE_WRITE)"); return 0; }
static char **vmap; static size_t nvmap;
static void *vma_writer(void *a) {
    for (;;) { size_t k = atomic_fetch_add(&cursor, 1); if (k >= nvmap) return 0; size_t off = k * STEP, len = N - off < STEP ? N - off : STEP;
        for (size_t o = 0; o < len; o += PIECE) { size_t l = len - o < PIECE ? len - o : PIECE; memcpy(vmap[k] + o, src + (off + o) % (SRC - PIECE), l); } } }
static size_t split;   // E: pwrite thread takes [0, ...) upward, mmap writers take pieces downward from N; they meet
static void *pw_front(void *a) {
    for (;;) { size_t off = atomic_fetch_add(&cursor, PIECE); if (off >= N || off + PIECE > N - atomic_load(&back_cursor)) { return 0; }
        size_t done = 0; while (done < PIECE) { ssize_t w = pwrite(fd, src + off % (SRC - PIECE) + done, PIECE - done, off + done); if (w <= 0) { perror("pwrite"); exit(1); } done += w; } } }
static void *mmap_back(void *a) {
    for (;;) { size_t b = atomic_fetch_add(&back_cursor, PIECE) + PIECE; if (b > N) return 0; size_t off = N - b; if (off < atomic_load(&cursor)) return 0;
        memcpy(map + off, src + off % (SRC - PIECE), PIECE); } }
static void *pw_share(void *a) { long t = (long)a; size_t lo = N / T * t, hi = t == T - 1 ? N : N / T * (t + 1);
    for (size_t off = lo; off < hi; off += PIECE) { size_t len = hi - off < PIECE ? hi - off : PIECE, done = 0;
        while (done < len) { ssize_t w = pwrite(fd, src + off % (SRC - PIECE) + done, len - done, off + done); if (w <= 0) { perror("pwrite"); exit(1); } done += w; } } return 0; }
int main(int c, char **v) {
    const char *path = v[1]; N = ((size_t)atol(v[2]) << 20) / (8u << 20) * (8u << 20); sched = v[3][0]; T = atoi(v[4]);
    if (c > 5) PIECE = (size_t)atol(v[5]) << 10; if (c > 6) STEP = (size_t)atol(v[6]) << 20; if (c > 7) K = atoi(v[7]);
    if (c > 9) { cpu_set_t cs; CPU_ZERO(&cs); for (int i = 0; i < atoi(v[9]); ++i) CPU_SET(atoi(v[8]) + i, &cs); if (sched_setaffinity(0, sizeof cs, &cs)) perror("sched_setaffinity"); }   /* inherited by every thread */
    src = malloc(SRC); memset(src, 7, SRC);
    unlink(path); fd = open(path, O_RDWR | O_CREAT, 0644);
    double t0 = now();
    if (ftruncate(fd, N)) perror("ftruncate");
    map = mmap(0, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); if (map == MAP_FAILED) { perror("mmap"); return 1; }
    pthread_t th[300]; int nt = 0; double tf = 0;
    if (sched == 'B') { falloc_thread(0); tf = now() - t0; }
    double tp = 0, tq0 = 0;
    if (sched == 'V' || sched == 'W') {
        if (sched == 'V') falloc_thread(0);
        tf = now() - t0;
        nvmap = (N + STEP - 1) / STEP; vmap = malloc(nvmap * sizeof *vmap);
        char *area = mmap(0, nvmap * (STEP + 4096), PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0); if (area == MAP_FAILED) { perror("mmap area"); return 1; }
        for (size_t k = 0; k < nvmap; ++k) { size_t off = k * STEP, len = N - off < STEP ? N - off : STEP;
            vmap[k] = mmap(area + k * (STEP + 4096), len, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, fd, off); if (vmap[k] == MAP_FAILED) { perror("mmap piece"); return 1; } }
        tq0 = now();
        for (int t = 0; t < T; ++t) pthread_create(&th[nt++], 0, vma_writer, 0);
    } else
    if (sched == 'Q' || sched == 'R') { if (sched == 'Q') falloc_thread(0); tf = now() - t0; double a = now(); pthread_t pt[64]; for (long t = 0; t < K; ++t) pthread_create(&pt[t], 0, populate_share, (void *)t);
        for (int t = 0; t < K; ++t) pthread_join(pt[t], 0); tp = now() - a; tq0 = now(); }
    if (sched == 'A' || sched == 'F') pthread_create(&th[nt++], 0, falloc_thread, 0);
    if (sched == 'E') { pthread_create(&th[nt++], 0, pw_front, 0); for (int t = 0; t < T; ++t) pthread_create(&th[nt++], 0, mmap_back, 0); }
    else if (sched == 'P') for (long t = 0; t < T; ++t) pthread_create(&th[nt++], 0, pw_share, (void *)t);
    else if (sched != 'V' && sched != 'W') for (int t = 0; t < T; ++t) pthread_create(&th[nt++], 0, mmap_writer, 0);
    for (int t = 0; t < nt; ++t) pthread_join(th[t], 0);
    double t1 = now();
    if (sched == 'V' || sched == 'W') printf("schedule V/W: fallocate %.3f s, %zu mappings of %zu MiB, copy with %d threads %.3f s = %.2f GB/s\n", tf, nvmap, STEP >> 20, T, t1 - tq0, N / 1e9 / (t1 - tq0));
    if (sched == 'Q' || sched == 'R') printf("schedule Q/R: fallocate %.3f s, populate with %d threads %.3f s, copy with %d threads %.3f s = %.2f GB/s\n", tf, K, tp, T, t1 - tq0, N / 1e9 / (t1 - tq0));
    printf("schedule %c T=%d piece=%zuK step=%zuM: %.3f s (fallocate first: %.3f) -> %.2f GB/s\n", sched, T, PIECE >> 10, STEP >> 20, t1 - t0, tf, N / 1e9 / (t1 - t0));
    munmap(map, N); close(fd); unlink(path); return 0; }
