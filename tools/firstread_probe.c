// tools/firstread_probe.c -- is the FIRST read of a freshly written tmpfs file slower than the next ones, with no device in the picture?
// One thread writes N MiB with write() (1 GiB calls, as bench.py writes its inputs), then T threads pread() the file in 8 MiB pieces into their own
// buffers, three passes, each timed.  FIRSTREAD_MMAP=1: the readers copy out of mappings of the file instead (one mapping per 8 MiB piece: read faults with
// fault-around, no pread).  usage: firstread_probe <path> <MiB> <T> [first cpu, cpus: bind every thread to that range] [writer: 0 same cpus, 1 = unbound]
#define _GNU_SOURCE
#include <fcntl.h>
#include <sys/mman.h>
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
static size_t N, PIECE = 8u << 20;
static int fd;
static atomic_size_t cursor;
static int use_mmap;
static void *reader(void *a) {
    char *buf = malloc(PIECE); memset(buf, 1, PIECE);
    for (;;) { size_t off = atomic_fetch_add(&cursor, PIECE); if (off >= N) break; size_t len = N - off < PIECE ? N - off : PIECE, done = 0;
        if (use_mmap) { char *m = mmap(0, len, PROT_READ, MAP_SHARED, fd, off); if (m == MAP_FAILED) { perror("mmap"); exit(1); } memcpy(buf, m, len); munmap(m, len); continue; }
        while (done < len) { ssize_t r = pread(fd, buf + done, len - done, off + done); if (r <= 0) { perror("pread"); exit(1); } done += r; } }
    free(buf); return 0; }
int main(int c, char **v) {
    const char *path = v[1]; N = (size_t)atol(v[2]) << 20; int T = atoi(v[3]);
    use_mmap = getenv("FIRSTREAD_MMAP") && atoi(getenv("FIRSTREAD_MMAP"));
    cpu_set_t cs; int bind = c > 5;
    if (bind) { CPU_ZERO(&cs); for (int i = 0; i < atoi(v[5]); ++i) CPU_SET(atoi(v[4]) + i, &cs); }
    if (bind && !(c > 6 && atoi(v[6]))) sched_setaffinity(0, sizeof cs, &cs);
    size_t W = 1u << 30; char *src = malloc(W); memset(src, 65, W);
    unlink(path); fd = open(path, O_RDWR | O_CREAT, 0644);
    double t0 = now();
    for (size_t o = 0; o < N; o += W) { size_t len = N - o < W ? N - o : W, done = 0; while (done < len) { ssize_t w = write(fd, src + done, len - done); if (w <= 0) { perror("write"); return 1; } done += w; } }
    printf("written %.1f GB in %.2f s by one thread\n", N / 1e9, now() - t0);
    if (bind) sched_setaffinity(0, sizeof cs, &cs);
    for (int pass = 0; pass < 3; ++pass) {
        atomic_store(&cursor, 0); pthread_t th[256]; double a = now();
        for (int t = 0; t < T; ++t) pthread_create(&th[t], 0, reader, 0);
        for (int t = 0; t < T; ++t) pthread_join(th[t], 0);
        printf("  read pass %d with %d threads: %.3f s = %.1f GB/s\n", pass + 1, T, now() - a, N / 1e9 / (now() - a));
    }
    close(fd); unlink(path); return 0; }
