/* libfxsynth.so -- host helpers of the SYNTHETIC workloads (pyfastx_amd/synth.py): the C4 inputs of bench.py and of the
 * tests are the C2 bytes BGZF-framed / as one gzip stream, and making them is set-up, not the measured path.  zlib from a
 * Python thread pool deflated 3 GB in 15 s on a box with 256 hardware threads (every deflate call of 64 KiB goes back
 * through the interpreter lock); here the members / pieces are handed out to plain threads: ~1 s.
 *
 * Same bytes as synth.bgzf_compress / synth.gzip_single_stream (level, raw deflate, window 15, memLevel 8, default strategy:
 * what zlib.compressobj(level, DEFLATED, -15) asks for) -- tests/test_host_logic.py compares them.
 * Not part of libfxgpu.so and not used by the product: nothing here reads a FASTA/FASTQ file or answers a query. */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

typedef struct {
    const uint8_t *src;
    int64_t n, nitem, item, slot;        /* item: input bytes per member / piece; slot: bytes of dst set aside for each */
    uint8_t *dst;
    int64_t *len;                        /* compressed bytes of every item (framing included) */
    uint32_t *crc;                       /* CRC-32 of every item's input */
    int level, bgzf;
    volatile int64_t next;
    volatile int err;
} job_t;

/* one z_stream per thread, reset for every item: deflateInit2 allocates ~260 KiB, and 192 threads doing that per 64 KiB
 * member spend their time in mmap / munmap (3 GB took 15 s that way, whatever the language above) */
static int deflate_raw(z_stream *z, const uint8_t *in, int64_t n, uint8_t *out, int64_t cap, int flush, int64_t *got) {
    if (deflateReset(z) != Z_OK) return -1;
    int64_t ip = 0, op = 0;
    int rc = Z_OK;
    do {                                  /* avail_in / avail_out are 32-bit: feed at most 1 GiB at a time */
        const int64_t in_now = n - ip > (1 << 30) ? (1 << 30) : n - ip;
        const int last = ip + in_now == n;
        z->next_in = (Bytef *)(in + ip); z->avail_in = (uInt)in_now;
        do {
            const int64_t room = cap - op > (1 << 30) ? (1 << 30) : cap - op;
            if (room <= 0) return -2;
            z->next_out = out + op; z->avail_out = (uInt)room;
            rc = deflate(z, last ? flush : Z_NO_FLUSH);
            op += room - z->avail_out;
            if (rc == Z_STREAM_ERROR) return -3;
        } while (z->avail_out == 0);
        ip += in_now;
    } while (ip < n);
    *got = op;
    return 0;
}

static void *worker(void *arg) {
    job_t *j = (job_t *)arg;
    z_stream z;
    memset(&z, 0, sizeof z);
    if (deflateInit2(&z, j->level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { j->err = 3; return 0; }
    for (;;) {
        const int64_t i = __sync_fetch_and_add(&j->next, 1);
        if (i >= j->nitem || j->err) break;
        const int64_t a = i * j->item, m = a + j->item <= j->n ? j->item : j->n - a;
        uint8_t *o = j->dst + i * j->slot;
        int64_t got = 0;
        if (j->bgzf) {                    /* SAM spec 4.1: 18-byte header with the 'BC' subfield, deflate, CRC-32, ISIZE */
            static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
            if (deflate_raw(&z, j->src + a, m, o + 18, j->slot - 26, Z_FINISH, &got)) { j->err = 1; break; }
            if (got + 25 > 65535) { j->err = 2; break; }
            memcpy(o, hdr, 16);
            const uint32_t bsize = (uint32_t)(got + 25), c = (uint32_t)crc32(crc32(0L, Z_NULL, 0), j->src + a, (uInt)m), isz = (uint32_t)m;
            o[16] = bsize & 0xFF; o[17] = bsize >> 8;
            memcpy(o + 18 + got, &c, 4); memcpy(o + 22 + got, &isz, 4);
            j->len[i] = got + 26;
        } else {                          /* one piece of a single stream, pigz-style: a full flush ends all but the last */
            const int last = i + 1 == j->nitem;
            if (deflate_raw(&z, j->src + a, m, o, j->slot, last ? Z_FINISH : Z_FULL_FLUSH, &got)) { j->err = 1; break; }
            uint32_t c = (uint32_t)crc32(0L, Z_NULL, 0);
            for (int64_t p = 0; p < m; p += 1 << 30) c = (uint32_t)crc32(c, j->src + a + p, (uInt)(m - p > (1 << 30) ? (1 << 30) : m - p));
            j->crc[i] = c;
            j->len[i] = got;
        }
    }
    deflateEnd(&z);
    return 0;
}

static int run(job_t *j, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    if ((int64_t)nthreads > j->nitem) nthreads = (int)(j->nitem ? j->nitem : 1);
    pthread_t th[256];
    int started = 0;
    for (int t = 0; t < nthreads; ++t)
        if (pthread_create(&th[t], 0, worker, j) == 0) ++started; else break;
    if (!started) worker(j);
    for (int t = 0; t < started; ++t) pthread_join(th[t], 0);
    return j->err;
}

/* bytes a caller must provide for fxs_bgzf_compress / fxs_gzip_stream of n input bytes */
int64_t fxs_bgzf_bound(int64_t n, int block) {
    const int64_t nmem = (n + block - 1) / block;
    return nmem * ((int64_t)compressBound((uLong)block) + 64) + 28;
}
int64_t fxs_gzip_bound(int64_t n, int64_t piece) {
    const int64_t np = n ? (n + piece - 1) / piece : 1;
    return np * ((int64_t)compressBound((uLong)piece) + 64) + 18;
}

/* `src` BGZF-framed into dst (members of `block` input bytes + the 28-byte EOF member) -> bytes written, < 0 on error */
int64_t fxs_bgzf_compress(const uint8_t *src, int64_t n, uint8_t *dst, int64_t cap, int block, int level, int nthreads) {
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (block < 1 || block > 65280 || cap < fxs_bgzf_bound(n, block)) return -1;
    job_t j;
    memset(&j, 0, sizeof j);
    j.src = src; j.n = n; j.item = block; j.nitem = (n + block - 1) / block; j.slot = (int64_t)compressBound((uLong)block) + 64;
    j.dst = dst; j.level = level; j.bgzf = 1;
    j.len = (int64_t *)malloc(sizeof(int64_t) * (size_t)(j.nitem + 1));
    if (!j.len) return -2;
    int64_t w = -3;
    if (!run(&j, nthreads)) {
        w = 0;
        for (int64_t i = 0; i < j.nitem; ++i) { memmove(dst + w, dst + i * j.slot, (size_t)j.len[i]); w += j.len[i]; }   /* a member never grows past its slot: left to right in place */
        memcpy(dst + w, eof, 28); w += 28;
    }
    free(j.len);
    return w;
}

/* ONE gzip member holding src: header, the pieces' deflate data one after the other, CRC-32 and ISIZE -> bytes written */
int64_t fxs_gzip_stream(const uint8_t *src, int64_t n, uint8_t *dst, int64_t cap, int64_t piece, int level, int nthreads) {
    static const uint8_t hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff};
    if (piece < 1 || cap < fxs_gzip_bound(n, piece)) return -1;
    job_t j;
    memset(&j, 0, sizeof j);
    j.src = src; j.n = n; j.item = piece; j.nitem = n ? (n + piece - 1) / piece : 1; j.slot = (int64_t)compressBound((uLong)piece) + 64;
    j.dst = dst + 10; j.level = level; j.bgzf = 0;
    j.len = (int64_t *)malloc(sizeof(int64_t) * (size_t)(j.nitem + 1));
    j.crc = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(j.nitem + 1));
    int64_t w = -3;
    if (j.len && j.crc && !run(&j, nthreads)) {
        memcpy(dst, hdr, 10);
        w = 10;
        uLong c = crc32(0L, Z_NULL, 0);
        for (int64_t i = 0; i < j.nitem; ++i) {
            memmove(dst + w, j.dst + i * j.slot, (size_t)j.len[i]); w += j.len[i];
            const int64_t m = (i + 1) * piece <= n ? piece : n - i * piece;
            c = i ? crc32_combine(c, j.crc[i], (z_off_t)m) : j.crc[0];
        }
        const uint32_t c32 = (uint32_t)c, isz = (uint32_t)(n & 0xFFFFFFFFll);
        memcpy(dst + w, &c32, 4); memcpy(dst + w + 4, &isz, 4); w += 8;
    }
    free(j.len); free(j.crc);
    return w;
}
