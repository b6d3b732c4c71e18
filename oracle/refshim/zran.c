/*
 * zran.c -- TEST INFRASTRUCTURE ONLY (part of oracle/, never shipped or linked by the product).
 *
 * A work-alike of indexed_gzip v1.10.3's zran.c, which the reference downloads at build time
 * (setup.py:53-69) and which is absent from /root/reference.  Written from the published
 * description of the algorithm (zlib's examples/zran.c idea as indexed_gzip extends it):
 *
 *   - zran_build_index inflates the file once with Z_BLOCK and, at deflate block boundaries at
 *     least `spacing` bytes of output apart, records a point {offset of the next compressed byte,
 *     offset in the output, number of bits of the byte BEFORE it that still belong to the stream,
 *     the `window_size` bytes of output before it}; the point at the start of a gzip member (right
 *     behind its header) carries no window data;
 *   - zran_seek picks the LAST point at or before the wanted offset -- from whatever list the index
 *     holds, in particular the one pyfastx_gzip_index_import (util.c:542-726) swapped in --, and
 *     starts a RAW inflate there: inflateInit2(-15), inflatePrime(bits, byte >> (8 - bits)),
 *     inflateSetDictionary(window);
 *   - zran_read inflates forward from there, dropping the bytes before the wanted offset, walking
 *     over member trailers and headers of a multi-member file.
 *
 * So an index file written by the product is exercised through the reference's own import and seek
 * path: wrong offsets, bit counts or windows give wrong bytes or an inflate error.  Counters
 * (fxshim_stats, fxshim_point_hits) say how every seek was served; the tests read them through
 * ctypes from the compiled module.  What stays unpinned is only WHERE real zran would have placed
 * its points -- no reference test asserts that, and any placement at block boundaries is valid.
 */
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/stat.h>
#include "zran.h"

const char    ZRAN_INDEX_FILE_ID[5]   = {'G', 'Z', 'I', 'D', 'X'};
const uint8_t ZRAN_INDEX_FILE_VERSION = 1;

#define SHIM_INBUF  65536
#define SHIM_HITCAP 65536

typedef struct {
    int       fdno;
    z_stream  z;
    int       live;              /* z is initialised (raw inflate) */
    int       eof;               /* no further member */
    uint64_t  in_pos;            /* file offset of the next byte to put into inbuf */
    uint64_t  out_pos;           /* offset in the uncompressed stream of the next byte inflate produces */
    uint64_t  want;              /* where the next zran_read starts */
    uint8_t   inbuf[SHIM_INBUF];
    uint8_t   drop[32768];
} shim_t;

/* ---- counters (process-wide; the list of an index is swapped by the reference's import, so nothing is kept per point) */
static uint64_t g_seeks, g_from_point, g_from_start, g_continued, g_built_points, g_errors;
static uint32_t g_hits[SHIM_HITCAP];

void fxshim_stats(uint64_t out[6])
{
    out[0] = g_seeks; out[1] = g_from_point; out[2] = g_from_start; out[3] = g_continued;
    out[4] = g_built_points; out[5] = g_errors;
}
void fxshim_reset(void)
{
    g_seeks = g_from_point = g_from_start = g_continued = g_built_points = g_errors = 0;
    memset(g_hits, 0, sizeof g_hits);
}
/* hits of points [0, n): how many seeks started at each */
void fxshim_point_hits(uint32_t *out, uint32_t n)
{
    uint32_t i;
    for (i = 0; i < n; ++i) out[i] = i < SHIM_HITCAP ? g_hits[i] : 0;
}

/* ---- input */
static int shim_fill(shim_t *s)
{
    ssize_t r;
    if (s->z.avail_in) return 1;
    r = pread(s->fdno, s->inbuf, SHIM_INBUF, (off_t)s->in_pos);
    if (r <= 0) return 0;
    s->in_pos += (uint64_t)r;
    s->z.next_in = s->inbuf;
    s->z.avail_in = (uInt)r;
    return 1;
}
static uint64_t shim_consumed(const shim_t *s) { return s->in_pos - s->z.avail_in; }
static void shim_seek_in(shim_t *s, uint64_t off) { s->in_pos = off; s->z.avail_in = 0; s->z.next_in = s->inbuf; }
static int shim_byte(shim_t *s)
{
    if (!shim_fill(s)) return -1;
    s->z.avail_in--;
    return *s->z.next_in++;
}
/* RFC 1952 member header at the input position; 0: parsed, 1: no (further) member, -1: damaged */
static int shim_header(shim_t *s)
{
    int b0, b1, cm, flg, i, c;
    b0 = shim_byte(s);
    if (b0 < 0) return 1;
    b1 = shim_byte(s);
    if (b0 != 0x1f || b1 != 0x8b) return 1;          /* gzread: anything else after a trailer is ignored */
    cm = shim_byte(s); flg = shim_byte(s);
    if (cm != 8 || flg < 0) return -1;
    for (i = 0; i < 6; ++i) if (shim_byte(s) < 0) return -1;          /* MTIME, XFL, OS */
    if (flg & 4) {
        int lo = shim_byte(s), hi = shim_byte(s), n;
        if (lo < 0 || hi < 0) return -1;
        for (n = lo | (hi << 8); n > 0; --n) if (shim_byte(s) < 0) return -1;
    }
    if (flg & 8)  do { c = shim_byte(s); if (c < 0) return -1; } while (c);
    if (flg & 16) do { c = shim_byte(s); if (c < 0) return -1; } while (c);
    if (flg & 2)  { if (shim_byte(s) < 0 || shim_byte(s) < 0) return -1; }
    return 0;
}
static int shim_raw_init(shim_t *s)
{
    uint8_t *ni = s->z.next_in;
    uInt ai = s->z.avail_in;
    if (s->live) inflateEnd(&s->z);
    memset(&s->z, 0, sizeof s->z);
    s->live = inflateInit2(&s->z, -15) == Z_OK;
    s->z.next_in = ni; s->z.avail_in = ai;
    return s->live ? 0 : -1;
}

int zran_init(zran_index_t *index, FILE *fd, void *f, uint32_t spacing,
              uint32_t window_size, uint32_t readbuf_size, uint16_t flags)
{
    struct stat st;
    shim_t *s;
    memset(index, 0, sizeof(*index));
    index->fd = fd;
    index->f = f;
    index->spacing = spacing ? spacing : 1048576;
    index->window_size = window_size ? window_size : 32768;
    index->log_window_size = 15;
    index->readbuf_size = readbuf_size ? readbuf_size : 16384;
    index->flags = flags;
    index->size = 8;
    index->list = (zran_point_t *)calloc(index->size, sizeof(zran_point_t));
    if (fstat(fileno(fd), &st) == 0) index->compressed_size = (uint64_t)st.st_size;
    s = (shim_t *)calloc(1, sizeof(shim_t));
    if (!s || !index->list) return -1;
    s->fdno = fileno(fd);
    index->shim = s;
    return 0;
}

void zran_free(zran_index_t *index)
{
    uint32_t i;
    shim_t *s = (shim_t *)index->shim;
    if (index->list) {
        for (i = 0; i < index->npoints; ++i) free(index->list[i].data);
        free(index->list);
        index->list = NULL;
    }
    if (s) {
        if (s->live) inflateEnd(&s->z);
        free(s);
        index->shim = NULL;
    }
}

static int shim_add_point(zran_index_t *index, uint64_t cin, uint64_t cout, int bits, const uint8_t *ring, uint32_t left, int with_data)
{
    zran_point_t *p;
    const uint32_t W = index->window_size;
    if (index->npoints == index->size) {
        zran_point_t *nl = (zran_point_t *)realloc(index->list, sizeof(zran_point_t) * index->size * 2);
        if (!nl) return -1;
        memset(nl + index->size, 0, sizeof(zran_point_t) * index->size);
        index->list = nl;
        index->size *= 2;
    }
    p = index->list + index->npoints;
    p->cmp_offset = cin; p->uncmp_offset = cout; p->bits = (uint8_t)bits; p->data = NULL;
    if (with_data) {
        /* ring[0, W) is filled round and round; `left` bytes of the current lap are still free, so the oldest byte of the
         * last W sits at ring + W - left */
        p->data = (uint8_t *)malloc(W);
        if (!p->data) return -1;
        if (left) memcpy(p->data, ring + W - left, left);
        if (left < W) memcpy(p->data + left, ring, W - left);
    }
    index->npoints++;
    g_built_points++;
    return 0;
}

int zran_build_index(zran_index_t *index, uint64_t from, uint64_t until)
{
    shim_t *s = (shim_t *)index->shim;
    z_stream z;
    uint8_t *ring, *in;
    uint64_t base_in = 0, base_out = 0, fpos = 0, last = 0;
    const uint32_t W = index->window_size;
    int ret = 0, member_start = 1;
    (void)from; (void)until;
    if (index->npoints) return 0;                    /* built or imported already */
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, 47) != Z_OK) return -1;
    ring = (uint8_t *)calloc(1, W);
    in = (uint8_t *)malloc(SHIM_INBUF);
    if (!ring || !in) { free(ring); free(in); inflateEnd(&z); return -1; }
    z.avail_out = 0;
    for (;;) {
        if (z.avail_in == 0) {
            ssize_t r = pread(s->fdno, in, SHIM_INBUF, (off_t)fpos);
            if (r <= 0) break;
            fpos += (uint64_t)r;
            z.next_in = in; z.avail_in = (uInt)r;
        }
        if (z.avail_out == 0) { z.next_out = ring; z.avail_out = W; }
        ret = inflate(&z, Z_BLOCK);
        if (ret == Z_STREAM_END) {
            base_in += z.total_in; base_out += z.total_out;
            {   /* a further gzip member?  (what follows is looked at without being consumed) */
                uint8_t m[2] = {0, 0};
                const uint64_t at = fpos - z.avail_in;
                if (pread(s->fdno, m, 2, (off_t)at) == 2 && m[0] == 0x1f && m[1] == 0x8b) {
                    uint8_t *ni = z.next_in; uInt ai = z.avail_in, ao = z.avail_out; uint8_t *no = z.next_out;
                    if (inflateReset(&z) != Z_OK) { ret = Z_DATA_ERROR; break; }
                    z.next_in = ni; z.avail_in = ai; z.next_out = no; z.avail_out = ao;
                    member_start = 1;
                    ret = Z_OK;
                    continue;
                }
            }
            break;
        }
        if (ret == Z_BUF_ERROR) { ret = Z_OK; continue; }
        if (ret != Z_OK) break;
        if ((z.data_type & 128) && !(z.data_type & 64)) {      /* at a block boundary (or right behind a member's header) */
            const uint64_t cout = base_out + z.total_out;
            if (index->npoints == 0 || cout - last >= index->spacing) {
                if (shim_add_point(index, base_in + z.total_in, cout, z.data_type & 7, ring, z.avail_out, !member_start)) { ret = Z_MEM_ERROR; break; }
                last = cout;
            }
            member_start = 0;
        }
    }
    inflateEnd(&z);
    free(ring); free(in);
    if (ret != Z_OK && ret != Z_STREAM_END) { g_errors++; return -1; }
    index->uncompressed_size = base_out;
    return 0;
}

/* last point at or before `off`; -1: none */
static long shim_find(const zran_index_t *index, uint64_t off)
{
    long lo = 0, hi = (long)index->npoints - 1, best = -1;
    while (lo <= hi) {
        const long mid = (lo + hi) / 2;
        if (index->list[mid].uncmp_offset <= off) { best = mid; lo = mid + 1; } else hi = mid - 1;
    }
    return best;
}

int zran_seek(zran_index_t *index, int64_t offset, uint8_t whence, zran_point_t **point)
{
    shim_t *s = (shim_t *)index->shim;
    long k;
    uint64_t off;
    if (point) *point = NULL;
    if (whence == SEEK_CUR) offset += (int64_t)s->want;
    else if (whence != SEEK_SET) return -1;
    if (offset < 0) return -1;
    off = (uint64_t)offset;
    s->want = off;
    g_seeks++;
    k = shim_find(index, off);
    /* going on from where the inflate stands beats every point that lies behind it */
    if (s->live && !s->eof && s->out_pos <= off && (k < 0 || index->list[k].uncmp_offset <= s->out_pos)) { g_continued++; return 0; }
    s->eof = 0;
    if (k < 0) {                                     /* from the start of the file: header, then raw deflate */
        shim_seek_in(s, 0);
        if (shim_header(s) != 0 || shim_raw_init(s)) { g_errors++; return -1; }
        s->out_pos = 0;
        g_from_start++;
        return 0;
    }
    {
        const zran_point_t *p = index->list + k;
        shim_seek_in(s, p->cmp_offset - (p->bits ? 1 : 0));
        if (shim_raw_init(s)) { g_errors++; return -1; }
        if (p->bits) {
            const int b = shim_byte(s);
            if (b < 0 || inflatePrime(&s->z, p->bits, b >> (8 - p->bits)) != Z_OK) { g_errors++; return -1; }
        }
        if (p->data && inflateSetDictionary(&s->z, p->data, index->window_size) != Z_OK) { g_errors++; return -1; }
        s->out_pos = p->uncmp_offset;
        g_from_point++;
        if (k < SHIM_HITCAP) g_hits[k]++;
        if (point) *point = index->list + k;
    }
    return 0;
}

int64_t zran_read(zran_index_t *index, void *buf, uint64_t len)
{
    shim_t *s = (shim_t *)index->shim;
    uint64_t done = 0;
    if (!s->live && zran_seek(index, (int64_t)s->want, SEEK_SET, NULL)) return -1;
    while (done < len && !s->eof) {
        const int dropping = s->out_pos < s->want;
        uint64_t room = dropping ? s->want - s->out_pos : len - done;
        uInt got;
        int ret;
        if (dropping && room > sizeof s->drop) room = sizeof s->drop;
        if (room > (1u << 30)) room = 1u << 30;
        if (!shim_fill(s)) { s->eof = 1; break; }
        s->z.next_out = dropping ? s->drop : (uint8_t *)buf + done;
        s->z.avail_out = (uInt)room;
        ret = inflate(&s->z, Z_NO_FLUSH);
        got = (uInt)room - s->z.avail_out;
        s->out_pos += got;
        if (!dropping) done += got;
        if (ret == Z_STREAM_END) {                   /* trailer (CRC-32, ISIZE: ZRAN_SKIP_CRC_CHECK), then the next member if any */
            int i, h;
            for (i = 0; i < 8; ++i) if (shim_byte(s) < 0) break;
            h = i == 8 ? shim_header(s) : 1;
            if (h < 0) g_errors++;
            if (h != 0 || inflateReset(&s->z) != Z_OK) s->eof = 1;
        } else if (ret != Z_OK && ret != Z_BUF_ERROR) { g_errors++; break; }
    }
    s->want = s->out_pos > s->want ? s->out_pos : s->want;
    (void)shim_consumed;
    return (int64_t)done;
}
