/*
 * zran.c -- TEST INFRASTRUCTURE ONLY.  gzseek-backed stand-in for
 * indexed_gzip v1.10.3 zran.c (not vendored by the reference; see zran.h).
 */
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/stat.h>
#include "zran.h"

const char    ZRAN_INDEX_FILE_ID[5]   = {'G', 'Z', 'I', 'D', 'X'};
const uint8_t ZRAN_INDEX_FILE_VERSION = 1;

int zran_init(zran_index_t *index, FILE *fd, void *f, uint32_t spacing,
              uint32_t window_size, uint32_t readbuf_size, uint16_t flags)
{
    struct stat st;
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
    {
        int d = dup(fileno(fd));
        lseek(d, 0, SEEK_SET);
        index->gz = gzdopen(d, "rb");
        if (!index->gz) return -1;
        gzbuffer(index->gz, 1 << 20);
    }
    return 0;
}

void zran_free(zran_index_t *index)
{
    uint32_t i;
    if (index->list) {
        for (i = 0; i < index->npoints; ++i) free(index->list[i].data);
        free(index->list);
        index->list = NULL;
    }
    if (index->gz) { gzclose(index->gz); index->gz = NULL; }
}

int zran_build_index(zran_index_t *index, uint64_t from, uint64_t until)
{
    (void)index; (void)from; (void)until;   /* no checkpoints: gzseek serves reads */
    return 0;
}

int zran_seek(zran_index_t *index, int64_t offset, uint8_t whence, zran_point_t **point)
{
    if (point) *point = NULL;
    return gzseek(index->gz, (z_off_t)offset, whence) < 0 ? -1 : 0;
}

int64_t zran_read(zran_index_t *index, void *buf, uint64_t len)
{
    uint64_t done = 0;
    while (done < len) {
        unsigned chunk = (len - done) > (1u << 30) ? (1u << 30) : (unsigned)(len - done);
        int n = gzread(index->gz, (char *)buf + done, chunk);
        if (n <= 0) break;
        done += (uint64_t)n;
    }
    return (int64_t)done;
}
