/*
 * zran.h -- TEST INFRASTRUCTURE ONLY (part of oracle/, never shipped or linked by the product).
 *
 * Minimal stand-in for indexed_gzip v1.10.3's zran.h, which the reference
 * (lmdu/pyfastx) downloads at build time (setup.py:53-69) and which is absent
 * from /root/reference.  It declares only the struct fields and entry points
 * that the reference sources touch (src/util.c:442-767, src/index.c:68-70,
 * 383,433,685-686, src/fastq.c:341-342,396, src/read.c:39-40,
 * src/sequence.c:40-41,169,206), so that the reference's own C files can be
 * compiled *where they lie* into oracle/_ref/ and used as the parity oracle.
 *
 * Random access works the way zran's does (see zran.c): restart points at deflate
 * block boundaries (bit offset + 32 KiB window), raw inflate from the last point at
 * or before the wanted offset -- so a gzindex table imported through
 * pyfastx_gzip_index_import (util.c:542-726), e.g. one the product wrote, is what
 * serves the reads.  The PLACEMENT of the points real indexed_gzip would choose is
 * not reproduced (no reference test asserts it; any block boundary is valid).
 */
#ifndef FX_ORACLE_ZRAN_SHIM_H
#define FX_ORACLE_ZRAN_SHIM_H
#include <stdio.h>
#include <stdint.h>
#include <zlib.h>

typedef struct _zran_point {
    uint64_t cmp_offset;
    uint64_t uncmp_offset;
    uint8_t  bits;
    uint8_t *data;
} zran_point_t;

typedef struct _zran_index {
    FILE         *fd;
    void         *f;
    uint64_t      compressed_size;
    uint64_t      uncompressed_size;
    uint32_t      spacing;
    uint32_t      window_size;
    uint32_t      log_window_size;
    uint32_t      readbuf_size;
    uint32_t      npoints;
    uint32_t      size;
    zran_point_t *list;
    uint16_t      flags;
    /* shim state (zran.c: shim_t) */
    void         *shim;
} zran_index_t;

enum { ZRAN_AUTO_BUILD = 1, ZRAN_SKIP_CRC_CHECK = 2 };
enum { ZRAN_EXPORT_OK = 0, ZRAN_EXPORT_WRITE_ERROR = -1 };
enum {
    ZRAN_IMPORT_OK = 0, ZRAN_IMPORT_FAIL = -1, ZRAN_IMPORT_EOF = -2,
    ZRAN_IMPORT_READ_ERROR = -3, ZRAN_IMPORT_INCONSISTENT = -4,
    ZRAN_IMPORT_MEMORY_ERROR = -5, ZRAN_IMPORT_UNKNOWN_FORMAT = -6,
    ZRAN_IMPORT_UNSUPPORTED_VERSION = -7
};

extern const char    ZRAN_INDEX_FILE_ID[5];
extern const uint8_t ZRAN_INDEX_FILE_VERSION;

int     zran_init(zran_index_t *index, FILE *fd, void *f, uint32_t spacing,
                  uint32_t window_size, uint32_t readbuf_size, uint16_t flags);
void    zran_free(zran_index_t *index);
int     zran_build_index(zran_index_t *index, uint64_t from, uint64_t until);
int     zran_seek(zran_index_t *index, int64_t offset, uint8_t whence, zran_point_t **point);
int64_t zran_read(zran_index_t *index, void *buf, uint64_t len);

/* how the seeks of this process were served (tests read these through ctypes from the compiled module):
 * out = {seeks, started at a point, started at the head of the file, continued from the current position,
 *        points created by zran_build_index, inflate / header errors} */
void fxshim_stats(uint64_t out[6]);
void fxshim_reset(void);
void fxshim_point_hits(uint32_t *out, uint32_t n);

#endif
