/*
 * fxgpu.h -- C ABI of libfxgpu.so: MI355X-native FASTA/FASTQ index build and
 * batched random access (the drop-in boundary under the pyfastx object API).
 *
 * pyfastx (lmdu/pyfastx v2.3.1) has no plugin/FFI layer: its hot path is a set
 * of internal C functions inside one CPython extension.  Each entry point
 * below names the reference function(s) whose work it replaces; the
 * Python-facing mirror (pyfastx_amd/) binds these with ctypes, and
 * INTEGRATION.md shows the C stub a pyfastx maintainer would add at those
 * call sites.
 *
 * Conventions
 *   - Plain C: opaque handle, pointers and sizes only.  No torch / HIP types.
 *   - All offsets are 0-based offsets into the UNCOMPRESSED byte stream, i.e.
 *     exactly the numbers the reference stores in its .fxi (seq.boff, read.soff...).
 *   - Every function returns FX_OK (0) or a negative fx_status; the message is
 *     available from fx_last_error() (thread-local).
 *   - `where` arguments say where caller-provided arrays live: FX_HOST or
 *     FX_DEVICE (device pointers must belong to the handle's GPU).
 *   - There is NO CPU fallback: without a usable gfx950 device every call that
 *     needs one fails with FX_EDEVICE.
 *   - A handle is not re-entrant; use one handle per thread / per GPU.
 */
#ifndef FXGPU_H
#define FXGPU_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fx_handle fx_handle;

typedef enum {
    FX_OK = 0,
    FX_ENOENT = -1,   /* input file missing            -> FileExistsError (fasta.c:84-87)  */
    FX_EFORMAT = -2,  /* not FASTA/FASTQ               -> RuntimeError    (fasta.c:107-110) */
    FX_EIO = -3,      /* read / inflate error                                               */
    FX_EDEVICE = -4,  /* no GPU, HIP error                                                  */
    FX_ENOMEM = -5,
    FX_ERANGE = -6,   /* index / interval out of range -> IndexError / ValueError           */
    FX_EINVAL = -7,
    FX_ESTATE = -8    /* call order (e.g. read before build)                                */
} fx_status;

enum { FX_HOST = 0, FX_DEVICE = 1 };

/* fetch flags (per call) */
enum {
    FX_UPPER = 1,       /* remove_space_uppercase, util.c:181-194                 */
    FX_REVERSE = 2,     /* reverse_seq, util.c:251-261                            */
    FX_COMPLEMENT = 4,  /* complement_seq / comp_map, util.c:228-237, 263-269     */
    FX_LONG = 16,       /* hint: ranges are long (KiB+): one wave per query, 1 KiB per step   */
    FX_RAW = 8          /* no despace: plain pyfastx_index_random_read (index.c:683-692), for
                           names, Sequence.raw / .description, Read.raw             */
};

const char *fx_last_error(void);
const char *fx_version(void);
int fx_device_count(void);

/* ------------------------------------------------------------------ staging
 * Replaces gzopen/gzread streaming in pyfastx_init_index (index.c:15-98) and
 * the per-query fseeko/fread | zran_seek/zran_read of pyfastx_index_random_read
 * (index.c:683-692) / pyfastx_read_random_reader (read.c:37-45): the whole
 * uncompressed stream becomes one resident blob in HBM.                        */

/* Plain or gzip file -> pinned double buffers -> hipMemcpyAsync -> HBM.  A BGZF file is inflated on the device (one wave per
 * member; every member checked against the CRC-32 of its trailer, as zlib's gzread does for the reference); a BGZF file of
 * 512 MiB or more in groups of FX_BGZF_GROUP bytes (256 MiB; 0: all at once) BEHIND its staging, the blob sized from the ratio
 * of the first group (what is left over stays with the blob: fx_size is what it holds).  A single gzip stream is inflated on
 * the host cores. */
int fx_open_file(const char *path, int device, fx_handle **out);
/* Where the last fx_open_file of a PLAIN file spent its time (this thread): *alloc_s = the device allocation of the blob --
 * the first block of tens of GB a process asks the driver for can take seconds (35 GB: 2.9 s measured), the next one of that
 * size 0.3 ms --, *stage_s = page cache -> pinned pieces -> HBM.  (The reference has no counterpart: it reads through a 1 MiB
 * buffer, kseq.h:13.) */
int fx_open_laps(double *alloc_s, double *stage_s);
/* A PLAIN file staged in the background (round 6): the handle comes back as soon as its blob is allocated, the staging lanes
 * copy the file into it in file order.  fx_stage_wait(h, upto): returns when the first `upto` bytes are on the device
 * (upto < 0: the whole file; the lanes are joined and the handle is an ordinary one from then on).  Until then the caller
 * works on PREFIXES through views of the blob -- fx_open_device(fx_device_ptr(h) + off, len + halo), fx_set_shard,
 * fx_set_halo: byte ranges as in the sharded build -- e.g. formats and ships the table leaves of an index file
 * (fx_fxi_part_*) while the rest of the input still arrives; no other call on THIS handle before fx_stage_wait(h, -1).
 * fx_close waits for the lanes.  gzip input: FX_EINVAL.  (The reference reads and indexes in one loop, fastq.c:8-182.) */
int fx_open_file_async(const char *path, int device, fx_handle **out);
int fx_stage_wait(fx_handle *h, int64_t upto);
/* ... and the last fx_fastq_build / fx_fastq_build_comp (this thread), eight doubles, seconds: [0] the sample of the stream and its
 * wait, [1] allocations + launches of the count pass, [2] the wait for it, [4] plan + allocation of the read table, [5] row kernels +
 * their wait ([3], [6], [7]: unused).  Diagnostic: a build that takes seconds instead of milliseconds is waiting for the driver. */
int fx_build_laps(double *out8);
/* What a file holds and how long its stream is once inflated -- kind 0: plain; 1: BGZF (*n_bytes = sum of the members'
 * ISIZE, from a walk over their headers); 2: a single gzip stream (*n_bytes = -1: unknown without inflating it). */
int fx_stream_size(const char *path, int64_t *n_bytes, int *kind);
/* ONE byte-range shard of a file (whole-file scan semantics of index.c:230-372 across the cuts come from the
 * stitch below): bytes [off, off + len + halo) of the uncompressed stream, clamped to its end, become the blob, and the
 * shard context is taken from the file -- base = off, prev_byte = the byte before it, is_last, halo (fx_set_shard /
 * fx_set_halo need not be called).  A plain file is read only in that range; of a BGZF file only the members that
 * cover it are read, staged and inflated.  A single gzip stream cannot be entered in the middle: FX_EINVAL. */
int fx_open_file_range(const char *path, int64_t off, int64_t len, int64_t halo, int device, fx_handle **out);
/* Copy nbytes from host memory into HBM. */
int fx_open_host(const void *data, int64_t nbytes, int device, fx_handle **out);
/* Adopt (do not copy, do not free) a blob already in this GPU's HBM; 16-byte aligned. */
int fx_open_device(const void *dptr, int64_t nbytes, int device, fx_handle **out);

/* Shard context (multi-GPU byte-range sharding): this handle holds
 * bytes [base, base+n) of a longer stream; prev_byte is stream[base-1] (pass
 * 10 when base == 0); is_last says the shard ends at end-of-stream.  Default
 * after fx_open_*: base 0, prev_byte 10, is_last 1. */
int fx_set_shard(fx_handle *h, int64_t base, int prev_byte, int is_last);

int fx_close(fx_handle *h);
/* Scratch of an open (compressed bytes of a BGZF file, its match map: hundreds of MB for tens of milliseconds) and, since
 * round 4, the blob of a closed handle are kept in a per-process pool for the next open (FX_SCRATCH_CACHE_MB, default 24576:
 * the driver clears memory it hands out or takes back, 20 ms per 3 GB in the way of the next open's copies; the library
 * empties the pool by itself before a device allocation fails).  This gives the idle blocks back to the driver now.
 * Blobs LARGER than that whole limit (the 35 GB of a sequencing run) are kept too when their handle closes -- freeing one
 * makes the process's next large allocation wait seconds for the driver -- under a cap of their own: FX_SCRATCH_KEEP_BIG_MB
 * per device (default half the device's memory; 0: none) and FX_SCRATCH_BIG_TTL_S (default 300 s idle, checked at the next
 * call into the pool; 0: no limit).  A process that SHARES its GPU with others (several ranks on one device) should call
 * fx_release_scratch after closing large files: nobody else can reclaim what idles here.  (index.c:431-474 frees
 * synchronously; there is no device memory in the reference.) */
int fx_release_scratch(void);
/* The decision behind that cap, as a pure function (tests pin it): the device holds n large idle blocks of idle_caps[]
 * bytes, one more of `cap` bytes comes back, keep_bytes are allowed in all.  -> 1: the new block is kept and the idle
 * blocks with evict[i] = 1 (the smallest, as many as needed) are freed; 0: the new block is freed and nothing else. */
int fx_scratch_policy(const int64_t *idle_caps, int n, int64_t cap, int64_t keep_bytes, int32_t *evict);
/* Pinned (page-locked) host memory out of a per-process pool, for the arrays a caller hands to the batched entry
 * points with FX_HOST: answers land in it by DMA -- no bounce buffer, no first-touch page faults of a fresh buffer
 * (what the copy out of pyfastx_index_fill_cache's cache costs the reference per getter, sequence.c:346-347, is paid
 * here once per batch) -- and query arrays in it go up without a staging copy.  Released blocks stay pinned for the
 * next request of a similar size (FX_PINNED_CACHE_MB, default 2048; fx_pinned_trim gives them back).
 * fx_pinned_holds: does [p, p + bytes) lie inside one block that is handed out? */
void *fx_pinned_alloc(int64_t bytes);
void fx_pinned_free(void *p);
int fx_pinned_holds(const void *p, int64_t bytes);
int fx_pinned_trim(void);
int64_t fx_size(const fx_handle *h);          /* uncompressed bytes held          */
/* Free and total HBM of a device, now.  The reference streams files of any size through a 1 MiB buffer (kseq.h:13,
 * index.c:229-230); here a stream larger than what is free is built and served in byte-range WINDOWS that take turns in HBM
 * (pyfastx_amd/windows.py, fx_open_file_range): this is the number that decides. */
int fx_device_memory(int device, int64_t *free_bytes, int64_t *total_bytes);
int fx_is_gzip(const fx_handle *h);           /* is_gzip_format, util.c:307-325   */
const void *fx_device_ptr(const fx_handle *h);/* device address of the blob       */
/* Raw bytes [off, off+n) of the stream -> dst (host).  Serves names,
 * Sequence.raw / .description (sequence.c:299-335), Read.raw (read.c:124-150). */
int fx_read_bytes(fx_handle *h, int64_t off, int64_t n, void *dst);
/* First non-space byte of the stream (fasta_validator / fastq_validator, util.c:95-150). */
int fx_first_byte(fx_handle *h, int *out);

/* ------------------------------------------------------------ FASTA index
 * Replaces the hot loop of pyfastx_create_index (index.c:230-372).           */
typedef struct {
    int64_t n_seq;        /* stat.seqnum                                      */
    int64_t seq_len;      /* stat.seqlen  (sum of slen)                       */
    int64_t n_lines;      /* newline-terminated lines seen (incl. EOF line)   */
    int64_t n_bytes;
} fx_fasta_summary;

/* Build the record table on the GPU (kept resident in HBM for fetches).
 * full_name, bit 0: chrom = whole header (index.c:282-285) instead of first token.
 * bit 1 (FX_BUILD_COMP): count the letters on the way -- the reference's second pass over the file
 * (pyfastx_fasta_calc_composition, fasta.c:851-961) rides on the index scan, ONE read of the stream for both; a later
 * fx_fasta_comp / _shard / _sparse then only attributes the counts to the records and reads the bytes of record
 * boundaries again.  The counts are dropped by the next build or fx_set_shard. */
#define FX_BUILD_FULL_NAME 1
#define FX_BUILD_COMP 2
int fx_fasta_build(fx_handle *h, int full_name, fx_fasta_summary *out);

/* The same in two halves: _begin ENQUEUES the whole build on the handle's stream and returns; _end waits for it and
 * reports the totals.  Device-side consumers -- fx_fasta_fetch with FX_DEVICE arrays, fx_shard_summary_dev,
 * fx_fasta_stitch_dev -- may be enqueued between the two (they read the record count from device memory), so a
 * "build, then answer a batch" step costs one host synchronisation.  Calls that need host-side totals
 * (fx_fasta_table, fx_fasta_comp, ...) complete a pending build themselves. */
int fx_fasta_build_begin(fx_handle *h, int full_name);
int fx_fasta_build_end(fx_handle *h, fx_fasta_summary *out);

/* Copy the SoA record table to caller arrays (any pointer may be NULL).
 * Columns are exactly the .fxi `seq` columns (index.c:178-189) plus the
 * header-line offset and the name span inside the stream. */
int fx_fasta_table(fx_handle *h, int where,
                   int64_t *hoff, int64_t *boff, int64_t *blen, int64_t *slen,
                   int64_t *llen, int32_t *elen, int32_t *norm, int32_t *dlen,
                   int32_t *name_len);

/* Install the record table of an existing .fxi instead of scanning (pyfastx_load_index, index.c:391-429):
 * host arrays of the `seq` columns; afterwards fx_fasta_fetch works exactly as after fx_fasta_build.
 * (Composition and the shard summary still need a scan.) */
int fx_fasta_set_table(fx_handle *h, int64_t n, const int64_t *boff, const int64_t *blen, const int64_t *slen,
                       const int64_t *llen, const int32_t *elen, const int32_t *norm);

/* Which records may be sliced with the line arithmetic of pyfastx_sequence_subscript (sequence.c:498-510)?  `norm`
 * (index.c:342: at most ONE line of another length) is 1 both for a record whose last line is the short one and for a
 * record with one odd line in the middle -- where the arithmetic addresses the wrong bytes (the reference returns them
 * from a cold cache and the true slice from a warm one, sequence.c:100-110).  reg[i] = 1: every line of record i but
 * the last holds exactly llen - elen bases (decided by the scan / fx_fasta_set_table / the stitch from the row and ONE
 * byte of the stream: the byte in front of the last line is a newline).  fx_fasta_fetch slices the other records after
 * despacing them; so do all batched paths above it.  n_seq values, where = FX_HOST / FX_DEVICE. */
int fx_fasta_line_regular(fx_handle *h, int where, int32_t *reg);

/* pyfastx_fasta_calc_composition (fasta.c:851-961): comp[n_seq][128] counts of
 * every byte value < 128 on the sequence lines of each record ('\r' included,
 * '\n' excluded).  Needs fx_fasta_build first. */
int fx_fasta_comp(fx_handle *h, int where, int64_t *comp);
/* The same for one byte-range shard (fx_set_shard).  The bytes of a shard that precede its first header line belong
 * to a record whose header lies in an earlier shard: they are counted into lead[128] (host), from global offset
 * lead_from on -- the later of the shard's start and that record's boff, so that a header line crossing the cut is
 * left out (lead_from < 0: nothing is counted, e.g. shard 0).  The owner of the record adds the lead rows of the
 * following shards up to and including the first one that holds a header line (pyfastx_amd/shard.py).              */
int fx_fasta_comp_shard(fx_handle *h, int where, int64_t *comp, int64_t lead_from, int64_t *lead);
/* The same composition in the form the `comp` table stores it (fasta.c:904-914): only the non-zero bins, as triples
 * (seqid = record index + 1, abc = byte value, num = count) in record order, letters ascending -- the dense matrix
 * never leaves HBM (5 M records: 5 GB dense, ~50 M triples).  cap: room in seqid / abc / num (where = FX_HOST /
 * FX_DEVICE); *n_out: number of triples -- with FX_ERANGE when it exceeds cap (nothing is written then);
 * total[128] (host): column sums, the seqid = 0 rows of the table (fasta.c:943-950). */
int fx_fasta_comp_sparse(fx_handle *h, int where, int64_t cap, int64_t *seqid, int64_t *abc, int64_t *num,
                         int64_t *n_out, int64_t *total);

/* ------------------------------------------------------------ FASTQ index
 * Replaces pyfastx_fastq_create_index (fastq.c:89-171).                      */
typedef struct {
    int64_t n_reads;      /* stat.counts = line_num / 4                       */
    int64_t size;         /* stat.size   = sum of rlen                        */
    int64_t n_lines;
    int64_t n_bytes;
    int64_t first_id;     /* global 0-based id of this shard's first read (0 for a whole file) */
} fx_fastq_summary;

int fx_fastq_build(fx_handle *h, fx_fastq_summary *out);
/* The same with the composition counted ON THE WAY -- pyfastx_fastq_create_index (fastq.c:8-182) and
 * pyfastx_fastq_calc_composition (fastq.c:663-795), the reference's two passes over the file, in ONE read of the stream: the
 * count pass of the build classifies the bytes it holds in registers anyway; which line of four a byte belongs to is guessed
 * per run of granules from the '+' lines and checked against the newline prefixes afterwards.  A later fx_fastq_comp then
 * only hands the counts out; when any guess was wrong (or the file has what the stream form does not do: CRLF, bytes outside
 * '!'..127 in a quality line) it counts from the read table as if this had been fx_fastq_build.  Whole streams only. */
int fx_fastq_build_comp(fx_handle *h, fx_fastq_summary *out);
/* How the last fx_fastq_build_comp counted: *runs = runs of 64 KiB the stream was cut into, each counted on a GUESS of its line
 * phase (a '+' line of one byte, or '+' and '\r'); *recounted = runs whose guess the newline prefixes proved wrong or that had
 * none -- counted again from the prefixes (-1: too many of them, nothing of the one-read result was used); *one_read = 1 when
 * fx_fastq_comp answers from the build's counters, 0 when it reads the stream again through the read table (a quality byte
 * outside '!'..127, a '\r' that is not the end of its line: fastq.c:731-745's quirks are the table kernels').  The reference
 * has no counterpart (its loop fastq.c:715-753 is one pass of its own over the file). */
int fx_fastq_comp_info(fx_handle *h, int64_t *runs, int64_t *recounted, int *one_read);


/* Sharded FASTQ: records are short, so a shard carries a HALO -- the first
 * bytes of the next shard appended to its own range (fx_set_halo) -- and owns every record
 * whose header line STARTS in its core.  The only context it needs is the global line
 * numbering: phase 1 (fx_fastq_scan) reports the newlines of the core, ONE all-gather of
 * those two integers per rank gives every rank line_offset (newlines in earlier cores) and
 * prev_nl (offset of the last of them, -1 if none), phase 2 (fx_fastq_build_ctx) builds the
 * table.  Composition and fetch then work on the owned records.                           */
int fx_set_halo(fx_handle *h, int64_t halo_bytes);
int fx_fastq_scan(fx_handle *h, int64_t *n_nl_core, int64_t *last_nl_core);
int fx_fastq_build_ctx(fx_handle *h, int64_t line_offset, int64_t prev_nl, fx_fastq_summary *out);
int fx_fastq_table(fx_handle *h, int where,
                   int64_t *name_off, int32_t *name_len, int32_t *dlen,
                   int64_t *rlen, int64_t *soff, int64_t *qoff);

/* pyfastx_fastq_calc_composition (fastq.c:715-774): base = {A,C,G,T,N},
 * meta = {maxlen, minlen, minqs, maxqs, phred} in the .fxi column order. */
int fx_fastq_comp(fx_handle *h, int64_t base[5], int64_t meta[5]);

/* ------------------------------------------------------------------ fetch
 * Replaces pyfastx_index_fill_cache (index.c:694-707) + remove_space[_uppercase]
 * (util.c:166-194) + reverse/complement (util.c:239-269) for a BATCH of byte
 * ranges: query i reads blen[i] bytes at off[i], drops bytes 10/13/32, keeps
 * the first min(kept, slen[i]) bytes, applies flags, and writes them at
 * dst + dst_off[i].  out_len[i] (optional) receives the bytes written.
 * flags: per-call if flags_per_query == NULL, else flags_per_query[i].        */
int fx_fetch_ranges(fx_handle *h, int where, int64_t n,
                    const int64_t *off, const int64_t *blen, const int64_t *slen,
                    int flags, const uint8_t *flags_per_query,
                    uint8_t *dst, const int64_t *dst_off, int64_t *out_len);

/* The same with a slice taken AFTER despacing: of the bytes query i keeps, the first skip[i] are dropped and at most
 * take[i] stored -- pyfastx_sequence_get_subseq on a record that is not line-regular (sequence.c:100-110: the whole
 * record is despaced, then `seq + start - 1` is copied), without moving the record to the host. */
int fx_fetch_slices(fx_handle *h, int where, int64_t n,
                    const int64_t *off, const int64_t *blen, const int64_t *skip, const int64_t *take,
                    int flags, const uint8_t *flags_per_query,
                    uint8_t *dst, const int64_t *dst_off, int64_t *out_len);

/* The same for ONE range and a host buffer -- what a single getter of the reference does (pyfastx_index_fill_cache +
 * the copy of slen bytes, index.c:694-707, sequence.c:346-347; pyfastx_read_random_reader, read.c:37-45): descriptor
 * and result travel through pinned host memory, one launch and one wait per call.  skip / take as in FetchQ: the
 * first `skip` kept bytes are dropped, at most `take` are stored at dst; *out_len = bytes stored. */
int fx_fetch_one(fx_handle *h, int64_t off, int64_t blen, int64_t skip, int64_t take, int flags, uint8_t *dst, int64_t *out_len);

/* Same, but queries are (record id 0-based, start, stop) half-open 0-based
 * base coordinates resolved against the resident FASTA table with the
 * arithmetic of pyfastx_sequence_subscript (sequence.c:498-510) for norm=1
 * records and despace-then-slice (sequence.c:100-110) for norm=0 records. */
int fx_fasta_fetch(fx_handle *h, int where, int64_t n,
                   const int64_t *seq_id, const int64_t *start, const int64_t *stop,
                   int flags, const uint8_t *flags_per_query,
                   uint8_t *dst, const int64_t *dst_off, int64_t *out_len);

/* The same batch from host arrays with the layout left to the library -- the loop of the reference's benchmark
 * (benchmark/pyfastx_fasta_extract_subsequences.py:8-12: fa[name][s:e].seq per interval; pyfastx_sequence_subscript
 * sequence.c:412-517 + pyfastx_sequence_get_subseq sequence.c:76-125 per call) as ONE call: the intervals are checked on
 * the device against the resident table (first invalid one -> *first_bad, FX_ERANGE), the answers are laid out back to
 * back (offsets by a device scan) and come home by DMA into pinned memory of fx_pinned_alloc: *dst (the bytes) and
 * *dst_off (n + 1 offsets) belong to the caller, who gives them back with fx_pinned_free. */
int fx_fasta_fetch_alloc(fx_handle *h, int64_t n, const int64_t *seq_id, const int64_t *start, const int64_t *stop,
                         int flags, const uint8_t *flags_per_query, uint8_t **dst, int64_t **dst_off, int64_t *first_bad);
/* Where the last fx_*_fetch_alloc call of this thread spent its time on the host, in milliseconds (measurement aid): ms[0]
 * query arrays staged + uploads enqueued, [1] counts, scan, offsets back (first wait), [2] pinned blocks, [3] kernels
 * enqueued, [4] answers back (second wait), [5] the whole call. */
int fx_fetch_phases(double *ms, int cap);

/* FASTQ reads by 0-based id (read.c:37-45, 152-167, 237-278): seq and qual
 * are rlen bytes each at dst_off[i]; quali = qual - phred as int8
 * (phred 0 -> 33, read.c:268).  Any of seq/qual/quali may be NULL. */
int fx_fastq_fetch(fx_handle *h, int where, int64_t n, const int64_t *read_id,
                   int phred, int seq_flags,
                   uint8_t *seq, uint8_t *qual, int8_t *quali, const int64_t *dst_off);
/* ... with the layout left to the library (fq[i].seq / .qual / .quali for many i, read.c:152-167, 237-278): read lengths
 * are gathered and scanned on the device, the wanted outputs (want: bit 0 seq, 1 qual, 2 quali) and *dst_off (n + 1)
 * come back in pinned memory of fx_pinned_alloc (fx_pinned_free each); an id outside the table -> *first_bad, FX_ERANGE. */
int fx_fastq_fetch_alloc(fx_handle *h, int64_t n, const int64_t *read_id, int phred, int seq_flags, int want,
                         uint8_t **seq, uint8_t **qual, int8_t **quali, int64_t **dst_off, int64_t *first_bad);

/* ------------------------------------------------------------------ Fastx
 * Replaces kseq_read (kseq.c:138-179) as pyfastx_fastx_next drives it (fastx.c:124-130): index-free iteration over a
 * file with kseq's own record rules -- FASTA and FASTQ records mixed, sequence / quality over any number of lines,
 * text between records skipped, white space kept inside the lines, a trailing CR dropped per ks_getuntil2 call
 * (kseq.c:106).  One record per kseq_read that returned >= 0. */
typedef struct {
    int64_t hdr_off;    /* first byte behind the '>' / '@'; name and comment are cut from these bytes (kseq.c:148-149)  */
    int64_t hdr_line;   /* number of the line that holds it                                                            */
    int64_t seq_len;    /* seq.l; qual.l of a FASTQ record is the same (kseq.c:176)                                    */
    int64_t seq_cum;    /* sum of seq_len over the records before this one                                             */
    uint32_t hdr_len;   /* bytes to the end of the line, '\n' excluded                                                  */
    uint32_t s_n;       /* lines between the header line and the line that ended the sequence                          */
    uint32_t q_n;       /* quality lines read                                                                          */
    uint32_t flags;     /* 1: FASTQ record (kseq_read went through kseq.c:167-177); 2: its quality read met the end of
                           the stream at once (the reference's buffer keeps its old content); 4: the header line has
                           no '\n' behind it                                                                           */
} fx_kseq_rec;
/* The walk over the whole resident stream.  *end_code = the negative value that ended the reference's iteration:
 * -1 end of file, -2 truncated quality string (kseq.c:131-136). */
int fx_kseq_scan(fx_handle *h, int64_t *n_records, int64_t *n_lines, int64_t *seq_bytes, int *end_code);
/* How many lines of the last scan the parallel passes took (a file of four-line FASTQ records, or of plain FASTA header /
 * sequence lines, up to the first line that is neither); the sequential walk did the rest.  FX_KSEQ_WALK_ONLY=1 in the
 * environment sends everything through the walk. */
int64_t fx_kseq_prefix_lines(const fx_handle *h);
/* Records [first, first + count) of the table, to host memory. */
int fx_kseq_records(fx_handle *h, int64_t first, int64_t count, fx_kseq_rec *out);
/* Their sequence strings, one behind the other (record k at seq_cum[k] - seq_cum[first]), and their quality strings at
 * the same offsets (zero bytes where a record has none); either destination may be NULL.  flags: FX_UPPER (sequence
 * only, fastx.c:14-22).  *n_bytes = bytes per destination. */
int fx_kseq_fetch(fx_handle *h, int where, int64_t first, int64_t count, int flags, uint8_t *seq_dst, uint8_t *qual_dst,
                  int64_t *n_bytes);


/* The same with explicit per-read offsets (the reference's call shape:
 * pyfastx_read_random_reader(read, buff, offset, bytes), read.c:37-45), for
 * callers that hold .fxi rows (soff, qoff, rlen) rather than ids. */
int fx_read_fetch(fx_handle *h, int where, int64_t n, const int64_t *soff, const int64_t *qoff,
                  const int64_t *rlen, int phred, int seq_flags,
                  uint8_t *seq, uint8_t *qual, int8_t *quali, const int64_t *dst_off);

/* ------------------------------------------------------------------ names
 * Batched name -> record id, replacing one `SELECT ... WHERE chrom=? LIMIT 1` B-tree probe per name
 * (pyfastx_index_get_seq_by_name, index.c:527-566; pyfastx_fastq_get_read_by_name, fastq.c:486-519) with an
 * open-addressing table of record ids in HBM whose keys are the name bytes of the resident stream.
 * kind: 0 = FASTA sequence names (first token, or the whole header after full_name), 1 = FASTQ read names.
 * Query names are packed: bytes of query i = qbytes[qoff[i] .. qoff[i+1]); with FX_HOST the packed buffer must
 * be readable 8 bytes past its end.  out_ids[i] = 0-based id of the first record with that name, -1 if none. */
int fx_names_build(fx_handle *h, int kind);
int fx_names_lookup(fx_handle *h, int where, int64_t nq, const uint8_t *qbytes, const int64_t *qoff, int64_t *out_ids);

/* Sorted order of the record names for the UNIQUE INDEX of the .fxi -- what `CREATE UNIQUE INDEX chromidx ON seq
 * (chrom)` (index.c:363) / `readidx ON read (name)` (fastq.c:152) sort on the CPU after the inserts.  order[i]
 * (n records, `where` = FX_HOST / FX_DEVICE) = 0-based id of the i-th smallest name in SQLite's BINARY collation
 * (memcmp, the shorter name first on a common prefix); equal names keep id order.  *n_dup (host) = number of
 * adjacent equal pairs: 0 means the names are distinct and fx_fxi_bulk_index may write the index from `order`;
 * otherwise the reference's CREATE UNIQUE INDEX fails (and is ignored), so no index is written.  kind as above. */
int fx_names_sort(fx_handle *h, int kind, int where, int64_t *order, int64_t *n_dup);

/* The record names back to back (what the `chrom` / `name` column of the .fxi stores), gathered from the record table
 * that is already in HBM: dst[name_off[i] .. name_off[i+1]) = name of record i, name_off: n + 1 host words, *total =
 * name_off[n].  FX_ERANGE when total > cap (only *total is valid then; n * longest name is a safe cap).  Host buffers. */
int fx_names_pack(fx_handle *h, int kind, uint8_t *dst, int64_t cap, int64_t *name_off, int64_t *total);
/* The same order for names that come from SEVERAL handles (the shards of one file): packed host buffer, n + 1 offsets;
 * order[i] = 0-based index of the i-th smallest name, *n_dup = adjacent equal pairs.  For the index b-tree of a merged
 * .fxi (fx_fxi_bulk_index) without CREATE UNIQUE INDEX's sort (index.c:363, fastq.c:155). */
int fx_sort_packed_names(int device, const uint8_t *names, const int64_t *name_off, int64_t n, int64_t *order, int64_t *n_dup);

/* ------------------------------------------------------------ statistics
 * Fasta.count(n), nl(p), longest, shortest, mean, median (fasta.c:573-849) ask SQLite to scan or sort the seq table, one
 * query each.  Here the lengths are already in HBM: one stable radix sort of (slen, id), a scan of the sorted lengths and
 * one probe kernel answer all of them in a call.
 *   longest_id / shortest_id: 0-based id of the FIRST record with the extreme length (SQLite's MAX()/MIN() keep the first
 *   row they met: `SELECT ID,MAX(slen) FROM seq`, fasta.c:682, 715);
 *   count_ge: records with slen >= count_min (`SELECT COUNT(*) FROM seq WHERE slen>=?`, fasta.c:586);
 *   med_lo, med_hi: the sorted lengths at (n-1)/2 and, for an even n, the one after it (fasta.c:812-816);
 *   nx_len, nx_count: N(p) / L(p) -- walking the lengths in descending order, the first length (and how many so far)
 *   at which the running sum reaches `half`, a double = p / 100.0 * stat.seqlen compared as in fasta.c:630-647; 0, 0
 *   when the sum never does.  Needs the record table (fx_fasta_build or fx_fasta_set_table). */
typedef struct {
    int64_t n_seq, sum_len;
    int64_t longest_id, longest_len, shortest_id, shortest_len;
    int64_t count_ge;
    int64_t med_lo, med_hi;
    int64_t nx_len, nx_count;
} fx_len_stats;
int fx_fasta_len_stats(fx_handle *h, int64_t count_min, double half, fx_len_stats *out);

/* pyfastx.reverse_complement / reverse_seq / complement_seq on a caller buffer
 * (module.c:44-59; util.c:239-269).  mode: FX_REVERSE | FX_COMPLEMENT.        */
int fx_revcomp(int device, int where, uint8_t *buf, int64_t n, int mode);

/* ------------------------------------------------------------ gzip inputs
 * fx_open_file inflates BGZF (bgzip) inputs on the GPU, one work-item per
 * member; other gzip files are inflated by zlib on the host (a single deflate
 * stream is serial).  fx_gz_points lists restart points for the .fxi `gzindex`
 * table (pyfastx_build_gzip_index / pyfastx_gzip_index_export, util.c:442-540,
 * 728-742): for BGZF, member boundaries about `spacing` uncompressed bytes apart
 * (cmp_off = the first byte of the member's deflate data, where zran_seek starts a raw
 * inflate; bit offset 0, no 32 KiB window needed).  Call with cap = 0 to get the count.
 * Non-BGZF inputs report 0 points here: see fx_gz_checkpoints.                 */
int fx_gz_points(fx_handle *h, int64_t spacing, int64_t *cmp_off, int64_t *uncmp_off, int64_t cap,
                 int64_t *n_out, int64_t *compressed_size);

/* The first open of a single gzip stream runs on all cores of the host (fx_pgzip.hpp: block starts searched behind T cuts
 * of the compressed bytes, the pieces decoded with markers for the 32 KiB in front of them, resolved in order; CRC-32 and
 * ISIZE checked; restart points captured) -- zran_build_index's serial pass (util.c:728-742, index.c:383) in parallel.
 * fx_open_file uses it by itself (FX_GZIP_SERIAL=1: zlib on one core, as before); this entry is the same code without
 * a device: 0 done, 1 not a case for it (small file, several members, nothing it is sure of: inflate serially),
 * FX_ERANGE when `out` is too small (*out_n then says how much is needed). */
int fx_gunzip_parallel(const uint8_t *in, int64_t n, int threads, uint8_t *out, int64_t cap, int64_t *out_n, int64_t *n_points);
/* how the open that made this handle inflated a gzip input: 0 plain, 1 BGZF on the device, 2 one stream serially (zlib),
 * 3 one stream on all host cores, 4 from the restart points of its index file (diagnostics) */
int fx_gz_open_mode(const fx_handle *h);

/* How the members of a BGZF file were inflated by the open that made this handle: out = {members, members the
 * wave-per-member decoder handed over to the serial one, INFL_RETRY + reason of the first of those} (fx_inflate_par.hpp:
 * the lanes of a wave start at 64 bit positions of a member and fall into step by Huffman self-synchronisation; anything
 * out of the ordinary, damaged members included, is decoded by one lane).  Diagnostics; replaces nothing in the reference. */
int fx_bgzf_counts(fx_handle *h, int64_t out[3]);

/* Single-stream gzip (not BGZF): the restart points zran would build (pyfastx_build_gzip_index, util.c:728-742; spacing
 * 1 MiB, window 32 KiB, index.c:70) are captured while fx_open_file inflates the stream on the host -- at deflate block
 * boundaries at least 1 MiB of output apart: offset of the next compressed byte, number of bits of the byte before it
 * that still belong to the point (zran's `bits`), offset in the inflated stream, and the 32 KiB of output before it
 * (point 0 is the start of the deflate data and has none).  fx_gz_checkpoints hands them out for the gzindex table
 * (windows: 32768 bytes per point with has_data, in order; call with cap = 0 for the counts);  fx_open_file_indexed is
 * fx_open_file with the points of an existing index: the segments between them are inflated by many host threads at
 * once, each a raw inflate primed with its bits and window (zran_seek + zran_read of index.c:685-686, for every
 * segment at the same time).  Points that do not describe the file: the serial inflate, silently.  indexed_gzip is
 * not part of the reference tree; the compiled reference of the tests imports these rows (util.c:542-726) and serves
 * its reads from them through a zran work-alike (oracle/refshim): offsets, bits and windows are pinned
 * that way, only the PLACEMENT real zran would choose for its points is not. */
int fx_gz_checkpoints(fx_handle *h, int64_t cap, int64_t *cmp_off, int64_t *uncmp_off, uint8_t *bits, uint8_t *has_data,
                      uint8_t *windows, int64_t *n_out, int64_t *n_windows);
int fx_open_file_indexed(const char *path, int device, int64_t n_points, const int64_t *cmp_off, const int64_t *uncmp_off,
                         const uint8_t *bits, const uint8_t *has_data, const uint8_t *windows, int64_t uncompressed_size,
                         fx_handle **out);

/* ------------------------------------------------------------ .fxi bulk load
 * Host-side.  Replaces the per-record `sqlite3_step(INSERT)` of index.c:239-251 / fastq.c:136-149 for the one big
 * table of an index file: rows arrive in rowid order, so the table b-tree (leaf pages left to right, a few interior
 * pages on top) is written straight into the database file that SQLite created.  `path`: a database with the
 * schema in place and NO open connection; `rootpage`: sqlite_master.rootpage of the (empty) table; rows i = 0..n-1
 * get rowid i+1 and are (NULL [the INTEGER PRIMARY KEY], names[name_off[i] .. name_off[i+1]) as TEXT, cols[0][i], ...
 * cols[ncols-1][i] as INTEGER).  FX_ERANGE: some row needs an overflow page -- use INSERTs instead.  The UNIQUE
 * INDEX is created by SQLite afterwards. */
int fx_fxi_bulk_rows(const char *path, int rootpage, int64_t n, const uint8_t *names, const int64_t *name_off,
                     int ncols, const int64_t *const *cols);

/* The UNIQUE INDEX on the name column (index.c:363 `CREATE UNIQUE INDEX chromidx`, fastq.c:152 `readidx`) loaded the
 * same way: `rootpage` of the (empty) index; names / name_off as given to fx_fxi_bulk_rows; order[i] = 0-based row of
 * the i-th smallest name (memcmp order, the shorter first on a common prefix = SQLite's BINARY collation;
 * fx_names_sort produces it on the GPU and reports whether the names are distinct -- only then may this be used).
 * FX_ERANGE: an entry would need an overflow page -- let SQLite build the index. */
int fx_fxi_bulk_index(const char *path, int rootpage, int64_t n, const uint8_t *names, const int64_t *name_off,
                      const int64_t *order);
/* A non-unique INDEX on an INTEGER column (fasta.c:952 `CREATE INDEX seqidx ON comp (seqid)`): key[row], order[i] =
 * row of the i-th entry in (key, rowid) order.  fx_fxi_bulk_rows with names = name_off = NULL loads a table without a
 * TEXT column (`comp`: rowid, then cols[] as INTEGERs). */
int fx_fxi_bulk_index_int(const char *path, int rootpage, int64_t n, const int64_t *key, const int64_t *order);

/* The same two b-trees formatted ON THE DEVICE (round 5, csrc/fx_fxi_dev.hpp): the record table of the handle's last
 * build, the names where they are in the resident stream and the sorted order never leave HBM -- only finished 4 KiB
 * pages cross PCIe and go into the file (pinned pieces, several copy threads, fallocate running ahead of them).  Replaces
 * the INSERT loop and CREATE UNIQUE INDEX of fastq.c:29-60, 136-171 (`read`, `readidx`) and index.c:178-207, 239-251, 363
 * (`seq`, `chromidx`) for a NEW index file.
 *   fx_fxi_dev_sort:  kind 0 = FASTA table, 1 = FASTQ table; the BINARY-collation order of the record names is computed
 *                     and kept in the handle; *n_dup = adjacent equal pairs (> 0: the names are not distinct, SQLite's
 *                     CREATE UNIQUE INDEX would fail and the reference ignores that: write the table only).
 *   fx_fxi_dev_write: `path` = a database with the schema in place and NO open connection, 4 KiB pages;
 *                     root_table / root_index = sqlite_master.rootpage of the empty table / of the empty UNIQUE INDEX on its
 *                     name column (0: no index; else fx_fxi_dev_sort must have run).  laps (8 doubles, may be NULL), seconds:
 *                     [0] table shape, [1] table leaf kernels, [2] table leaves to the file, [3] file grown and allocated
 *                     (fallocate), [4] index shape + the dividers for the upper levels, [5] index leaf kernels, [6] index
 *                     leaves to the file, [7] rest of the host's levels + header; filled on every way out.  FX_ERANGE: a row / entry needs an overflow page (or
 *                     >= 2^32 records) -- nothing usable was written, use the host loaders or INSERTs; FX_EINVAL: not a
 *                     database this loader can extend (other page size, reserved bytes, auto-vacuum). */
int fx_fxi_dev_sort(fx_handle *h, int kind, int64_t *n_dup);
int fx_fxi_dev_write(fx_handle *h, int kind, const char *path, int root_table, int root_index, double *laps);
/* Both in one call (round 6), the sort and the shape of the index computed BESIDE the copy-out of the table's leaves (a second
 * stream, a second thread: the device is idle while 6 GB of pages cross PCIe).  The schema must hold the empty UNIQUE INDEX
 * already (root_index >= 2) -- whether the names are distinct is only known when the sort is done: *n_dup > 0 means no index
 * was written and the caller drops the empty one (fastq.c:152-156 / index.c:363-366 ignore the failed CREATE UNIQUE INDEX).
 * laps as fx_fxi_dev_write, [4] = what of sort + index shape the table's copy-out did NOT hide. */
int fx_fxi_dev_build(fx_handle *h, int kind, const char *path, int root_table, int root_index, int64_t *n_dup, double *laps);
/* Room for those pages set aside while the stream is still being staged: `path` = the database SQLite has just created
 * (schema in place, no open connection); a thread of the library grows the file to `bytes` with fallocate (on tmpfs the
 * allocation of 10 GB takes 0.6 s and must not run beside the stores into the file; beside the staging it costs nothing);
 * device >= 0: the thread runs on the CPUs next to that device, so that the pages lie in the memory the copy threads of
 * fx_fxi_dev_write are next to (-1: wherever).
 * The database header keeps saying where the database ends; fx_fxi_dev_write uses the room and cuts the file to what it
 * needed.  fx_fxi_presize_end waits for the thread (cancel != 0: stops it at the next 256 MiB step first).  An estimate
 * that is too small only means the rest is allocated by fx_fxi_dev_write.  No counterpart in the reference: its index
 * file grows one INSERT at a time (fastq.c:136-149). */
int fx_fxi_presize_begin(const char *path, int64_t bytes, int device, void **token);
int fx_fxi_presize_end(void *token, int cancel);

/* ONE index file from SEVERAL handles (round 6): a file indexed by byte range -- one process per GPU, the
 * devices of one process, or windows of one device that take turns -- has the rows of `read` / `seq` in several handles;
 * part r holds rows [row_base_r, row_base_r + n_r), in order.  Every part formats ITS table leaves on its own device and
 * copies them into its own page range of the one file; the index needs all names in one order, so every part hands its
 * names to device memory the caller moves to the writing rank (the process group's gather: RCCL over xGMI), which sorts
 * them once and formats the index leaves from that buffer.  Replaces, for a sharded build, the same reference loops as
 * fx_fxi_dev_write (fastq.c:29-60, 136-171; index.c:178-207, 239-251, 363).  Order of the calls:
 *   every part   fx_fxi_part_shape(h, kind, row_base, out3): out3 = rows, table leaves, bytes of names.  The caller adds up
 *                the leaves over the parts (they travel with the build's all-gather).
 *   writer       creates the database (schema + the empty UNIQUE INDEX), no open connection; fx_fxi_join_grow(path,
 *                root_table, all parts' leaves, extra_bytes, device, &first_new_page) makes room for the table's pages and
 *                extra_bytes behind them (an estimate of the index; best effort) and says where the new pages begin --
 *                every part needs that number.
 *   every part   fx_fxi_part_leaves(h, kind, path, first_new_page, leaves of the parts before this one, laps2): kernels +
 *                copy-out into the file (laps2: seconds in the kernels, in the copies).  The parts may run at the same
 *                time, from different processes.  A table with ONE leaf in all: leaf_base = -root_table (it lives in
 *                the root page).
 *   every part   fx_fxi_part_names(h, kind, d_names, d_lens): DEVICE pointers on the handle's device, room for out3[2] + 64
 *                bytes and out3[0] lengths; fx_fxi_part_firsts(h, first_rows): 0-based first row of each of its leaves
 *                (host, out3[1] values) for the table's interior pages.
 *   writer       fx_fxi_join_begin(device, all names back to back in part order, all lengths, n, &j, &n_dup): one sort;
 *                n_dup > 0: the names are not distinct, write no index (fastq.c:152-156 ignores the failure of CREATE
 *                UNIQUE INDEX) -- pass root_index = 0 below and drop the empty index from the schema.
 *                fx_fxi_join_write(j, path, root_table, root_index, rows, table leaves, first_rows of all parts in order,
 *                first_new_page, laps6) once every part's fx_fxi_part_leaves has returned: the table's interior levels,
 *                the index (leaves on the device, upper levels on the host), the header; laps6: file grown, index shape +
 *                dividers, index leaf kernels, index leaves to the file, host levels + header, table interior.
 *                fx_fxi_join_end(j).
 * FX_ERANGE from any of them: a row / entry needs an overflow page -- remove the file and use the host loaders. */
typedef struct fx_fxi_join fx_fxi_join;
int fx_fxi_part_shape(fx_handle *h, int kind, int64_t row_base, int64_t *out3);
int fx_fxi_part_firsts(fx_handle *h, int64_t *first_rows);
int fx_fxi_part_names(fx_handle *h, int kind, uint8_t *d_names, int32_t *d_lens);
int fx_fxi_part_leaves(fx_handle *h, int kind, const char *path, int64_t first_new_page, int64_t leaf_base, double *laps);
int fx_fxi_join_grow(const char *path, int root_table, int64_t nleaf_table, int64_t extra_bytes, int device, int64_t *first_new_page);
int fx_fxi_join_begin(int device, const uint8_t *d_names, const int32_t *d_lens, int64_t n, fx_fxi_join **out, int64_t *n_dup);
int fx_fxi_join_write(fx_fxi_join *j, const char *path, int root_table, int root_index, int64_t n_rows, int64_t nleaf_table,
                      const int64_t *first_rows, int64_t first_new_page, double *laps);
void fx_fxi_join_end(fx_fxi_join *j);

/* ------------------------------------------------------- sync and timing
 * Calls that take FX_DEVICE arrays return after ENQUEUEING work on the
 * handle's stream; fx_sync waits for it.  (FX_HOST calls are synchronous.)   */
int fx_sync(fx_handle *h);

/* Optional per-kernel timing with HIP events on the handle's own stream
 * (what bench.py's roofline leg reads).  Kernel ids 0..fx_prof_count()-1,
 * names from fx_prof_name (they match the rocprofv3 kernel names).           */
int fx_prof_enable(fx_handle *h, int on);      /* 0 off, 1 every kernel, 2 only k_scan (2 events per build) */
int fx_prof_default(int on);   /* handles opened afterwards start with timing on (covers k_bgzf_inflate in fx_open_file) */
int fx_prof_reset(fx_handle *h);
int fx_prof_count(void);
const char *fx_prof_name(int id);
int fx_prof_read(fx_handle *h, int id, double *total_ms, int64_t *launches);

/* ----------------------------------------------------- multi-GPU stitching
 * Boundary summary of this shard for the single all-gather of SURVEY 8e.
 * Fixed size, plain integers; valid after fx_fasta_build / fx_fastq_build on
 * a handle configured with fx_set_shard.                                      */
typedef struct {
    int64_t base, n_bytes, is_last;
    int64_t n_nl;              /* entries of the shard's line table (virtual EOF newline included) */
    int64_t first_nl, second_nl, last_nl;  /* global offsets, -1 if absent                          */
    int64_t first_nl_prev;     /* byte before first_nl; -1 if first_nl == base (= previous shard's last_byte) */
    int64_t first_byte, last_byte;
    int64_t n_hdr;             /* FASTA header lines that START in the shard                        */
    int64_t first_hdr, last_hdr;
    int64_t lead_nl;           /* newlines before first_hdr (all of them if n_hdr == 0)             */
    int64_t lead_ws;           /* first ' ' or '\t' in [base, first_nl) (the whole shard if it has no newline), -1 if none */
    int64_t lead_v1, lead_c1;  /* lead lines with both newlines in the shard: the first two ...     */
    int64_t lead_v2, lead_c2;  /* ... distinct (len+1) values and how many lines have them          */
    /* the last record that starts in this shard, as far as the shard can tell */
    int64_t tail_e;            /* newline ending its header line, -1 if that is in a later shard    */
    int64_t tail_first_end;    /* newline ending its first sequence line, -1 if later               */
    int64_t tail_nl_after;     /* shard newlines after tail_e                                       */
    int64_t tail_bad;          /* bad lines counted locally (valid when tail_first_end >= 0)        */
    int64_t tail_elen, tail_dlen, tail_name_len;   /* dlen -1: header unterminated; name_len -1: no whitespace seen */
    int64_t lead_prev_nl;      /* second-to-last newline before first_hdr (of the shard if n_hdr == 0), -1 if lead_nl < 2   */
    int64_t second_last_nl;    /* second-to-last newline of the shard, -1 if n_nl < 2: with the field above, the length of
                                  the LAST line of a record that ends in a later shard (line-regular test, see above)     */
} fx_shard_summary;            /* 28 x int64: travels as one all-gather payload */

int fx_shard_summary_get(fx_handle *h, fx_shard_summary *out);

/* Device-resident variant of the same exchange, so that a sharded build never leaves the GPU between
 * the collective and the fetches: fx_shard_summary_dev ENQUEUES the summary (28 x int64) into d_out (a
 * device buffer of the caller, e.g. the send buffer of the all-gather); after the all-gather,
 * fx_fasta_stitch_dev ENQUEUES the completion of this rank's last record from the gathered summaries
 * d_all[world][28] (device) -- the integer logic of index.c:234-353 across shard cuts -- and rewrites
 * that row of the resident table.  Both run on the handle's stream: order them against the collective's
 * stream with fx_stream() (an opaque hipStream_t) and stream events.                                    */
int fx_shard_summary_dev(fx_handle *h, int64_t *d_out);
int fx_fasta_stitch_dev(fx_handle *h, const int64_t *d_all, int world, int rank, int full_name);
void *fx_stream(fx_handle *h);

/* After the all-gather the owner of a record that crosses shard cuts rewrites
 * that row of the resident table (and of what fx_fasta_table returns).  norm: bit 0 = the norm column, bit 1 = the
 * line-regular bit of that record (fx_fasta_line_regular). */
int fx_fasta_set_row(fx_handle *h, int64_t k, int64_t boff, int64_t blen, int64_t slen, int64_t llen,
                     int32_t elen, int32_t norm, int32_t dlen, int32_t name_len);

/* Host side of a batched fetch over byte-range shards (the "Fetch" half of SURVEY 8e; no device work, no handle):
 * n queries (record id, 0-based [start, stop)) against the GLOBAL record table -> byte ranges of the global stream by
 * the reference's line arithmetic (sequence.c:498-510: line-regular records exactly the bytes; the others the whole
 * record, sliced after despacing, sequence.c:100-110) -> the shard [bases[r], ends[r]) that holds the first byte of
 * each (that rank answers it) and the number of shards the range touches.  Outputs are in ROUTED order -- by
 * answering shard, then by position in the batch: order[k] = index of the query at routed position k;
 * shard_start[r] .. shard_start[r + 1] = the routed positions of shard r (n_shard + 1 entries); off / len / skip /
 * take / fl[k] = what fx_fetch_slices takes for that query (fl: flags_per_query or `flags`), cnt[k] = shards touched
 * (0: empty range or past the end of the stream, fread semantics index.c:689; > 1: the range crosses a cut and its
 * pieces are put together by the caller).  Replaces per-batch numpy passes of the Python layer (29 ms per 1 M
 * queries in round 1). */
int fx_shard_route(int64_t n, const int64_t *ids, const int64_t *starts, const int64_t *stops,
                   int64_t n_rec, const int64_t *boff, const int64_t *blen, const int64_t *llen, const int64_t *elen,
                   const uint8_t *reg, int n_shard, const int64_t *bases, const int64_t *ends,
                   int flags, const uint8_t *flags_per_query,
                   int64_t *order, int64_t *shard_start, int64_t *off, int64_t *len, int64_t *skip, int64_t *take,
                   uint8_t *fl, int32_t *cnt);

/* ------------------------------------------------- the collective under the C ABI
 * pyfastx_create_index (index.c:109-388) and pyfastx_fastq_create_index (fastq.c:8-182) scan one file in one
 * thread; the reference's only parallel story is "re-open per process" (docs/advance.rst:1-40).  Here a whole-file
 * build shards by byte range over the GPUs of a node, ONE process per GPU, and the ONE exchange it needs -- the
 * all-gather of the 28-word boundary summaries (RCCL over xGMI) -- lives in this library, so that a C caller
 * (multi-GPU) needs no Python and no torch:
 *
 *     rank 0: fx_comm_unique_id(id);  ... id reaches every rank (file, pipe, MPI_Bcast, env) ...
 *     all:    fx_comm_init(rank, world, id, device, &c);
 *             fx_open_file_range(path, size * rank / world, size * (rank + 1) / world - size * rank / world, 0, device, &h);
 *             fx_fasta_build_sharded(h, c, full_name, &summary);      rows of the records that START in this range,
 *             fx_fasta_table(h, FX_HOST, ...);                        the one that crosses the cut completed
 *
 * RCCL is loaded at run time (dlopen librccl.so.1, or the copy the process already carries; FX_RCCL_LIB overrides):
 * a single-GPU user never loads it; without it these entries return FX_EDEVICE.                               */
typedef struct fx_comm fx_comm;
int fx_comm_unique_id(uint8_t id[128]);                       /* ncclGetUniqueId: once, on one rank */
int fx_comm_init(int rank, int world, const uint8_t id[128], int device, fx_comm **out);   /* ncclCommInitRank: all ranks */
int fx_comm_destroy(fx_comm *c);
int fx_comm_rank(const fx_comm *c);
int fx_comm_world(const fx_comm *c);
/* nbytes (<= 65536) of every rank to every rank, host buffers (recv: world x nbytes) */
int fx_comm_allgather(fx_comm *c, const void *send, void *recv, int64_t nbytes);
/* scan + tables + boundary summary + ncclAllGather + stitch, all enqueued on the handle's stream; _begin returns at
 * once (device-side consumers may follow), fx_fasta_build_end reads the totals; full_name as in fx_fasta_build.    */
int fx_fasta_build_sharded_begin(fx_handle *h, fx_comm *c, int full_name);
int fx_fasta_build_sharded(fx_handle *h, fx_comm *c, int full_name, fx_fasta_summary *out);
int fx_comm_summaries(fx_comm *c, fx_handle *h, fx_shard_summary *out);    /* world entries, as the all-gather delivered them */
/* FASTQ: count pass, all-gather of (newlines of the core, last newline) per rank, rows with the global line numbering
 * (fx_set_halo first: a record that begins in the core is finished from the halo) */
int fx_fastq_build_sharded(fx_handle *h, fx_comm *c, fx_fastq_summary *out);

#ifdef __cplusplus
}
#endif
#endif
