/*
 * fx_oracle.c -- CPU ORACLE: TEST INFRASTRUCTURE ONLY (see fx_oracle.h).
 *
 * Line-at-a-time restatement of the reference's index scan, composition,
 * despace / reverse-complement and fetch arithmetic.  Deliberately written
 * as the same sequential state machine the reference uses (so it is easy to
 * audit against the cited lines), which is the opposite of how the HIP path
 * computes the same results (bulk newline table -> segmented reductions).
 *
 * Parity status: PINNED.  tests/golden/make_golden.py runs the real reference
 * (oracle/_ref, compiled from /root/reference/src) on the fixture files and
 * on generated edge cases and asserts this file agrees row for row; the
 * vectors are committed under tests/golden/ for machines without the
 * reference.
 */
#include <string.h>
#include <stdlib.h>
#include "fx_oracle.h"

/* ---- line reader: behaviour of ks_getuntil(ks, '\n', &line, 0), kseq.c:59-109.
 * '\r' is retained (delimiter 10 > KS_SEP_MAX, kseq.c:78-80,106).  A final
 * line without '\n' is still returned once; after that -1 (EOF). ---- */
typedef struct { const uint8_t *d; int64_t n, pos; } linerd;

static int next_line(linerd *r, const uint8_t **s, int64_t *l)
{
    const uint8_t *p;
    if (r->pos >= r->n) return -1;
    p = (const uint8_t *)memchr(r->d + r->pos, '\n', (size_t)(r->n - r->pos));
    *s = r->d + r->pos;
    if (p) { *l = (int64_t)(p - *s); r->pos = (int64_t)(p - r->d) + 1; }
    else   { *l = r->n - r->pos;     r->pos = r->n; }
    return 0;
}

/* ---- FASTA index: index.c:230-372 ---- */
int64_t fxo_fasta_index(const uint8_t *data, int64_t n, int full_name,
                        fxo_fasta_rec *out, int64_t cap, int64_t *seqlen)
{
    linerd r = { data, n, 0 };
    const uint8_t *s; int64_t l;
    int64_t position = 0, start = 0, seq_len = 0, line_len = 0, temp_len, bad_line = 0;
    int64_t total_seq = 0, total_len = 0, hoff = 0, name_len = 0, line_start;
    int line_end = 1, desc_len = 0;

#define EMIT(blen_expr) do {                                              \
        if (out && total_seq < cap) {                                     \
            fxo_fasta_rec *o = &out[total_seq];                           \
            o->hoff = hoff; o->boff = start; o->blen = (blen_expr);       \
            o->slen = seq_len; o->llen = line_len; o->elen = line_end;    \
            o->norm = (bad_line > 1) ? 0 : 1;       /* index.c:237,342 */ \
            o->dlen = desc_len; o->name_off = hoff + 1;                   \
            o->name_len = (int32_t)name_len;                              \
        }                                                                 \
        ++total_seq; total_len += seq_len;                                \
    } while (0)

    while (next_line(&r, &s, &l) == 0) {
        line_start = position;
        position += l + 1;                                  /* index.c:231 */
        if (l > 0 && s[0] == 62) {                          /* index.c:234 */
            if (start > 0) EMIT(position - start - l - 1);  /* index.c:243 */
            start = position;                               /* index.c:258 */
            seq_len = 0; line_len = 0; line_end = 1; bad_line = 0;
            if (s[l - 1] == '\r') line_end = 2;             /* index.c:266-269 */
            desc_len = (int)(l - line_end);                 /* index.c:271 */
            hoff = line_start;
            if (full_name) {
                name_len = desc_len;                        /* index.c:282-285 */
            } else {                                        /* index.c:289-293 */
                for (name_len = 0; name_len < desc_len; ++name_len)
                    if (s[1 + name_len] == ' ' || s[1 + name_len] == '\t') break;
            }
            continue;
        }
        temp_len = l + 1;                                   /* index.c:323 */
        if (line_len > 0 && line_len != temp_len) bad_line++;
        if (line_len == 0) line_len = temp_len;             /* index.c:330-332 */
        seq_len += l - line_end + 1;                        /* index.c:335-338 */
    }
    EMIT(position - start);                                 /* tail record, index.c:342-353 (unconditional) */
#undef EMIT
    if (seqlen) *seqlen = total_len;
    return total_seq;
}

/* ---- FASTA composition: fasta.c:901-950.  Bytes >= 128 index seq_comp[]
 * out of range in the reference (signed char, UB); the oracle ignores them
 * (documented divergence). ---- */
int64_t fxo_fasta_comp(const uint8_t *data, int64_t n, int64_t *comp, int64_t cap)
{
    linerd r = { data, n, 0 };
    const uint8_t *s; int64_t l, i, seqid = 0;
    while (next_line(&r, &s, &l) == 0) {
        if (l > 0 && s[0] == 62) { seqid++; continue; }    /* fasta.c:902-918 */
        if (seqid == 0 || seqid > cap || !comp) continue;
        for (i = 0; i < l; ++i)                             /* fasta.c:922-926 */
            if (s[i] < 128) comp[(seqid - 1) * 128 + s[i]]++;
    }
    return seqid;
}

/* ---- FASTQ index: fastq.c:89-171 ---- */
int64_t fxo_fastq_index(const uint8_t *data, int64_t n, fxo_fastq_rec *out,
                        int64_t cap, int64_t *size_out, int64_t *line_num_out)
{
    linerd r = { data, n, 0 };
    const uint8_t *s; int64_t l;
    int64_t pos = 0, size = 0, line_num = 0, rlen = 0, soff = 0, qoff, nrec = 0;
    int64_t name_off = 0, name_len = 0; int dlen = 0;
    while (next_line(&r, &s, &l) == 0) {
        ++line_num;
        switch (line_num % 4) {
        case 1:                                             /* fastq.c:99-117 */
            dlen = (int)l;
            name_off = pos + 1;
            name_len = l - 1;
            if (name_len > 0 && s[name_len] == '\r') --name_len;   /* name.s[name.l-1] */
            { const uint8_t *sp = (const uint8_t *)memchr(s + 1, ' ', (size_t)(name_len > 0 ? name_len : 0));
              if (sp) name_len = (int64_t)(sp - (s + 1)); }
            break;
        case 2:                                             /* fastq.c:121-129 */
            soff = pos;
            rlen = (l > 0 && s[l - 1] == '\r') ? l - 1 : l;
            size += rlen;
            break;
        case 0:                                             /* fastq.c:132-145 */
            qoff = pos;
            if (out && nrec < cap) {
                fxo_fastq_rec *o = &out[nrec];
                o->name_off = name_off; o->name_len = (int32_t)name_len;
                o->dlen = dlen; o->rlen = rlen; o->soff = soff; o->qoff = qoff;
            }
            ++nrec;
            break;
        }
        pos += l + 1;                                       /* fastq.c:148 */
    }
    if (size_out) *size_out = size;
    if (line_num_out) *line_num_out = line_num;
    return line_num / 4;                                    /* fastq.c:159 */
}

/* ---- FASTQ composition: fastq.c:715-774 ---- */
void fxo_fastq_composition(const uint8_t *data, int64_t n, fxo_fastq_comp *o)
{
    linerd r = { data, n, 0 };
    const uint8_t *s; int64_t l, i, line_num = 0;
    int minqs = 104, maxqs = 33, phred = 0;                 /* fastq.c:667-668 */
    int64_t maxlen = 0, minlen = 10000000000LL;
    memset(o, 0, sizeof(*o));
    while (next_line(&r, &s, &l) == 0) {
        ++line_num;
        if (line_num % 4 == 2) {
            for (i = 0; i < l; i++) {
                switch (s[i]) {                             /* fastq.c:722-729 */
                case 65: ++o->a; break;
                case 67: ++o->c; break;
                case 71: ++o->g; break;
                case 84: ++o->t; break;
                case 13: break;
                default: ++o->n;
                }
            }
        } else if (line_num % 4 == 0) {
            for (i = 0; i < l; i++) {                       /* fastq.c:733-745 */
                int q = (int)(signed char)s[i];
                if (s[i] == 13) { --l; continue; }
                if (q < minqs) minqs = q;
                if (q > maxqs) maxqs = q;
            }
            if (l > maxlen) maxlen = l;
            if (l < minlen) minlen = l;
        }
    }
    if (maxqs > 74) phred = 64;                             /* fastq.c:768-774 */
    if (minqs < 59) phred = 33;
    o->maxlen = maxlen; o->minlen = minlen;
    o->minqs = minqs; o->maxqs = maxqs; o->phred = phred;
}

/* ---- despace: util.c:157-194 (jump_table drops 10, 13, 32 only) ---- */
int64_t fxo_despace(uint8_t *buf, int64_t n, int upper)
{
    int64_t i = 0, j = 0;
    while (i < n) {
        uint8_t c = buf[i++];
        buf[j] = (upper && c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c;  /* Py_TOUPPER */
        j += !(c == 10 || c == 13 || c == 32);
    }
    return j;
}

/* ---- complement LUT: util.c:228-237, derived from the IUPAC table in the
 * comment block util.c:204-226 rather than copied as numbers. ---- */
static uint8_t comp_lut[256];
static int comp_ready = 0;
static void comp_init(void)
{
    static const char *pairs = "ATCGMKRYVBHD";   /* A<->T C<->G M<->K R<->Y V<->B H<->D */
    int i;
    for (i = 0; i < 256; ++i) comp_lut[i] = (uint8_t)i;      /* W S N and non-letters: identity */
    for (i = 0; pairs[i]; i += 2) {
        uint8_t x = (uint8_t)pairs[i], y = (uint8_t)pairs[i + 1];
        comp_lut[x] = y; comp_lut[y] = x;
        comp_lut[x + 32] = (uint8_t)(y + 32); comp_lut[y + 32] = (uint8_t)(x + 32);
    }
    comp_lut['U'] = 'A'; comp_lut['u'] = 'a';                /* util.c:233-236 */
    comp_ready = 1;
}

void fxo_revcomp(uint8_t *buf, int64_t n, int mode)
{
    int64_t i;
    if (!comp_ready) comp_init();
    if (mode & 2) for (i = 0; i < n; ++i) buf[i] = comp_lut[buf[i]];     /* util.c:263-269 */
    if (mode & 1) for (i = 0; i < n / 2; ++i) {                          /* util.c:251-261 */
        uint8_t t = buf[i]; buf[i] = buf[n - 1 - i]; buf[n - 1 - i] = t;
    }
}

/* ---- slice -> byte range: sequence.c:498-510, fasta.c:293-320 ---- */
void fxo_slice_range(int64_t boff, int64_t llen, int32_t elen, int64_t start,
                     int64_t stop, int64_t *off, int64_t *blen)
{
    int64_t bpl = llen - elen;
    int64_t before_s = start / bpl, before_e = stop / bpl;
    *off = boff + start + (int64_t)elen * before_s;
    *blen = (stop - start) + (before_e - before_s) * elen;
}

/* ---- fetch: index.c:683-707 (read + despace[+upper]) then the getter's copy
 * of seq_len bytes (sequence.c:346-347) and reverse/complement
 * (sequence.c:352-398).  Reads past EOF return what exists (fread semantics). ---- */
int64_t fxo_fetch(const uint8_t *data, int64_t n, int64_t off, int64_t blen,
                  int64_t slen, int flags, uint8_t *out)
{
    int64_t avail, m;
    uint8_t *tmp;
    if (off < 0 || off >= n || blen <= 0) return 0;
    avail = (off + blen > n) ? n - off : blen;
    tmp = (uint8_t *)malloc((size_t)avail + 1);
    memcpy(tmp, data + off, (size_t)avail);
    m = fxo_despace(tmp, avail, flags & 1);
    if (m > slen) m = slen;
    fxo_revcomp(tmp, m, ((flags & 2) ? 1 : 0) | ((flags & 4) ? 2 : 0));
    memcpy(out, tmp, (size_t)m);
    free(tmp);
    return m;
}

/* ---- quali: read.c:251-278 ---- */
void fxo_quali(const uint8_t *data, int64_t qoff, int64_t rlen, int phred, int8_t *out)
{
    int64_t i;
    if (!phred) phred = 33;                                 /* read.c:268 */
    for (i = 0; i < rlen; ++i) out[i] = (int8_t)((int)(signed char)data[qoff + i] - phred);
}

/* ------------------------------------------------------------------ kseq_read (Fastx)
 * ks_getuntil2 in line mode with append = 1 (kseq.c:59-109): the bytes up to the next '\n' (or the end of the
 * stream) are appended to dst[0..*l); -1 when nothing at all is left; a trailing '\r' goes only from a string that
 * is longer than one byte (kseq.c:106). */
static int64_t kq_line(const uint8_t *d, int64_t n, int64_t *p, uint8_t *dst, int64_t *l) {
    if (*p >= n) return -1;
    const uint8_t *e = memchr(d + *p, '\n', (size_t)(n - *p));
    const int64_t end = e ? (int64_t)(e - d) : n;
    memcpy(dst + *l, d + *p, (size_t)(end - *p));
    *l += end - *p;
    *p = e ? end + 1 : n;
    if (*l > 1 && dst[*l - 1] == '\r') --*l;
    return *l;
}
static int kq_space(int c) { return c == ' ' || (c >= 9 && c <= 13); }   /* isspace() in the C locale */

int64_t fxo_kseq(const uint8_t *d, int64_t n, fxo_kseq_rec *out, int64_t cap, uint8_t *seqbuf, uint8_t *qualbuf,
                 int *end_code) {
    int64_t p = 0, nrec = 0, so = 0, qo = 0;
    int last = 0, code = -1;
    for (;;) {
        fxo_kseq_rec r;
        if (last == 0) {                                   /* kseq.c:142-146: jump to the next header character */
            while (p < n && d[p] != '>' && d[p] != '@') ++p;
            if (p >= n) { code = -1; break; }
            last = d[p++];
        }
        if (p >= n) { code = -1; break; }                  /* kseq.c:148: nothing behind the header character */
        int64_t q = p;
        while (q < n && !kq_space(d[q])) ++q;
        const int delim = q < n ? d[q] : 0;
        r.name_off = p; r.name_len = q - p;
        p = q < n ? q + 1 : n;
        r.com_off = p; r.com_len = -1;
        if (delim != '\n' && p < n) {                      /* kseq.c:149 */
            const uint8_t *e = memchr(d + p, '\n', (size_t)(n - p));
            const int64_t end = e ? (int64_t)(e - d) : n;
            r.com_len = end - p;
            if (r.com_len > 1 && d[end - 1] == '\r') --r.com_len;
            p = e ? end + 1 : n;
        }
        uint8_t *s = seqbuf + so;                          /* kseq.c:154-158 */
        int64_t sl = 0;
        int c = -1;
        for (;;) {
            if (p >= n) { c = -1; break; }
            c = d[p++];
            if (c == '>' || c == '+' || c == '@') break;
            if (c == '\n') continue;
            s[sl++] = (uint8_t)c;
            (void)kq_line(d, n, &p, s, &sl);
        }
        if (c == '>' || c == '@') last = c;                /* kseq.c:159 */
        r.seq_off = so; r.seq_len = sl; r.qual_off = qo; r.qual_len = -1;
        so += sl;
        if (c != '+') {                                    /* kseq.c:166: FASTA */
            if (nrec < cap && out) out[nrec] = r;
            ++nrec;
            continue;
        }
        while (p < n && d[p] != '\n') ++p;                 /* kseq.c:171-172: the rest of the '+' line */
        if (p >= n) { code = -2; break; }
        ++p;
        uint8_t *ql = qualbuf + qo;
        int64_t qlen = 0, got, calls = 0;
        do { got = kq_line(d, n, &p, ql, &qlen); ++calls; } while (got >= 0 && qlen < sl);   /* kseq.c:173 */
        last = 0;                                          /* kseq.c:175 */
        if (sl != qlen) { code = -2; break; }              /* kseq.c:176 */
        /* the stream ended behind the '+' line of a record without bases: ks_getuntil2 returned -1 before it touched
         * the quality buffer, which still holds the previous record's string (or nothing defined at all) */
        r.qual_len = (calls == 1 && got < 0) ? -2 : qlen;
        qo += qlen;
        if (nrec < cap && out) out[nrec] = r;
        ++nrec;
    }
    if (end_code) *end_code = code;
    return nrec;
}
