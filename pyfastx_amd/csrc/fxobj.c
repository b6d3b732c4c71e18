/* fxobj.c -- the per-object getter path of pyfastx_amd.Fasta / Sequence as CPython types (module pyfastx_amd._fxobj).
 *
 * The reference's own benchmark idiom is one getter at a time -- `fa[name][s:e].seq`
 * (benchmark/pyfastx_fasta_extract_subsequences.py:8-12) -- and there every step is C: pyfastx_fasta_subscript
 * (fasta.c:521-546) -> pyfastx_index_get_seq_by_name (index.c:527-566), pyfastx_sequence_subscript (sequence.c:412-517),
 * pyfastx_sequence_seq (sequence.c:337-350) -> pyfastx_index_fill_cache (index.c:694-707): 4.1 us on the test box.
 * Here the bytes come from the resident k_mailbox kernel (fx_fetch_one: a request line in pinned memory, the answer back
 * through pinned memory -- two PCIe round trips, ~4 us), so everything around it has to cost next to nothing: these two
 * base types do the three steps of the idiom without entering the interpreter --
 *
 *   FastaCore.__getitem__(str)   the row of the name from a dict (filled by the Python side's SQLite probe, once per name)
 *                                -> a new Sequence, fields copied from the row, no __init__
 *   SeqCore.__getitem__(slice)   the arithmetic of sequence.c:459-493 -> a new Sequence
 *   SeqCore.seq                  line-regular record, a slice: the byte range of sequence.c:498-510, fx_fetch_one straight
 *                                into a stack buffer, PyUnicode_DecodeLatin1
 *
 *   (round 4) on a PLAIN file whose stream is staged, a getter of at most 64 KiB is answered from the page cache -- pread,
 *   despace / upper-case / complement / reverse in C (util.c:157-269) -- as the reference answers it (index.c:683-707); the
 *   resident kernel keeps the gzip inputs, and the GPU everything batched (hard part 7: "a host fast-path")
 *
 * -- and hand everything else (integer subscripts, whole records, records with an odd line, the first touch that stages the
 * file, ...) to the methods of the Python subclasses (api.Fasta / api.Sequence: `_getitem_slow`, `_subscript_slow`, `_get`),
 * which keep all the behaviour they had.  libfxgpu.so is not linked: the address of fx_fetch_one comes from the ctypes
 * binding (set_api), so the one library instance of the process is used.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <structmember.h>
#include <stdint.h>
#include <time.h>
#include <dlfcn.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>

typedef int (*fetch_one_fn)(void *h, int64_t off, int64_t blen, int64_t skip, int64_t take, int flags, uint8_t *dst, int64_t *out_len);
static fetch_one_fn g_fetch_one = NULL;
static PyTypeObject *g_seq_type = NULL;          /* api.Sequence (subclass of SeqCore) */

#define FX_GETTER_CAP 65536                      /* larger answers take the Python path (its buffers) */

typedef struct {
    PyObject_HEAD
    PyObject *rows;                              /* dict: name -> (ID, chrom, boff, blen, slen, llen, elen, norm, dlen[, reg]) */
    unsigned long long handle;                   /* fx_handle* of the staged stream, 0 until it is staged */
    int upper;                                   /* Fasta(uppercase=True) */
    int fd;                                      /* the plain file, for single getters from the page cache; -1: none (gzip input, not staged yet) */
} FastaCore;

typedef struct {
    PyObject_HEAD
    PyObject *fa, *name;
    PyObject *pre;                               /* the whole sequence, when it came with the iterator's batch (Fasta.__iter__) */
    long long id, offset, byte_len, full_len, line_len, end_len, normal, desc_len, start, end, seq_len;
    char complete;
    signed char reg;                             /* line-regular: -1 not known yet, 0, 1 */
} SeqCore;

static PyTypeObject FastaCoreType;

/* api.Sequence is a heap type and so a GC type, but a Sequence refers to its Fasta, its name and a string -- nothing that can
 * lead back to it (the subclass has __slots__ = (): no instance dict): it stays out of the cyclic collector's lists, and its
 * allocations out of the collector's counts (see read_untrack) */
static void seq_untrack(PyObject *o)
{
    if (PyType_HasFeature(Py_TYPE(o), Py_TPFLAGS_HAVE_GC) && PyObject_GC_IsTracked(o)) PyObject_GC_UnTrack(o);
}

/* ------------------------------------------------------------------ SeqCore */
static void seq_dealloc(SeqCore *s)
{
    Py_XDECREF(s->fa);
    Py_XDECREF(s->name);
    Py_XDECREF(s->pre);
    Py_TYPE(s)->tp_free((PyObject *)s);
}

static PyObject *seq_new(PyTypeObject *type, PyObject *args, PyObject *kw)
{
    SeqCore *s = (SeqCore *)type->tp_alloc(type, 0);
    (void)args; (void)kw;
    if (s) { if (type->tp_dictoffset == 0) seq_untrack((PyObject *)s); s->fa = Py_NewRef(Py_None); s->name = Py_NewRef(Py_None); s->reg = -1; s->start = 1; }
    return (PyObject *)s;
}

static SeqCore *seq_like(SeqCore *p, long long start, long long end, int complete)
{
    PyTypeObject *tp = Py_TYPE(p);
    SeqCore *s = (SeqCore *)tp->tp_alloc(tp, 0);
    if (!s) return NULL;
    if (tp->tp_dictoffset == 0) seq_untrack((PyObject *)s);
    s->fa = Py_NewRef(p->fa); s->name = Py_NewRef(p->name);
    s->id = p->id; s->offset = p->offset; s->byte_len = p->byte_len; s->full_len = p->full_len; s->line_len = p->line_len;
    s->end_len = p->end_len; s->normal = p->normal; s->desc_len = p->desc_len; s->reg = p->reg;
    s->start = start; s->end = end; s->complete = (char)complete;
    s->seq_len = complete ? p->full_len : end - start + 1;
    return s;
}

/* pyfastx_sequence_subscript (sequence.c:412-517), from ABSOLUTE coordinates */
static PyObject *seq_subscript(SeqCore *s, PyObject *item)
{
    if (PySlice_Check(item)) {
        Py_ssize_t a, b, step;
        if (PySlice_Unpack(item, &a, &b, &step) < 0) return NULL;
        PySlice_AdjustIndices((Py_ssize_t)(s->seq_len > 0 ? s->seq_len : 0), &a, &b, step);
        if (step != 1) { PyErr_SetString(PyExc_ValueError, "slice step cannot > 1"); return NULL; }
        if (b < a) b = a;
        return (PyObject *)seq_like(s, s->start + a, s->start + b - 1, s->complete && (long long)(b - a) == s->seq_len);
    }
    return PyObject_CallMethod((PyObject *)s, "_subscript_slow", "O", item);
}

static Py_ssize_t seq_length(SeqCore *s) { return (Py_ssize_t)(s->seq_len > 0 ? s->seq_len : 0); }

/* ---- the byte work of ONE getter on the host: remove_space / remove_space_uppercase (util.c:157-194: bytes 10, 13 and 32
 * go, nothing else), complement_seq / reverse_seq / reverse_complement_seq (util.c:228-269: IUPAC, case kept, U -> A; bytes
 * of 128 and more stay as they are) */
static const uint8_t *comp_table(void)
{
    static uint8_t t[256];
    static int ready = 0;
    if (!ready) {
        int i;
        for (i = 0; i < 256; ++i) {
            uint8_t c = (uint8_t)i, u = (uint8_t)(c & 0xDF), r = c;
            if (c < 128 && u >= 'A' && u <= 'Z') {
                uint8_t m = u;
                switch (u) {
                case 'A': m = 'T'; break; case 'T': m = 'A'; break; case 'U': m = 'A'; break;
                case 'C': m = 'G'; break; case 'G': m = 'C'; break;
                case 'M': m = 'K'; break; case 'K': m = 'M'; break;
                case 'R': m = 'Y'; break; case 'Y': m = 'R'; break;
                case 'V': m = 'B'; break; case 'B': m = 'V'; break;
                case 'H': m = 'D'; break; case 'D': m = 'H'; break;
                default: break;
                }
                r = (uint8_t)(m | (c & 0x20));
            }
            t[i] = r;
        }
        ready = 1;
    }
    return t;
}
/* n raw bytes in b -> at most `take` kept bytes in place, flags as in fxgpu.h (1 upper, 2 reverse, 4 complement, 8 raw) */
static Py_ssize_t host_bytes(uint8_t *b, Py_ssize_t n, Py_ssize_t take, int flags)
{
    Py_ssize_t i, k = 0;
    if (flags & 8) k = n < take ? n : take;
    else
        for (i = 0; i < n && k < take; ++i) {
            const uint8_t c = b[i];
            if (c == 10 || c == 13 || c == 32) continue;
            b[k++] = (flags & 1) && c >= 'a' && c <= 'z' ? (uint8_t)(c - 32) : c;
        }
    if (flags & 4) { const uint8_t *t = comp_table(); for (i = 0; i < k; ++i) b[i] = t[b[i]]; }
    if (flags & 2) for (i = 0; i < k / 2; ++i) { const uint8_t x = b[i]; b[i] = b[k - 1 - i]; b[k - 1 - i] = x; }
    return k;
}
/* n bytes at `off` of the plain file -> bytes read (short at the end of the file), -1 on an error */
static Py_ssize_t host_read(int fd, uint8_t *b, Py_ssize_t n, long long off)
{
    Py_ssize_t got = 0;
    while (got < n) {
        const ssize_t r = pread(fd, b + got, (size_t)(n - got), (off_t)(off + got));
        if (r < 0) return -1;
        if (r == 0) break;
        got += r;
    }
    return got;
}

/* the bytes of a slice of a line-regular record, flags as in fxgpu.h (1 upper, 2 reverse, 4 complement); NULL + no error
 * set: not a case for the fast path */
static PyObject *seq_fast(SeqCore *s, int flags)
{
    FastaCore *fa;
    long long bpl, a, b, off, bl;
    int64_t got = 0;
    uint8_t buf[FX_GETTER_CAP];
    if (s->complete || s->reg != 1 || s->seq_len <= 0 || s->seq_len > FX_GETTER_CAP || !s->fa) return NULL;
    if (!PyObject_TypeCheck(s->fa, &FastaCoreType)) return NULL;
    fa = (FastaCore *)s->fa;
    if (!fa->handle) return NULL;
    bpl = s->line_len - s->end_len;
    if (bpl <= 0) return NULL;
    a = s->start - 1; b = s->end;
    off = s->offset + a + s->end_len * (a / bpl);                         /* sequence.c:498-510 */
    bl = (b - a) + (b / bpl - a / bpl) * s->end_len;
    if (fa->fd >= 0 && bl <= FX_GETTER_CAP) {                             /* a plain file: the page cache answers (index.c:683-707) */
        const Py_ssize_t n = host_read(fa->fd, buf, (Py_ssize_t)bl, off);
        if (n >= 0) return PyUnicode_DecodeLatin1((const char *)buf, host_bytes(buf, n, (Py_ssize_t)s->seq_len, flags | (fa->upper ? 1 : 0)), NULL);
    }
    if (!g_fetch_one) return NULL;
    if (g_fetch_one((void *)(uintptr_t)fa->handle, off, bl, 0, s->seq_len, flags | (fa->upper ? 1 : 0), buf, &got) != 0) return NULL;
    return PyUnicode_DecodeLatin1((const char *)buf, (Py_ssize_t)got, NULL);
}

static PyObject *seq_get(SeqCore *s, int flags)
{
    PyObject *r;
    if (flags == 0 && s->complete && s->pre) return Py_NewRef(s->pre);     /* came with the iterator's batch */
    if (s->reg < 0 && !s->complete && s->seq_len > 0) {                   /* ask the Python side once per object (it caches per record) */
        PyObject *v = PyObject_CallMethod((PyObject *)s, "_line_regular", NULL);
        if (!v) return NULL;
        s->reg = (signed char)(PyObject_IsTrue(v) ? 1 : 0);
        Py_DECREF(v);
        if (s->fa && s->name && PyObject_TypeCheck(s->fa, &FastaCoreType)) {  /* ... and keep it with the row, for the next fa[name] */
            FastaCore *fa = (FastaCore *)s->fa;
            PyObject *row = fa->rows ? PyDict_GetItem(fa->rows, s->name) : NULL;
            if (row && PyTuple_Check(row) && PyTuple_GET_SIZE(row) == 9) {
                PyObject *nr = PyTuple_New(10);
                if (nr) {
                    for (int i = 0; i < 9; ++i) PyTuple_SET_ITEM(nr, i, Py_NewRef(PyTuple_GET_ITEM(row, i)));
                    PyTuple_SET_ITEM(nr, 9, PyLong_FromLong(s->reg));
                    PyDict_SetItem(fa->rows, s->name, nr);
                    Py_DECREF(nr);
                }
            }
        }
    }
    r = seq_fast(s, flags);
    if (r || PyErr_Occurred()) return r;
    return PyObject_CallMethod((PyObject *)s, "_get", "i", flags);
}

static PyObject *seq_seq(SeqCore *s, void *c) { (void)c; return seq_get(s, 0); }
static PyObject *seq_reverse(SeqCore *s, void *c) { (void)c; return seq_get(s, 2); }
static PyObject *seq_complement(SeqCore *s, void *c) { (void)c; return seq_get(s, 4); }
static PyObject *seq_antisense(SeqCore *s, void *c) { (void)c; return seq_get(s, 6); }

static PyMemberDef seq_members[] = {
    {"id", T_LONGLONG, offsetof(SeqCore, id), 0, NULL}, {"_offset", T_LONGLONG, offsetof(SeqCore, offset), 0, NULL},
    {"_byte_len", T_LONGLONG, offsetof(SeqCore, byte_len), 0, NULL}, {"_full_len", T_LONGLONG, offsetof(SeqCore, full_len), 0, NULL},
    {"_line_len", T_LONGLONG, offsetof(SeqCore, line_len), 0, NULL}, {"_end_len", T_LONGLONG, offsetof(SeqCore, end_len), 0, NULL},
    {"_normal", T_LONGLONG, offsetof(SeqCore, normal), 0, NULL}, {"_desc_len", T_LONGLONG, offsetof(SeqCore, desc_len), 0, NULL},
    {"start", T_LONGLONG, offsetof(SeqCore, start), 0, NULL}, {"end", T_LONGLONG, offsetof(SeqCore, end), 0, NULL},
    {"_seq_len", T_LONGLONG, offsetof(SeqCore, seq_len), 0, NULL}, {"_complete", T_BOOL, offsetof(SeqCore, complete), 0, NULL},
    {"_reg", T_BYTE, offsetof(SeqCore, reg), 0, NULL}, {"_prefetched", T_OBJECT, offsetof(SeqCore, pre), 0, NULL}, {NULL, 0, 0, 0, NULL}};
/* _fa / _name: plain attributes, except that they cannot be deleted (the C paths read them without asking) */
static PyObject *seq_get_fa(SeqCore *s, void *c) { (void)c; return Py_NewRef(s->fa ? s->fa : Py_None); }
static PyObject *seq_get_name(SeqCore *s, void *c) { (void)c; return Py_NewRef(s->name ? s->name : Py_None); }
static int seq_set_fa(SeqCore *s, PyObject *v, void *c)
{
    (void)c;
    if (!v) { PyErr_SetString(PyExc_TypeError, "_fa cannot be deleted"); return -1; }
    Py_XSETREF(s->fa, Py_NewRef(v));
    return 0;
}
static int seq_set_name(SeqCore *s, PyObject *v, void *c)
{
    (void)c;
    if (!v) { PyErr_SetString(PyExc_TypeError, "_name cannot be deleted"); return -1; }
    Py_XSETREF(s->name, Py_NewRef(v));
    return 0;
}
static PyGetSetDef seq_getset[] = {
    {"_fa", (getter)seq_get_fa, (setter)seq_set_fa, NULL, NULL}, {"_name", (getter)seq_get_name, (setter)seq_set_name, NULL, NULL},
    {"seq", (getter)seq_seq, NULL, NULL, NULL}, {"reverse", (getter)seq_reverse, NULL, NULL, NULL},
    {"complement", (getter)seq_complement, NULL, NULL, NULL}, {"antisense", (getter)seq_antisense, NULL, NULL, NULL},
    {NULL, NULL, NULL, NULL, NULL}};
static PyMappingMethods seq_mapping = {(lenfunc)seq_length, (binaryfunc)seq_subscript, NULL};
static PyTypeObject SeqCoreType = {
    PyVarObject_HEAD_INIT(NULL, 0).tp_name = "pyfastx_amd._fxobj.SeqCore", .tp_basicsize = sizeof(SeqCore),
    .tp_dealloc = (destructor)seq_dealloc, .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_BASETYPE, .tp_new = seq_new,
    .tp_members = seq_members, .tp_getset = seq_getset, .tp_as_mapping = &seq_mapping,
    .tp_doc = "fields and fast paths of pyfastx_amd.Sequence"};

/* ---------------------------------------------------------------- FastaCore */
static void fasta_dealloc(FastaCore *f)
{
    if (f->fd >= 0) close(f->fd);
    Py_XDECREF(f->rows);
    Py_TYPE(f)->tp_free((PyObject *)f);
}

static PyObject *fasta_new(PyTypeObject *type, PyObject *args, PyObject *kw)
{
    FastaCore *f = (FastaCore *)type->tp_alloc(type, 0);
    (void)args; (void)kw;
    if (f) { f->fd = -1; f->rows = PyDict_New(); if (!f->rows) { Py_DECREF(f); return NULL; } }
    return (PyObject *)f;
}

static long long tup_ll(PyObject *t, Py_ssize_t i) { return PyLong_AsLongLong(PyTuple_GET_ITEM(t, i)); }

/* pyfastx_fasta_subscript (fasta.c:521-546) for a name whose row is known */
static PyObject *fasta_subscript(FastaCore *f, PyObject *key)
{
    if (PyUnicode_CheckExact(key) && g_seq_type && f->rows) {
        PyObject *row = PyDict_GetItemWithError(f->rows, key);
        if (!row && PyErr_Occurred()) return NULL;
        if (row && PyTuple_Check(row) && PyTuple_GET_SIZE(row) >= 9) {
            SeqCore *s = (SeqCore *)g_seq_type->tp_alloc(g_seq_type, 0);
            if (!s) return NULL;
            if (g_seq_type->tp_dictoffset == 0) seq_untrack((PyObject *)s);
            s->fa = Py_NewRef((PyObject *)f);
            s->name = Py_NewRef(PyTuple_GET_ITEM(row, 1));
            s->id = tup_ll(row, 0); s->offset = tup_ll(row, 2); s->byte_len = tup_ll(row, 3); s->full_len = tup_ll(row, 4);
            s->line_len = tup_ll(row, 5); s->end_len = tup_ll(row, 6); s->normal = tup_ll(row, 7); s->desc_len = tup_ll(row, 8);
            s->reg = PyTuple_GET_SIZE(row) > 9 ? (signed char)tup_ll(row, 9) : -1;
            s->start = 1; s->end = s->full_len; s->complete = 1; s->seq_len = s->full_len;
            if (PyErr_Occurred()) { Py_DECREF(s); return NULL; }
            return (PyObject *)s;
        }
    }
    return PyObject_CallMethod((PyObject *)f, "_getitem_slow", "O", key);
}

static PyObject *fasta_tag(FastaCore *f, void *c) { (void)f; (void)c; Py_RETURN_TRUE; }
/* _core_stage(handle, path | None): the stream is staged under this fx_handle (0: it is gone -- Blob.close -- and the C
 * getters stop using it); path: the PLAIN file behind it, opened here for the getters that the page cache answers */
static PyObject *fasta_stage(FastaCore *f, PyObject *args)
{
    unsigned long long h = 0;
    PyObject *path = Py_None;
    if (!PyArg_ParseTuple(args, "K|O", &h, &path)) return NULL;
    if (f->fd >= 0) { close(f->fd); f->fd = -1; }
    f->handle = h;
    if (h && path != Py_None) {
        PyObject *b = NULL;
        if (!PyUnicode_FSConverter(path, &b)) return NULL;
        f->fd = open(PyBytes_AS_STRING(b), O_RDONLY | O_CLOEXEC);         /* -1: the resident kernel answers */
        Py_DECREF(b);
    }
    Py_RETURN_NONE;
}
static PyMethodDef fasta_methods[] = {
    {"_core_stage", (PyCFunction)fasta_stage, METH_VARARGS, "_core_stage(handle, plain path | None)"}, {NULL, NULL, 0, NULL}};
static PyMemberDef fasta_members[] = {
    {"_rows_by_name", T_OBJECT_EX, offsetof(FastaCore, rows), READONLY, NULL}, {"_core_handle", T_ULONGLONG, offsetof(FastaCore, handle), READONLY, NULL},
    {"_core_fd", T_INT, offsetof(FastaCore, fd), READONLY, NULL},
    {"_core_upper", T_INT, offsetof(FastaCore, upper), 0, NULL}, {NULL, 0, 0, 0, NULL}};
static PyGetSetDef fasta_getset[] = {{"_core_tag", (getter)fasta_tag, NULL, NULL, NULL}, {NULL, NULL, NULL, NULL, NULL}};
static PyMappingMethods fasta_mapping = {NULL, (binaryfunc)fasta_subscript, NULL};     /* (__len__ stays with the Python class) */
static PyTypeObject FastaCoreType = {
    PyVarObject_HEAD_INIT(NULL, 0).tp_name = "pyfastx_amd._fxobj.FastaCore", .tp_basicsize = sizeof(FastaCore),
    .tp_dealloc = (destructor)fasta_dealloc, .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_BASETYPE, .tp_new = fasta_new,
    .tp_members = fasta_members, .tp_methods = fasta_methods, .tp_getset = fasta_getset, .tp_as_mapping = &fasta_mapping,
    .tp_doc = "name -> row cache and subscript fast path of pyfastx_amd.Fasta"};

/* ------------------------------------------------------------------- module */
static PyObject *mod_set_api(PyObject *m, PyObject *args)
{
    unsigned long long addr = 0;
    PyObject *seq_type = NULL;
    (void)m;
    if (!PyArg_ParseTuple(args, "KO", &addr, &seq_type)) return NULL;
    if (!PyType_Check(seq_type) || !PyType_IsSubtype((PyTypeObject *)seq_type, &SeqCoreType)) {
        PyErr_SetString(PyExc_TypeError, "the sequence type must derive from SeqCore");
        return NULL;
    }
    g_fetch_one = (fetch_one_fn)(uintptr_t)addr;
    Py_XDECREF(g_seq_type);
    g_seq_type = (PyTypeObject *)Py_NewRef(seq_type);
    Py_RETURN_NONE;
}

/* the floor under any single getter: n calls of fx_fetch_one from C, nothing else -- (handle, off, blen, take, n) -> us per call */
static PyObject *mod_bench(PyObject *m, PyObject *args)
{
    unsigned long long h = 0;
    long long off = 0, blen = 0, take = 0, n = 0, i;
    uint8_t buf[FX_GETTER_CAP];
    int64_t got = 0;
    struct timespec t0, t1;
    (void)m;
    if (!PyArg_ParseTuple(args, "KLLLL", &h, &off, &blen, &take, &n)) return NULL;
    if (!g_fetch_one || !h || take > FX_GETTER_CAP || n <= 0) { PyErr_SetString(PyExc_ValueError, "bad argument"); return NULL; }
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (i = 0; i < n; ++i)
        if (g_fetch_one((void *)(uintptr_t)h, off + (i & 1023) * 61, blen, 0, take, 0, buf, &got) != 0) { PyErr_SetString(PyExc_RuntimeError, "fx_fetch_one failed"); return NULL; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return PyFloat_FromDouble(((double)(t1.tv_sec - t0.tv_sec) * 1e9 + (double)(t1.tv_nsec - t0.tv_nsec)) / 1e3 / (double)n);
}

/* ------------------------------------------------------------------ Fastx: the tuples of one gathered batch
 * What pyfastx_fastx_fasta* / pyfastx_fastx_fastq* build per record (fastx.c:6-30), for the k records of a batch in one
 * call: fastx_batch(hdr, hdr_off int64[k+1], seq, qual | None, recs (fx_kseq_rec[k]), fastq, with_comment, state) -> list.
 * state = [comment buffer exists (bool), the reference's quality buffer (str | None)] carries over between batches. */
typedef struct { int64_t hdr_off, hdr_line, seq_len, seq_cum; uint32_t hdr_len, s_n, q_n, flags; } kq_rec;

static PyObject *cstr_text(const uint8_t *p, int64_t n)          /* Py_BuildValue "s": up to the first NUL */
{
    /* Bases and qualities are ASCII without a NUL in every file there is: one pass over the bytes, eight at a time, finds out, and
     * the string is then made without the decoder (a compact ASCII object + memcpy).  Anything else -- a NUL, a byte above 127 --
     * takes the route "s" takes. */
    int64_t i = 0;
    uint64_t bad = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        bad |= (w & 0x8080808080808080ull) | ((w - 0x0101010101010101ull) & ~w & 0x8080808080808080ull);
    }
    for (; i < n; ++i) bad |= (uint64_t)(p[i] >= 0x80 || p[i] == 0);
    if (!bad) {
        PyObject *u = PyUnicode_New((Py_ssize_t)n, 127);
        if (u && n) memcpy(PyUnicode_1BYTE_DATA(u), p, (size_t)n);
        return u;
    }
    {
        const uint8_t *z = (const uint8_t *)memchr(p, 0, (size_t)n);
        return PyUnicode_DecodeUTF8((const char *)p, z ? (Py_ssize_t)(z - p) : (Py_ssize_t)n, "surrogateescape");
    }
}
static int kq_isspace(int c) { return c == ' ' || (c >= 9 && c <= 13); }

/* one record -> its tuple; *buffered / *last_qual carry the reference's comment / quality buffer state */
/* with_comment: 0 no comment element; 1 the comment as the last element (Fastx, fastx.c:10-12, 28-30); 2 no element, but a
 * non-empty comment is joined to the name with ONE SPACE, whatever the delimiter was (the index-free iteration of Fasta /
 * Fastq with full_name: PyUnicode_FromFormat("%s %s", name, comment), index.c:624-664, fastq.c:607-622) */
static PyObject *kq_tuple(const uint8_t *h, int64_t hl, const uint8_t *s_ptr, const uint8_t *q_ptr, const kq_rec *r, int fastq, int with_comment,
                          int *buffered, PyObject **last_qual)
{
    int64_t nl = 0, cl = -1;                                     /* name length; comment length, -1: buffer not written */
    PyObject *name, *s, *com = NULL, *t;
    while (nl < hl && !kq_isspace(h[nl])) ++nl;                  /* kseq.c:148: the name ends at the first isspace() byte */
    if (nl < hl) {                                               /* kseq.c:149: the rest of the line is the comment ... */
        cl = hl - nl - 1;
        if ((r->flags & 4) && cl == 0) cl = -1;                  /* ... unless the stream ends right behind the delimiter */
        else if (cl > 1 && h[hl - 1] == '\r') --cl;              /* kseq.c:106 */
    }
    if (cl >= 0) *buffered = 1;
    if (fastq && (r->flags & 3) == 1 && q_ptr) {                 /* a FASTQ record whose quality string was read */
        PyObject *q = cstr_text(q_ptr, r->seq_len);
        if (!q) return NULL;
        Py_SETREF(*last_qual, q);
    }
    name = cstr_text(h, nl);
    s = cstr_text(s_ptr, r->seq_len);
    if (with_comment == 2) {
        if (cl > 0 && name) {
            PyObject *c = cstr_text(h + nl + 1, cl), *j = c ? PyUnicode_FromFormat("%U %U", name, c) : NULL;
            Py_XDECREF(c);
            Py_SETREF(name, j);
        }
        with_comment = 0;
    }
    if (with_comment) {
        if (cl >= 0) com = PyUnicode_DecodeUTF8((const char *)h + nl + 1, (Py_ssize_t)cl, "surrogateescape");
        else if (*buffered) com = PyUnicode_FromStringAndSize("", 0);
        else com = Py_NewRef(Py_None);
    }
    if (!name || !s || (with_comment && !com)) { Py_XDECREF(name); Py_XDECREF(s); Py_XDECREF(com); return NULL; }
    t = PyTuple_New((fastq ? 3 : 2) + (with_comment ? 1 : 0));
    if (!t) { Py_DECREF(name); Py_DECREF(s); Py_XDECREF(com); return NULL; }
    {
        Py_ssize_t k = 0;
        PyTuple_SET_ITEM(t, k++, name);                          /* (the references move into the tuple) */
        PyTuple_SET_ITEM(t, k++, s);
        if (fastq) PyTuple_SET_ITEM(t, k++, Py_NewRef(*last_qual));
        if (with_comment) PyTuple_SET_ITEM(t, k++, com);
    }
    /* a tuple of str / None can be part of no cycle: out of the collector's lists at once (CPython does the same for such
     * tuples, but only when a collection first meets them -- two million of them are met by several) */
    PyObject_GC_UnTrack(t);
    return t;
}

static PyObject *mod_fastx_batch(PyObject *m, PyObject *args)
{
    Py_buffer hdr, ho, seq, qual, recs;
    PyObject *qual_obj, *state, *out = NULL, *last_qual;
    int fastq = 0, with_comment = 0, buffered, has_qual;
    Py_ssize_t k, i;
    (void)m;
    if (!PyArg_ParseTuple(args, "y*y*y*Oy*piO!", &hdr, &ho, &seq, &qual_obj, &recs, &fastq, &with_comment, &PyList_Type, &state)) return NULL;
    has_qual = qual_obj != Py_None;
    if (has_qual && PyObject_GetBuffer(qual_obj, &qual, PyBUF_SIMPLE) < 0) { has_qual = 0; goto done; }
    k = recs.len / (Py_ssize_t)sizeof(kq_rec);
    if (PyList_GET_SIZE(state) != 2 || ho.len < (k + 1) * 8) { PyErr_SetString(PyExc_ValueError, "bad argument"); goto done; }
    buffered = PyObject_IsTrue(PyList_GET_ITEM(state, 0));
    last_qual = Py_NewRef(PyList_GET_ITEM(state, 1));
    out = PyList_New(k);
    if (out) {
        const kq_rec *r = (const kq_rec *)recs.buf;
        const int64_t *o = (const int64_t *)ho.buf;
        const uint8_t *hb = (const uint8_t *)hdr.buf, *sb = (const uint8_t *)seq.buf, *qb = has_qual ? (const uint8_t *)qual.buf : NULL;
        const int64_t base = k ? r[0].seq_cum : 0;
        for (i = 0; i < k; ++i) {
            const int64_t so = r[i].seq_cum - base;
            PyObject *t;
            if (o[i + 1] > hdr.len || so + r[i].seq_len > seq.len) { PyErr_SetString(PyExc_ValueError, "batch buffers too short"); Py_CLEAR(out); break; }
            t = kq_tuple(hb + o[i], o[i + 1] - o[i], sb + so, qb ? qb + so : NULL, &r[i], fastq, with_comment, &buffered, &last_qual);
            if (!t) { Py_CLEAR(out); break; }
            PyList_SET_ITEM(out, i, t);
        }
    }
    if (out) {
        PyList_SetItem(state, 0, PyBool_FromLong(buffered));
        PyList_SetItem(state, 1, last_qual);                     /* steals the reference */
    } else Py_DECREF(last_qual);
done:
    PyBuffer_Release(&hdr); PyBuffer_Release(&ho); PyBuffer_Release(&seq); PyBuffer_Release(&recs);
    if (has_qual) PyBuffer_Release(&qual);
    return out;
}

/* The iterator behind Fastx.__iter__ (pyfastx_fastx_next, fastx.c:124-130): one tuple per call out of the current batch;
 * next_batch() -- a Python callable -- brings the next one as (hdr, hdr_off, seq, qual | None, recs), or None at the end. */
typedef struct {
    PyObject_HEAD
    PyObject *next_batch, *last_qual;
    Py_buffer hdr, ho, seq, qual, recs;
    int held, has_qual, fastq, with_comment, buffered;
    Py_ssize_t i, k;
    int64_t base;
} FastxIter;

static void fxi_release(FastxIter *it)
{
    if (it->held) {
        PyBuffer_Release(&it->hdr); PyBuffer_Release(&it->ho); PyBuffer_Release(&it->seq); PyBuffer_Release(&it->recs);
        if (it->has_qual) PyBuffer_Release(&it->qual);
        it->held = 0; it->has_qual = 0;
    }
    it->i = it->k = 0;
}
static void fxi_dealloc(FastxIter *it)
{
    fxi_release(it);
    Py_XDECREF(it->next_batch); Py_XDECREF(it->last_qual);
    Py_TYPE(it)->tp_free((PyObject *)it);
}
static PyObject *fxi_new(PyTypeObject *type, PyObject *args, PyObject *kw)
{
    PyObject *fn;
    int fastq = 0, with_comment = 0;
    FastxIter *it;
    (void)kw;
    if (!PyArg_ParseTuple(args, "Opi", &fn, &fastq, &with_comment)) return NULL;
    if (!PyCallable_Check(fn)) { PyErr_SetString(PyExc_TypeError, "next_batch must be callable"); return NULL; }
    it = (FastxIter *)type->tp_alloc(type, 0);
    if (!it) return NULL;
    it->next_batch = Py_NewRef(fn); it->last_qual = Py_NewRef(Py_None);
    it->held = it->has_qual = it->buffered = 0; it->fastq = fastq; it->with_comment = with_comment;
    it->i = it->k = 0; it->base = 0;
    return (PyObject *)it;
}
static PyObject *fxi_next(FastxIter *it)
{
    const kq_rec *r;
    const int64_t *o;
    int64_t so;
    while (it->i >= it->k) {                                     /* the next batch that holds records */
        PyObject *b, *q;
        fxi_release(it);
        if (!it->next_batch) return NULL;
        b = PyObject_CallNoArgs(it->next_batch);
        if (!b) return NULL;
        if (b == Py_None) { Py_DECREF(b); Py_CLEAR(it->next_batch); return NULL; }     /* StopIteration */
        if (!PyTuple_Check(b) || PyTuple_GET_SIZE(b) != 5) { Py_DECREF(b); PyErr_SetString(PyExc_TypeError, "next_batch() must return a 5-tuple or None"); return NULL; }
        q = PyTuple_GET_ITEM(b, 3);
        if (PyObject_GetBuffer(PyTuple_GET_ITEM(b, 0), &it->hdr, PyBUF_SIMPLE) < 0) { Py_DECREF(b); return NULL; }
        if (PyObject_GetBuffer(PyTuple_GET_ITEM(b, 1), &it->ho, PyBUF_SIMPLE) < 0) { PyBuffer_Release(&it->hdr); Py_DECREF(b); return NULL; }
        if (PyObject_GetBuffer(PyTuple_GET_ITEM(b, 2), &it->seq, PyBUF_SIMPLE) < 0) { PyBuffer_Release(&it->hdr); PyBuffer_Release(&it->ho); Py_DECREF(b); return NULL; }
        if (PyObject_GetBuffer(PyTuple_GET_ITEM(b, 4), &it->recs, PyBUF_SIMPLE) < 0) { PyBuffer_Release(&it->hdr); PyBuffer_Release(&it->ho); PyBuffer_Release(&it->seq); Py_DECREF(b); return NULL; }
        it->held = 1;
        if (q != Py_None) {
            if (PyObject_GetBuffer(q, &it->qual, PyBUF_SIMPLE) < 0) { fxi_release(it); Py_DECREF(b); return NULL; }
            it->has_qual = 1;
        }
        Py_DECREF(b);                                            /* the buffers keep their exporters alive */
        it->k = it->recs.len / (Py_ssize_t)sizeof(kq_rec);
        if (it->ho.len < (it->k + 1) * 8) { fxi_release(it); PyErr_SetString(PyExc_ValueError, "bad batch"); return NULL; }
        it->base = it->k ? ((const kq_rec *)it->recs.buf)[0].seq_cum : 0;
    }
    r = (const kq_rec *)it->recs.buf + it->i;
    o = (const int64_t *)it->ho.buf + it->i;
    so = r->seq_cum - it->base;
    if (o[1] > it->hdr.len || so + r->seq_len > it->seq.len) { PyErr_SetString(PyExc_ValueError, "batch buffers too short"); return NULL; }
    ++it->i;
    return kq_tuple((const uint8_t *)it->hdr.buf + o[0], o[1] - o[0], (const uint8_t *)it->seq.buf + so,
                    it->has_qual ? (const uint8_t *)it->qual.buf + so : NULL, r, it->fastq, it->with_comment, &it->buffered, &it->last_qual);
}
static PyTypeObject FastxIterType = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "pyfastx_amd._fxobj.FastxIter",
    .tp_basicsize = sizeof(FastxIter),
    .tp_dealloc = (destructor)fxi_dealloc,
    .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_iter = PyObject_SelfIter,
    .tp_iternext = (iternextfunc)fxi_next,
    .tp_new = fxi_new,
};

/* ------------------------------------------------------------------ RowCursor: a table of the index file stepped from C
 * The reference iterates an indexed file with sqlite3_step + sqlite3_column_* per record (fastq.c:566-596, index.c:525-560);
 * through Python's sqlite3 module every row becomes a tuple of Python objects first (0.5-0.9 us per row).  This cursor opens
 * its own READ-ONLY connection to the index file through the SAME library the sqlite3 module has loaded (dlopen: no build
 * dependency) and hands out a batch of rows as one list of names + one block of int64 columns.
 *   RowCursor(path, sql)   sql: first column an integer, second a TEXT, the others integers
 *   .fetch(n) -> None at the end, else (k, names, cols): cols = bytes of (ncol - 1) x k int64, column after column  */
typedef struct sqlite3 sqlite3;
typedef struct sqlite3_stmt sqlite3_stmt;
static struct {
    int state;                                                   /* 0 not tried, 1 loaded, -1 not there */
    int (*open_v2)(const char *, sqlite3 **, int, const char *);
    int (*close_v2)(sqlite3 *);
    int (*prepare_v2)(sqlite3 *, const char *, int, sqlite3_stmt **, const char **);
    int (*step)(sqlite3_stmt *);
    int (*finalize)(sqlite3_stmt *);
    int (*column_count)(sqlite3_stmt *);
    long long (*column_int64)(sqlite3_stmt *, int);
    const unsigned char *(*column_text)(sqlite3_stmt *, int);
    int (*column_bytes)(sqlite3_stmt *, int);
    const char *(*errmsg)(sqlite3 *);
    int (*busy_timeout)(sqlite3 *, int);
    int (*bind_int64)(sqlite3_stmt *, int, long long);
    int (*bind_text)(sqlite3_stmt *, int, const char *, int, void (*)(void *));
    int (*reset)(sqlite3_stmt *);
} SQ;
static int sq_load(void)
{
    void *h;
    if (SQ.state) return SQ.state > 0;
    SQ.state = -1;
    /* ONLY the copy the sqlite3 module of this process already uses (RTLD_NOLOAD): a second copy of SQLite on the same index
     * file would drop the first one's POSIX locks when it closes its descriptor (SQLite "how to corrupt", 2.2).  An
     * interpreter whose _sqlite3 carries SQLite inside itself has no such library: the callers then keep to the module's rows. */
    h = dlopen("libsqlite3.so.0", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (!h) h = dlopen("libsqlite3.so", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (!h) return 0;
#define SQ_SYM(field, name) do { *(void **)&SQ.field = dlsym(h, name); if (!SQ.field) return 0; } while (0)
    SQ_SYM(open_v2, "sqlite3_open_v2"); SQ_SYM(close_v2, "sqlite3_close_v2"); SQ_SYM(prepare_v2, "sqlite3_prepare_v2");
    SQ_SYM(step, "sqlite3_step"); SQ_SYM(finalize, "sqlite3_finalize"); SQ_SYM(column_count, "sqlite3_column_count");
    SQ_SYM(column_int64, "sqlite3_column_int64"); SQ_SYM(column_text, "sqlite3_column_text"); SQ_SYM(column_bytes, "sqlite3_column_bytes");
    SQ_SYM(errmsg, "sqlite3_errmsg"); SQ_SYM(busy_timeout, "sqlite3_busy_timeout");
    SQ_SYM(bind_int64, "sqlite3_bind_int64"); SQ_SYM(bind_text, "sqlite3_bind_text"); SQ_SYM(reset, "sqlite3_reset");
#undef SQ_SYM
    SQ.state = 1;
    return 1;
}
/* ------------------------------------------------------------------ FastqCore: the C base of pyfastx_amd.Fastq
 * pyfastx_fastq_subscript (fastq.c:521-545) is C in the reference: a prepared statement stepped once (fastq.c:454-519) and a
 * Read filled from its columns.  Here the same, on a read-only connection of its own to the index file (through the SQLite
 * library the sqlite3 module has loaded): fq[i] / fq[name] make the Read without entering the interpreter; the Read's
 * .seq / .qual / .quali of a PLAIN file come from the page cache (pread, read.c:37-45, 152-167, 237-278), of a gzip file
 * from the resident kernel through the Python methods as before.  Anything the C side cannot do (an index in memory, a
 * library it cannot bind, keys of other types) goes to the subclass's _getitem_slow. */
typedef struct {
    PyObject_HEAD
    sqlite3 *db;
    sqlite3_stmt *by_id, *by_name;
    unsigned long long handle;                   /* fx_handle* of the staged stream, 0 until it is staged */
    int fd;                                      /* the plain file (-1: none) */
    int phred;                                   /* Fastq(phred=) or meta.phred; 0: 33 (read.c:268) */
    long long counts;
    /* the read table this process built (fx_fastq_table: the arrays the index file was written from), kept by _core_table():
     * fq[i] then needs no statement at all -- the row is six array elements, the name is read from the file when asked for */
    Py_buffer tab[6];                            /* name_off i64, name_len i32, dlen i32, rlen i64, soff i64, qoff i64 */
    long long tab_n;                             /* rows of the table (0: none) */
    const int64_t *c_name_off, *c_rlen, *c_soff, *c_qoff;     /* the columns, wherever they live (c_name_off == NULL: names by statement) */
    const int32_t *c_name_len, *c_dlen;
    /* an object that LOADED its index file: once fq[i] has been asked for often enough to pay for it (hits * 22 > reads: a
     * statement costs ~4 us more than an array element, stepping through the table ~0.1 us per row), the four integer columns
     * are read from the file in one pass into arrays of its own; tab_cap = 0: never (set from FX_FQ_HOST_TABLE by the subclass) */
    void *own[4];
    long long tab_cap, int_hits;
    int tab_tried;
    /* the names of the reads as the build packed them for the index file (one buffer + n + 1 offsets), kept by _core_names():
     * fq[name] is then a hash look-up into the host table (the table of ids is made at the first such subscript) */
    Py_buffer nm_buf, nm_off;
    const unsigned char *nm_bytes;               /* the names and their n + 1 offsets, wherever they live: the two buffers above, or ... */
    const int64_t *nm_offs;
    void *own_names, *own_offs;                  /* ... arrays of the object's own, read from a LOADED index file in one pass (fqc_load_names) */
    int nm_tried;
    long long nm_n, nm_hits;                     /* the id table costs ~0.06 us per read to make and saves ~5 us per look-up: made once hits * 90 > reads */
    uint32_t *nm_ht;                             /* open addressing, id + 1 (0: empty); the lowest id of equal names wins */
    uint64_t nm_mask;
} FastqCore;
static PyTypeObject FastqCoreType;
static PyTypeObject *g_read_type = NULL;         /* api.Read (subclass of ReadCore) */

/* ------------------------------------------------------------------ Read: the C base of pyfastx_amd.Read (read.c:288-323)
 * The fields of a row of the `read` table and, for objects that come out of Fastq's iterator, the sequence and quality
 * strings that came with the iterator's batch.  read_batch() makes the objects of a whole batch in one call. */
typedef struct {
    PyObject_HEAD
    PyObject *fq, *name, *pre_seq, *pre_qual;
    long long id, desc_len, read_len, soff, qoff;
    long long name_off;                          /* lazy_name: where the name stands in the file ... */
    int name_len, lazy_name;                     /* ... and its bytes; name == NULL until somebody asks */
} ReadCore;

/* A Read refers to its Fastq, its name and two strings: none of them can lead back to it, so it has no business in the cyclic
 * collector's lists -- but api.Read is a heap type, CPython makes those GC types, and every allocation of a tracked object
 * counts towards the next collection: 16 384 reads per batch were 23 young collections and, with a large process around
 * them (torch: half a million tracked objects), now and then a full one.  Untracked right after the allocation, as tuples
 * of atoms are. */
static void read_untrack(ReadCore *r)
{
    /* (a subclass WITH an instance dict or weak references can be part of a cycle -- r.x = r -- and stays with the collector) */
    if (Py_TYPE(r)->tp_dictoffset != 0 || Py_TYPE(r)->tp_weaklistoffset != 0) return;
    if (PyType_HasFeature(Py_TYPE(r), Py_TPFLAGS_HAVE_GC) && PyObject_GC_IsTracked((PyObject *)r)) PyObject_GC_UnTrack((PyObject *)r);
}
static void read_dealloc(ReadCore *r)
{
    Py_XDECREF(r->fq); Py_XDECREF(r->name); Py_XDECREF(r->pre_seq); Py_XDECREF(r->pre_qual);
    Py_TYPE(r)->tp_free((PyObject *)r);
}
static int read_init(ReadCore *r, PyObject *args, PyObject *kw)
{
    PyObject *fq, *name;
    long long id, dlen, rlen, soff, qoff;
    (void)kw;
    if (!PyArg_ParseTuple(args, "OLOLLLL", &fq, &id, &name, &dlen, &rlen, &soff, &qoff)) return -1;
    Py_XSETREF(r->fq, Py_NewRef(fq));
    Py_XSETREF(r->name, Py_NewRef(name));
    r->id = id; r->desc_len = dlen; r->read_len = rlen; r->soff = soff; r->qoff = qoff;
    return 0;
}
static PyMemberDef read_members[] = {
    {"_fq", T_OBJECT_EX, offsetof(ReadCore, fq), READONLY, "the Fastq object"},
    {"_name_off", T_LONGLONG, offsetof(ReadCore, name_off), READONLY, NULL},
    {"_name_len", T_INT, offsetof(ReadCore, name_len), READONLY, NULL},
    {"id", T_LONGLONG, offsetof(ReadCore, id), 0, "1-based id"},
    {"_desc_len", T_LONGLONG, offsetof(ReadCore, desc_len), READONLY, NULL},
    {"_read_len", T_LONGLONG, offsetof(ReadCore, read_len), READONLY, NULL},
    {"_soff", T_LONGLONG, offsetof(ReadCore, soff), READONLY, NULL},
    {"_qoff", T_LONGLONG, offsetof(ReadCore, qoff), READONLY, NULL},
    {"_pre_seq", T_OBJECT, offsetof(ReadCore, pre_seq), READONLY, "the sequence, when it came with the iterator's batch (else None)"},
    {"_pre_qual", T_OBJECT, offsetof(ReadCore, pre_qual), READONLY, "the quality string, likewise"},
    {NULL, 0, 0, 0, NULL}};
/* read_len bytes at `off` of the plain file behind the read's Fastq into buf (FX_GETTER_CAP bytes) -> 1; 0: not a case
 * for the host path (gzip input, stream not staged, a long read, a short file) */
static int read_host_bytes(ReadCore *r, long long off, uint8_t *buf)
{
    FastqCore *fq;
    if (!r->fq || !PyObject_TypeCheck(r->fq, &FastqCoreType)) return 0;
    fq = (FastqCore *)r->fq;
    if (fq->fd < 0 || !fq->handle || r->read_len <= 0 || r->read_len > FX_GETTER_CAP) return 0;
    return host_read(fq->fd, buf, (Py_ssize_t)r->read_len, off) == (Py_ssize_t)r->read_len;
}
/* .seq / .qual: the string that came with the iterator's batch, else the page cache (a plain file), else the subclass's
 * _seq_slow() / _qual_slow() (one fetch through the resident kernel) */
static PyObject *read_get_seq(ReadCore *r, void *c)
{
    uint8_t buf[FX_GETTER_CAP];
    (void)c;
    if (r->pre_seq) return Py_NewRef(r->pre_seq);
    if (read_host_bytes(r, r->soff, buf)) return PyUnicode_DecodeLatin1((const char *)buf, (Py_ssize_t)r->read_len, NULL);      /* read.c:152-167 */
    return PyObject_CallMethod((PyObject *)r, "_seq_slow", NULL);
}
static PyObject *read_get_qual(ReadCore *r, void *c)
{
    uint8_t buf[FX_GETTER_CAP];
    (void)c;
    if (r->pre_qual) return Py_NewRef(r->pre_qual);
    if (read_host_bytes(r, r->qoff, buf)) return PyUnicode_DecodeLatin1((const char *)buf, (Py_ssize_t)r->read_len, NULL);      /* read.c:237-249 */
    return PyObject_CallMethod((PyObject *)r, "_qual_slow", NULL);
}
/* .quali: qual - phred, a list of ints (read.c:251-278; the bytes are C chars there: signed) */
static PyObject *read_get_quali(ReadCore *r, void *c)
{
    uint8_t buf[FX_GETTER_CAP];
    (void)c;
    if (read_host_bytes(r, r->qoff, buf)) {
        const int phred = ((FastqCore *)r->fq)->phred ? ((FastqCore *)r->fq)->phred : 33;                 /* read.c:268 */
        PyObject *l = PyList_New((Py_ssize_t)r->read_len);
        Py_ssize_t i;
        for (i = 0; l && i < (Py_ssize_t)r->read_len; ++i) {
            PyObject *v = PyLong_FromLong((long)(signed char)buf[i] - phred);
            if (!v) { Py_CLEAR(l); break; }
            PyList_SET_ITEM(l, i, v);
        }
        return l;
    }
    return PyObject_CallMethod((PyObject *)r, "_quali_slow", NULL);
}
/* .name: the str the row came with; for a read made from the host table (fq[i]) the bytes name_off .. + name_len of the file,
 * read when first asked for (fastq.c:112-117 cut them out of the header line; the index file stores them verbatim) */
static PyObject *read_get_name(ReadCore *r, void *c)
{
    (void)c;
    if (!r->name && r->lazy_name == 2) {                          /* a row of the table read from the index file: its name is there */
        FastqCore *fq = r->fq && PyObject_TypeCheck(r->fq, &FastqCoreType) ? (FastqCore *)r->fq : NULL;
        if (fq && fq->by_id) {
            SQ.bind_int64(fq->by_id, 1, r->id);
            if (SQ.step(fq->by_id) == 100) {
                const unsigned char *t = SQ.column_text(fq->by_id, 1);
                r->name = PyUnicode_DecodeUTF8(t ? (const char *)t : "", t ? SQ.column_bytes(fq->by_id, 1) : 0, "surrogateescape");
            }
            SQ.reset(fq->by_id);
        }
        if (!r->name) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_RuntimeError, "the index file no longer answers for this read"); return NULL; }
    }
    if (!r->name && r->lazy_name) {
        uint8_t buf[FX_GETTER_CAP];
        FastqCore *fq = r->fq && PyObject_TypeCheck(r->fq, &FastqCoreType) ? (FastqCore *)r->fq : NULL;
        if (fq && fq->fd >= 0 && r->name_len >= 0 && r->name_len <= FX_GETTER_CAP &&
            host_read(fq->fd, buf, (Py_ssize_t)r->name_len, r->name_off) == (Py_ssize_t)r->name_len)
            r->name = PyUnicode_DecodeUTF8((const char *)buf, (Py_ssize_t)r->name_len, "surrogateescape");
        else r->name = PyObject_CallMethod((PyObject *)r, "_name_slow", NULL);        /* gzip input: through the resident stream */
        if (!r->name) return NULL;
    }
    if (!r->name) { PyErr_SetString(PyExc_AttributeError, "name"); return NULL; }
    return Py_NewRef(r->name);
}
static int read_set_name(ReadCore *r, PyObject *v, void *c)
{
    (void)c;
    if (!v) { PyErr_SetString(PyExc_AttributeError, "name cannot be deleted"); return -1; }
    Py_XSETREF(r->name, Py_NewRef(v));
    return 0;
}
static PyGetSetDef read_getset[] = {
    {"name", (getter)read_get_name, (setter)read_set_name, "read name", NULL},
    {"seq", (getter)read_get_seq, NULL, "read.c:152-167", NULL},
    {"qual", (getter)read_get_qual, NULL, "read.c:237-249", NULL},
    {"quali", (getter)read_get_quali, NULL, "read.c:251-278", NULL},
    {NULL, NULL, NULL, NULL, NULL}};
static Py_ssize_t read_length(ReadCore *r) { return (Py_ssize_t)r->read_len; }
static PySequenceMethods read_as_sequence = {.sq_length = (lenfunc)read_length};
static PyTypeObject ReadCoreType = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "pyfastx_amd._fxobj.ReadCore",
    .tp_getset = read_getset,
    .tp_as_sequence = &read_as_sequence,
    .tp_basicsize = sizeof(ReadCore),
    .tp_dealloc = (destructor)read_dealloc,
    .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_BASETYPE,
    .tp_members = read_members,
    .tp_init = (initproc)read_init,
    .tp_new = PyType_GenericNew,
};

/* ---- FastqCore methods */
static void fqc_close_db(FastqCore *f)
{
    if (f->by_id) { SQ.finalize(f->by_id); f->by_id = NULL; }
    if (f->by_name) { SQ.finalize(f->by_name); f->by_name = NULL; }
    if (f->db) { SQ.close_v2(f->db); f->db = NULL; }
}
static void fqc_drop_names(FastqCore *f)
{
    f->nm_n = 0; f->nm_hits = 0;
    f->nm_bytes = NULL; f->nm_offs = NULL;
    free(f->own_names); free(f->own_offs); f->own_names = f->own_offs = NULL;
    free(f->nm_ht); f->nm_ht = NULL; f->nm_mask = 0;
    if (f->nm_buf.obj) PyBuffer_Release(&f->nm_buf);
    if (f->nm_off.obj) PyBuffer_Release(&f->nm_off);
}
static uint64_t fq_name_hash(const unsigned char *p, Py_ssize_t l)
{
    uint64_t h = 0xcbf29ce484222325ull;                          /* FNV-1a, then a finishing mix (short keys that differ in their last digits) */
    Py_ssize_t i;
    for (i = 0; i < l; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
    h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
    return h;
}
/* the id table of the packed names (at the first fq[name]); 0: no memory, the statements go on answering */
static int fqc_build_ht(FastqCore *f)
{
    const unsigned char *nb = f->nm_bytes;
    const int64_t *no = f->nm_offs;
    uint64_t cap = 16;
    long long i;
    while (cap < (uint64_t)f->nm_n * 2) cap <<= 1;
    f->nm_ht = (uint32_t *)calloc((size_t)cap, 4);
    if (!f->nm_ht) return 0;
    f->nm_mask = cap - 1;
    for (i = 0; i < f->nm_n; ++i) {
        const Py_ssize_t l = (Py_ssize_t)(no[i + 1] - no[i]);
        uint64_t at = fq_name_hash(nb + no[i], l) & f->nm_mask;
        for (;;) {
            const uint32_t e = f->nm_ht[at];
            if (!e) { f->nm_ht[at] = (uint32_t)(i + 1); break; }
            if (no[e] - no[e - 1] == l && memcmp(nb + no[e - 1], nb + no[i], (size_t)l) == 0) break;   /* the same name again: the first one stays */
            at = (at + 1) & f->nm_mask;
        }
    }
    return 1;
}
/* id (0-based) of the read called t, -1: no such read */
static long long fqc_find_name(FastqCore *f, const char *t, Py_ssize_t l)
{
    const unsigned char *nb = f->nm_bytes;
    const int64_t *no = f->nm_offs;
    uint64_t at = fq_name_hash((const unsigned char *)t, l) & f->nm_mask;
    for (;;) {
        const uint32_t e = f->nm_ht[at];
        if (!e) return -1;
        if (no[e] - no[e - 1] == l && memcmp(nb + no[e - 1], t, (size_t)l) == 0) return (long long)e - 1;
        at = (at + 1) & f->nm_mask;
    }
}
static void fqc_drop_table(FastqCore *f)
{
    int k;
    f->tab_n = 0;
    f->c_name_off = f->c_rlen = f->c_soff = f->c_qoff = NULL; f->c_name_len = f->c_dlen = NULL;
    for (k = 0; k < 6; ++k) if (f->tab[k].obj) PyBuffer_Release(&f->tab[k]);
    for (k = 0; k < 4; ++k) { free(f->own[k]); f->own[k] = NULL; }
}
/* dlen, rlen, soff, qoff of every row of the index file into arrays of the object's own (IDs must run 1 .. counts) */
static void fqc_load_table(FastqCore *f)
{
    sqlite3_stmt *st = NULL;
    const long long n = f->counts;
    long long k = 0;
    int32_t *dl; int64_t *rl, *so, *qo;
    if (!f->db || n <= 0 || SQ.prepare_v2(f->db, "SELECT ID, dlen, rlen, soff, qoff FROM read ORDER BY ID", -1, &st, NULL) != 0) return;
    fqc_drop_table(f);
    f->own[0] = dl = (int32_t *)malloc((size_t)n * 4); f->own[1] = rl = (int64_t *)malloc((size_t)n * 8);
    f->own[2] = so = (int64_t *)malloc((size_t)n * 8); f->own[3] = qo = (int64_t *)malloc((size_t)n * 8);
    if (dl && rl && so && qo) {
        while (k < n && SQ.step(st) == 100 && SQ.column_int64(st, 0) == k + 1) {
            dl[k] = (int32_t)SQ.column_int64(st, 1); rl[k] = SQ.column_int64(st, 2); so[k] = SQ.column_int64(st, 3); qo[k] = SQ.column_int64(st, 4);
            ++k;
        }
    }
    SQ.finalize(st);
    if (k != n) { fqc_drop_table(f); return; }                   /* not the table it says it is: the statements go on answering */
    f->c_dlen = dl; f->c_rlen = rl; f->c_soff = so; f->c_qoff = qo;
    f->tab_n = n;
}
static void fqc_dealloc(FastqCore *f)
{
    fqc_drop_names(f);
    fqc_drop_table(f);
    fqc_close_db(f);
    if (f->fd >= 0) close(f->fd);
    Py_TYPE(f)->tp_free((PyObject *)f);
}
static PyObject *fqc_new(PyTypeObject *type, PyObject *args, PyObject *kw)
{
    FastqCore *f = (FastqCore *)type->tp_alloc(type, 0);
    (void)args; (void)kw;
    if (f) f->fd = -1;
    return (PyObject *)f;
}
/* _core_open(index file | None) -> True when fq[i] / fq[name] are served from C */
static PyObject *fqc_open(FastqCore *f, PyObject *arg)
{
    PyObject *b = NULL;
    int ok;
    if (f->db || f->by_id || f->by_name) fqc_close_db(f);
    fqc_drop_names(f);
    fqc_drop_table(f); f->tab_tried = 0; f->int_hits = 0; f->nm_tried = 0;   /* another index file: whatever was known of the old one goes */
    if (arg == Py_None || !sq_load()) Py_RETURN_FALSE;
    if (!PyUnicode_FSConverter(arg, &b)) return NULL;
    ok = SQ.open_v2(PyBytes_AS_STRING(b), &f->db, 1 /* SQLITE_OPEN_READONLY */, NULL) == 0 &&
         SQ.prepare_v2(f->db, "SELECT ID, name, dlen, rlen, soff, qoff FROM read WHERE ID=? LIMIT 1", -1, &f->by_id, NULL) == 0 &&      /* fastq.c:454-484 */
         SQ.prepare_v2(f->db, "SELECT ID, name, dlen, rlen, soff, qoff FROM read WHERE name=? LIMIT 1", -1, &f->by_name, NULL) == 0;   /* fastq.c:486-519 */
    Py_DECREF(b);
    if (!ok) { fqc_close_db(f); Py_RETURN_FALSE; }
    SQ.busy_timeout(f->db, 5000);
    Py_RETURN_TRUE;
}
static PyObject *fqc_stage(FastqCore *f, PyObject *args)
{
    unsigned long long h = 0;
    PyObject *path = Py_None;
    if (!PyArg_ParseTuple(args, "K|O", &h, &path)) return NULL;
    if (f->fd >= 0) { close(f->fd); f->fd = -1; }
    f->handle = h;
    if (h && path != Py_None) {
        PyObject *b = NULL;
        if (!PyUnicode_FSConverter(path, &b)) return NULL;
        f->fd = open(PyBytes_AS_STRING(b), O_RDONLY | O_CLOEXEC);
        Py_DECREF(b);
    }
    Py_RETURN_NONE;
}
/* An object that loaded its index file and is asked by name often enough: the integer columns (fqc_load_table) and the names of
 * every row, in one pass over the file each, into arrays of its own; fq[name] then is the same hash look-up as on the
 * object that built the index. */
static void fqc_load_names(FastqCore *f)
{
    sqlite3_stmt *st = NULL;
    const long long n = f->counts;
    long long k = 0;
    size_t cap = 0, used = 0;
    unsigned char *buf = NULL;
    int64_t *off = NULL;
    if (!f->tab_n) { f->tab_tried = 1; fqc_load_table(f); }
    if (f->tab_n != n || n <= 0 || n >= 0xFFFFFFFFll || !f->db) return;
    if (SQ.prepare_v2(f->db, "SELECT ID, name FROM read ORDER BY ID", -1, &st, NULL) != 0) return;
    off = (int64_t *)malloc((size_t)(n + 1) * 8);
    cap = (size_t)n * 24 + 64;
    buf = (unsigned char *)malloc(cap);
    if (off && buf) {
        off[0] = 0;
        while (k < n && SQ.step(st) == 100 && SQ.column_int64(st, 0) == k + 1) {
            const unsigned char *t = SQ.column_text(st, 1);
            const size_t l = t ? (size_t)SQ.column_bytes(st, 1) : 0;
            if (used + l > cap) {
                unsigned char *nb;
                cap = (used + l) * 2;
                nb = (unsigned char *)realloc(buf, cap);
                if (!nb) break;
                buf = nb;
            }
            if (l) memcpy(buf + used, t, l);
            used += l;
            off[++k] = (int64_t)used;
        }
    }
    SQ.finalize(st);
    if (k != n || !off || !buf) { free(off); free(buf); return; }
    fqc_drop_names(f);
    f->own_names = buf; f->own_offs = off;
    f->nm_bytes = buf; f->nm_offs = off; f->nm_n = n;
}
/* _core_table(name_off, name_len, dlen, rlen, soff, qoff) keeps the six columns (any objects with the buffer protocol; the
 * item sizes are checked, the row count is the shortest of them); _core_table() forgets them */
static PyObject *fqc_table(FastqCore *f, PyObject *args)
{
    static const Py_ssize_t width[6] = {8, 4, 4, 8, 8, 8};
    PyObject *o[6] = {NULL, NULL, NULL, NULL, NULL, NULL};
    long long n = -1;
    int k;
    if (!PyArg_ParseTuple(args, "|OOOOOO", &o[0], &o[1], &o[2], &o[3], &o[4], &o[5])) return NULL;
    fqc_drop_table(f);
    if (!o[0]) Py_RETURN_NONE;
    if (!o[5]) { PyErr_SetString(PyExc_TypeError, "_core_table(name_off, name_len, dlen, rlen, soff, qoff) or _core_table()"); return NULL; }
    for (k = 0; k < 6; ++k) {
        if (PyObject_GetBuffer(o[k], &f->tab[k], PyBUF_SIMPLE) != 0) { f->tab[k].obj = NULL; fqc_drop_table(f); return NULL; }
        if (f->tab[k].len % width[k]) { fqc_drop_table(f); PyErr_SetString(PyExc_ValueError, "_core_table: int64, int32, int32, int64, int64, int64 columns"); return NULL; }
        if (n < 0 || f->tab[k].len / width[k] < n) n = f->tab[k].len / width[k];
    }
    f->c_name_off = (const int64_t *)f->tab[0].buf; f->c_name_len = (const int32_t *)f->tab[1].buf; f->c_dlen = (const int32_t *)f->tab[2].buf;
    f->c_rlen = (const int64_t *)f->tab[3].buf; f->c_soff = (const int64_t *)f->tab[4].buf; f->c_qoff = (const int64_t *)f->tab[5].buf;
    f->tab_n = n;
    Py_RETURN_NONE;
}
/* _core_names(packed names, int64 offsets[n + 1]) keeps them for fq[name] (with the host table of _core_table); _core_names() forgets */
static PyObject *fqc_names(FastqCore *f, PyObject *args)
{
    PyObject *b = NULL, *o = NULL;
    long long n;
    const int64_t *no;
    if (!PyArg_ParseTuple(args, "|OO", &b, &o)) return NULL;
    fqc_drop_names(f);
    if (!b) Py_RETURN_NONE;
    if (!o) { PyErr_SetString(PyExc_TypeError, "_core_names(packed names, int64 offsets[n + 1]) or _core_names()"); return NULL; }
    if (PyObject_GetBuffer(b, &f->nm_buf, PyBUF_SIMPLE) != 0) { f->nm_buf.obj = NULL; return NULL; }
    if (PyObject_GetBuffer(o, &f->nm_off, PyBUF_SIMPLE) != 0) { f->nm_off.obj = NULL; fqc_drop_names(f); return NULL; }
    n = (long long)(f->nm_off.len / 8) - 1;
    no = (const int64_t *)f->nm_off.buf;
    if (f->nm_off.len % 8 || n < 1 || n >= 0xFFFFFFFFll || no[0] != 0 || no[n] > (int64_t)f->nm_buf.len) {
        fqc_drop_names(f);
        PyErr_SetString(PyExc_ValueError, "_core_names: offsets must be int64[n + 1] from 0 to at most the length of the names");
        return NULL;
    }
    {   /* rising offsets: checked here once, trusted by every look-up */
        long long i;
        for (i = 0; i < n; ++i) if (no[i + 1] < no[i]) { fqc_drop_names(f); PyErr_SetString(PyExc_ValueError, "_core_names: offsets must not fall"); return NULL; }
    }
    f->nm_bytes = (const unsigned char *)f->nm_buf.buf; f->nm_offs = no;
    f->nm_n = n;
    Py_RETURN_NONE;
}
/* the row the statement stands on -> a Read (name: the key itself when the caller asked by name) */
static PyObject *fqc_read_of_row(FastqCore *f, sqlite3_stmt *st, PyObject *name)
{
    ReadCore *r = (ReadCore *)g_read_type->tp_alloc(g_read_type, 0);
    if (!r) return NULL;
    read_untrack(r);
    r->fq = Py_NewRef((PyObject *)f);
    r->id = SQ.column_int64(st, 0);
    if (name) r->name = Py_NewRef(name);
    else {
        const unsigned char *t = SQ.column_text(st, 1);
        r->name = PyUnicode_DecodeUTF8(t ? (const char *)t : "", t ? SQ.column_bytes(st, 1) : 0, "surrogateescape");
    }
    r->desc_len = SQ.column_int64(st, 2); r->read_len = SQ.column_int64(st, 3);
    r->soff = SQ.column_int64(st, 4); r->qoff = SQ.column_int64(st, 5);
    if (!r->name) { Py_DECREF(r); return NULL; }
    return (PyObject *)r;
}
static PyObject *fqc_subscript(FastqCore *f, PyObject *key)
{
    if (f->by_id && g_read_type && PyLong_CheckExact(key)) {                     /* fastq.c:527-534 */
        int ovf = 0;
        long long i = PyLong_AsLongLongAndOverflow(key, &ovf);
        if (!ovf) {
            PyObject *r = NULL;
            int rc;
            if (i < 0) i += f->counts;
            if (i >= f->counts) { PyErr_SetString(PyExc_IndexError, "index out of range"); return NULL; }
            if (!f->tab_n && !f->tab_tried && f->counts <= f->tab_cap && ++f->int_hits >= 64 && f->int_hits * 22 > f->counts) {
                f->tab_tried = 1;                                                  /* asked often enough: the columns in one pass */
                fqc_load_table(f);
            }
            if (i >= 0 && i < f->tab_n) {                                          /* the row from the table on the host */
                ReadCore *rd = (ReadCore *)g_read_type->tp_alloc(g_read_type, 0);
                if (!rd) return NULL;
                read_untrack(rd);
                rd->fq = Py_NewRef((PyObject *)f);
                rd->id = i + 1;
                if (f->c_name_off) { rd->lazy_name = 1; rd->name_off = f->c_name_off[i]; rd->name_len = f->c_name_len[i]; }
                else rd->lazy_name = 2;                                            /* the name by statement, when asked for */
                rd->desc_len = f->c_dlen[i]; rd->read_len = f->c_rlen[i];
                rd->soff = f->c_soff[i]; rd->qoff = f->c_qoff[i];
                return (PyObject *)rd;
            }
            SQ.bind_int64(f->by_id, 1, i + 1);
            rc = SQ.step(f->by_id);
            if (rc == 100) r = fqc_read_of_row(f, f->by_id, NULL);
            else if (rc == 101) PyErr_SetString(PyExc_IndexError, "Index Error");
            SQ.reset(f->by_id);
            if (r || PyErr_Occurred()) return r;                                   /* (another code: the slow path reports it) */
        }
    } else if (f->by_name && g_read_type && PyUnicode_CheckExact(key)) {         /* fastq.c:535-541 */
        Py_ssize_t l = 0;
        const char *t = PyUnicode_AsUTF8AndSize(key, &l);
        if (t && !f->nm_n && !f->nm_tried && f->counts <= f->tab_cap && ++f->nm_hits >= 64 && f->nm_hits * 25 > f->counts) {
            f->nm_tried = 1;                                                       /* a loaded index file, asked by name often enough */
            fqc_load_names(f);
            f->nm_hits = 1ll << 40;                                                /* (the id table at once: the look-ups have paid for it) */
        }
        if (t && f->nm_n > 0 && f->nm_n == f->tab_n && (f->nm_ht || (++f->nm_hits >= 64 && f->nm_hits * 90 > f->nm_n && fqc_build_ht(f)))) {   /* the names this process packed: no statement */
            const long long i = fqc_find_name(f, t, l);
            ReadCore *rd;
            if (i < 0) { PyErr_Format(PyExc_KeyError, "%U does not exist in fastq file", key); return NULL; }
            rd = (ReadCore *)g_read_type->tp_alloc(g_read_type, 0);
            if (!rd) return NULL;
            read_untrack(rd);
            rd->fq = Py_NewRef((PyObject *)f);
            rd->id = i + 1;
            rd->name = Py_NewRef(key);
            rd->desc_len = f->c_dlen[i]; rd->read_len = f->c_rlen[i]; rd->soff = f->c_soff[i]; rd->qoff = f->c_qoff[i];
            if (f->c_name_off) { rd->name_off = f->c_name_off[i]; rd->name_len = f->c_name_len[i]; }
            return (PyObject *)rd;
        }
        if (t) {
            PyObject *r = NULL;
            int rc;
            SQ.bind_text(f->by_name, 1, t, (int)l, NULL /* SQLITE_STATIC: the key outlives the step */);
            rc = SQ.step(f->by_name);
            if (rc == 100) r = fqc_read_of_row(f, f->by_name, key);
            else if (rc == 101) PyErr_Format(PyExc_KeyError, "%U does not exist in fastq file", key);
            SQ.reset(f->by_name);
            if (r || PyErr_Occurred()) return r;
        } else PyErr_Clear();                                                      /* lone surrogates: the slow path encodes them its way */
    }
    return PyObject_CallMethod((PyObject *)f, "_getitem_slow", "O", key);
}
static PyMethodDef fqc_methods[] = {
    {"_core_open", (PyCFunction)fqc_open, METH_O, "_core_open(index file | None) -> bool"},
    {"_core_stage", (PyCFunction)fqc_stage, METH_VARARGS, "_core_stage(handle, plain path | None)"},
    {"_core_table", (PyCFunction)fqc_table, METH_VARARGS, "_core_table(name_off, name_len, dlen, rlen, soff, qoff) | _core_table()"},
    {"_core_names", (PyCFunction)fqc_names, METH_VARARGS, "_core_names(packed names, int64 offsets[n + 1]) | _core_names()"},
    {NULL, NULL, 0, NULL}};
static PyMemberDef fqc_members[] = {
    {"_counts", T_LONGLONG, offsetof(FastqCore, counts), 0, "reads in the index"},
    {"_phred", T_INT, offsetof(FastqCore, phred), 0, "quality offset (0: 33)"},
    {"_core_handle", T_ULONGLONG, offsetof(FastqCore, handle), READONLY, NULL},
    {"_core_fd", T_INT, offsetof(FastqCore, fd), READONLY, NULL},
    {"_core_table_rows", T_LONGLONG, offsetof(FastqCore, tab_n), READONLY, "rows of the host table fq[i] is served from (0: the index file)"},
    {"_core_names_rows", T_LONGLONG, offsetof(FastqCore, nm_n), READONLY, "names kept on the host for fq[name] (0: the index file's statement)"},
    {"_core_table_cap", T_LONGLONG, offsetof(FastqCore, tab_cap), 0, "an object that loaded its index reads the table from it once fq[i] is used enough, up to this many reads (0: never)"},
    {NULL, 0, 0, 0, NULL}};
static PyMappingMethods fqc_mapping = {NULL, (binaryfunc)fqc_subscript, NULL};      /* (__len__ stays with the Python class) */
static PyTypeObject FastqCoreType = {
    PyVarObject_HEAD_INIT(NULL, 0).tp_name = "pyfastx_amd._fxobj.FastqCore", .tp_basicsize = sizeof(FastqCore),
    .tp_dealloc = (destructor)fqc_dealloc, .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_BASETYPE, .tp_new = fqc_new,
    .tp_members = fqc_members, .tp_methods = fqc_methods, .tp_as_mapping = &fqc_mapping,
    .tp_doc = "prepared statements and subscript fast path of pyfastx_amd.Fastq"};
static PyObject *mod_set_read_type(PyObject *m, PyObject *t)
{
    (void)m;
    if (!PyType_Check(t) || !PyType_IsSubtype((PyTypeObject *)t, &ReadCoreType)) { PyErr_SetString(PyExc_TypeError, "the read type must derive from ReadCore"); return NULL; }
    Py_XDECREF(g_read_type);
    g_read_type = (PyTypeObject *)Py_NewRef(t);
    Py_RETURN_NONE;
}

/* read_batch(ReadType, fq, rows, seq, qual, offs) -> list: rows = the (ID, name, dlen, rlen, soff, qoff) tuples of the batch,
 * seq / qual = the bytes of their sequence / quality lines one behind the other, offs = int64[k + 1] */
static PyObject *mod_read_batch(PyObject *m, PyObject *args)
{
    PyObject *type, *fq, *rows, *out = NULL;
    Py_buffer seq, qual, offs;
    Py_ssize_t k, i;
    (void)m;
    if (!PyArg_ParseTuple(args, "OOO!y*y*y*", &type, &fq, &PyList_Type, &rows, &seq, &qual, &offs)) return NULL;
    k = PyList_GET_SIZE(rows);
    if (!PyType_Check(type) || !PyType_IsSubtype((PyTypeObject *)type, &ReadCoreType) || offs.len < (k + 1) * 8) {
        PyErr_SetString(PyExc_TypeError, "read_batch(ReadCore subtype, fq, rows, seq, qual, int64 offsets[k + 1])");
        goto done;
    }
    out = PyList_New(k);
    for (i = 0; out && i < k; ++i) {
        PyObject *row = PyList_GET_ITEM(rows, i);
        const int64_t *o = (const int64_t *)offs.buf;
        ReadCore *r;
        if (!PyTuple_Check(row) || PyTuple_GET_SIZE(row) < 6 || o[i] < 0 || o[i + 1] < o[i] || o[i + 1] > seq.len || o[i + 1] > qual.len) {
            PyErr_SetString(PyExc_ValueError, "bad row or offsets");
            Py_CLEAR(out);
            break;
        }
        r = (ReadCore *)((PyTypeObject *)type)->tp_alloc((PyTypeObject *)type, 0);
        if (!r) { Py_CLEAR(out); break; }
        read_untrack(r);
        PyList_SET_ITEM(out, i, (PyObject *)r);
        r->fq = Py_NewRef(fq);
        {   /* the name: a str, or the column's bytes (SELECT CAST(name AS BLOB): no text_factory call per row) decoded as fxi.connect does */
            PyObject *nm = PyTuple_GET_ITEM(row, 1);
            r->name = PyBytes_Check(nm) ? PyUnicode_DecodeUTF8(PyBytes_AS_STRING(nm), PyBytes_GET_SIZE(nm), "surrogateescape") : Py_NewRef(nm);
        }
        r->id = PyLong_AsLongLong(PyTuple_GET_ITEM(row, 0));
        r->desc_len = PyLong_AsLongLong(PyTuple_GET_ITEM(row, 2));
        r->read_len = PyLong_AsLongLong(PyTuple_GET_ITEM(row, 3));
        r->soff = PyLong_AsLongLong(PyTuple_GET_ITEM(row, 4));
        r->qoff = PyLong_AsLongLong(PyTuple_GET_ITEM(row, 5));
        r->pre_seq = PyUnicode_DecodeLatin1((const char *)seq.buf + o[i], (Py_ssize_t)(o[i + 1] - o[i]), NULL);
        r->pre_qual = PyUnicode_DecodeLatin1((const char *)qual.buf + o[i], (Py_ssize_t)(o[i + 1] - o[i]), NULL);
        if (PyErr_Occurred() || !r->name || !r->pre_seq || !r->pre_qual) { Py_CLEAR(out); break; }
    }
done:
    PyBuffer_Release(&seq); PyBuffer_Release(&qual); PyBuffer_Release(&offs);
    return out;
}

typedef struct { PyObject_HEAD sqlite3 *db; sqlite3_stmt *st; int ncol, done; } RowCursor;
static void rc_close(RowCursor *c)
{
    if (c->st) { SQ.finalize(c->st); c->st = NULL; }
    if (c->db) { SQ.close_v2(c->db); c->db = NULL; }
}
static void rc_dealloc(RowCursor *c) { rc_close(c); Py_TYPE(c)->tp_free((PyObject *)c); }
static PyObject *rc_new(PyTypeObject *type, PyObject *args, PyObject *kw)
{
    const char *path, *sql;
    RowCursor *c;
    (void)kw;
    if (!PyArg_ParseTuple(args, "ss", &path, &sql)) return NULL;
    if (!sq_load()) { PyErr_SetString(PyExc_RuntimeError, "libsqlite3 could not be loaded"); return NULL; }
    c = (RowCursor *)type->tp_alloc(type, 0);
    if (!c) return NULL;
    c->db = NULL; c->st = NULL; c->done = 0;
    if (SQ.open_v2(path, &c->db, 1 /* SQLITE_OPEN_READONLY */, NULL) != 0 || SQ.prepare_v2(c->db, sql, -1, &c->st, NULL) != 0) {
        PyErr_Format(PyExc_RuntimeError, "RowCursor: %s", c->db ? SQ.errmsg(c->db) : "cannot open the index file");
        rc_close(c);
        Py_DECREF(c);
        return NULL;
    }
    SQ.busy_timeout(c->db, 5000);                                /* a writer in another process: wait, do not fail the iteration */
    c->ncol = SQ.column_count(c->st);
    if (c->ncol < 2) { PyErr_SetString(PyExc_ValueError, "RowCursor: the statement must yield an integer, a text and integers"); rc_close(c); Py_DECREF(c); return NULL; }
    return (PyObject *)c;
}
static PyObject *rc_fetch(RowCursor *c, PyObject *arg)
{
    Py_ssize_t n = PyLong_AsSsize_t(arg), k = 0;
    const int ni = c->ncol - 1;
    PyObject *names, *cols, *out;
    long long *v;
    int j;
    if (n <= 0) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "fetch(n): n > 0"); return NULL; }
    if (c->done || !c->st) Py_RETURN_NONE;
    names = PyList_New(n);
    cols = PyBytes_FromStringAndSize(NULL, (Py_ssize_t)ni * n * 8);
    if (!names || !cols) { Py_XDECREF(names); Py_XDECREF(cols); return NULL; }
    v = (long long *)PyBytes_AS_STRING(cols);
    while (k < n) {
        const int rc = SQ.step(c->st);
        PyObject *nm;
        if (rc == 101 /* SQLITE_DONE */) { c->done = 1; break; }
        if (rc != 100 /* SQLITE_ROW */) {
            PyErr_Format(PyExc_RuntimeError, "RowCursor: %s (code %d)", SQ.errmsg(c->db), rc);
            Py_DECREF(names); Py_DECREF(cols);
            return NULL;
        }
        v[k] = SQ.column_int64(c->st, 0);
        for (j = 2; j < c->ncol; ++j) v[(Py_ssize_t)(j - 1) * n + k] = SQ.column_int64(c->st, j);
        {
            const unsigned char *t = SQ.column_text(c->st, 1);
            nm = PyUnicode_DecodeUTF8(t ? (const char *)t : "", t ? SQ.column_bytes(c->st, 1) : 0, "surrogateescape");   /* as fxi.connect's text_factory */
        }
        if (!nm) { Py_DECREF(names); Py_DECREF(cols); return NULL; }
        PyList_SET_ITEM(names, k, nm);
        ++k;
    }
    if (k == 0) { Py_DECREF(names); Py_DECREF(cols); Py_RETURN_NONE; }
    if (k < n) {                                                 /* the last, short batch: the columns move together */
        PyObject *c2 = PyBytes_FromStringAndSize(NULL, (Py_ssize_t)ni * k * 8);
        if (!c2 || PyList_SetSlice(names, k, n, NULL) < 0) { Py_XDECREF(c2); Py_DECREF(names); Py_DECREF(cols); return NULL; }
        for (j = 0; j < ni; ++j) memcpy(PyBytes_AS_STRING(c2) + (Py_ssize_t)j * k * 8, v + (Py_ssize_t)j * n, (size_t)k * 8);
        Py_DECREF(cols);
        cols = c2;
    }
    out = Py_BuildValue("(nNN)", k, names, cols);
    return out;
}
static PyMethodDef rc_methods[] = {
    {"fetch", (PyCFunction)rc_fetch, METH_O, "fetch(n) -> None | (k, names, int64 columns as bytes)"},
    {NULL, NULL, 0, NULL}};
static PyTypeObject RowCursorType = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "pyfastx_amd._fxobj.RowCursor",
    .tp_basicsize = sizeof(RowCursor),
    .tp_dealloc = (destructor)rc_dealloc,
    .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_methods = rc_methods,
    .tp_new = rc_new,
};

/* read_batch_cols(ReadType, fq, names, cols, seq, qual, offs) -> list: the same objects as read_batch from a RowCursor batch of
 * "SELECT ID, name, dlen, rlen, soff, qoff FROM read" (cols: ID, dlen, rlen, soff, qoff -- k int64 each) */
static PyObject *mod_read_batch_cols(PyObject *m, PyObject *args)
{
    PyObject *type, *fq, *names, *out = NULL;
    Py_buffer cols, seq, qual, offs;
    Py_ssize_t k, i;
    (void)m;
    if (!PyArg_ParseTuple(args, "OOO!y*y*y*y*", &type, &fq, &PyList_Type, &names, &cols, &seq, &qual, &offs)) return NULL;
    k = PyList_GET_SIZE(names);
    if (!PyType_Check(type) || !PyType_IsSubtype((PyTypeObject *)type, &ReadCoreType) || cols.len < 5 * k * 8 || offs.len < (k + 1) * 8) {
        PyErr_SetString(PyExc_TypeError, "read_batch_cols(ReadCore subtype, fq, names, 5 x k int64, seq, qual, int64 offsets[k + 1])");
        goto done;
    }
    out = PyList_New(k);
    for (i = 0; out && i < k; ++i) {
        const long long *v = (const long long *)cols.buf;
        const int64_t *o = (const int64_t *)offs.buf;
        ReadCore *r;
        if (o[i] < 0 || o[i + 1] < o[i] || o[i + 1] > seq.len || o[i + 1] > qual.len) { PyErr_SetString(PyExc_ValueError, "bad offsets"); Py_CLEAR(out); break; }
        r = (ReadCore *)((PyTypeObject *)type)->tp_alloc((PyTypeObject *)type, 0);
        if (!r) { Py_CLEAR(out); break; }
        read_untrack(r);
        PyList_SET_ITEM(out, i, (PyObject *)r);
        r->fq = Py_NewRef(fq);
        r->name = Py_NewRef(PyList_GET_ITEM(names, i));
        r->id = v[i]; r->desc_len = v[k + i]; r->read_len = v[2 * k + i]; r->soff = v[3 * k + i]; r->qoff = v[4 * k + i];
        r->pre_seq = PyUnicode_DecodeLatin1((const char *)seq.buf + o[i], (Py_ssize_t)(o[i + 1] - o[i]), NULL);
        r->pre_qual = PyUnicode_DecodeLatin1((const char *)qual.buf + o[i], (Py_ssize_t)(o[i + 1] - o[i]), NULL);
        if (!r->pre_seq || !r->pre_qual) { Py_CLEAR(out); break; }
    }
done:
    PyBuffer_Release(&cols); PyBuffer_Release(&seq); PyBuffer_Release(&qual); PyBuffer_Release(&offs);
    return out;
}

/* read_batch_arrays(ReadType, fq, first_id, names, name_offs, dlen, rlen, soff, qoff, seq, qual, offs) -> list: the Read objects of
 * reads first_id .. first_id + k - 1 (1-based ids) from slices of the read table as it stands in host memory (int64 each, k
 * entries; name_offs and offs k + 1) and three gathered buffers -- names, sequence lines, quality lines.  No SQLite in the loop:
 * the table came off the device once (fx_fastq_table), where the reference steps a statement per read (fastq.c:566-596). */
static PyObject *mod_read_batch_arrays(PyObject *m, PyObject *args)
{
    PyObject *type, *fq, *out = NULL;
    long long first;
    Py_buffer nb, no, dl, rl, so, qo, seq, qual, offs;
    Py_ssize_t k, i;
    (void)m;
    if (!PyArg_ParseTuple(args, "OOLy*y*y*y*y*y*y*y*y*", &type, &fq, &first, &nb, &no, &dl, &rl, &so, &qo, &seq, &qual, &offs)) return NULL;
    k = rl.len / 8;
    if (!PyType_Check(type) || !PyType_IsSubtype((PyTypeObject *)type, &ReadCoreType) || no.len < (k + 1) * 8 || dl.len < k * 8 || so.len < k * 8 ||
        qo.len < k * 8 || offs.len < (k + 1) * 8) {
        PyErr_SetString(PyExc_TypeError, "read_batch_arrays(ReadCore subtype, fq, first id, names, int64 name offsets[k + 1], dlen, rlen, soff, qoff (int64[k]), seq, qual, int64 offsets[k + 1])");
        goto done;
    }
    out = PyList_New(k);
    for (i = 0; out && i < k; ++i) {
        const int64_t *o = (const int64_t *)offs.buf, *n0 = (const int64_t *)no.buf;
        ReadCore *r;
        if (o[i] < 0 || o[i + 1] < o[i] || o[i + 1] > seq.len || o[i + 1] > qual.len || n0[i] < 0 || n0[i + 1] < n0[i] || n0[i + 1] > nb.len) {
            PyErr_SetString(PyExc_ValueError, "bad offsets");
            Py_CLEAR(out);
            break;
        }
        r = (ReadCore *)((PyTypeObject *)type)->tp_alloc((PyTypeObject *)type, 0);
        if (!r) { Py_CLEAR(out); break; }
        read_untrack(r);
        PyList_SET_ITEM(out, i, (PyObject *)r);
        r->fq = Py_NewRef(fq);
        r->name = PyUnicode_DecodeUTF8((const char *)nb.buf + n0[i], (Py_ssize_t)(n0[i + 1] - n0[i]), "surrogateescape");      /* as fxi.connect's text_factory */
        r->id = first + i;
        r->desc_len = ((const int64_t *)dl.buf)[i]; r->read_len = ((const int64_t *)rl.buf)[i];
        r->soff = ((const int64_t *)so.buf)[i]; r->qoff = ((const int64_t *)qo.buf)[i];
        r->pre_seq = PyUnicode_DecodeLatin1((const char *)seq.buf + o[i], (Py_ssize_t)(o[i + 1] - o[i]), NULL);
        r->pre_qual = PyUnicode_DecodeLatin1((const char *)qual.buf + o[i], (Py_ssize_t)(o[i + 1] - o[i]), NULL);
        if (!r->name || !r->pre_seq || !r->pre_qual) { Py_CLEAR(out); break; }
    }
done:
    PyBuffer_Release(&nb); PyBuffer_Release(&no); PyBuffer_Release(&dl); PyBuffer_Release(&rl); PyBuffer_Release(&so); PyBuffer_Release(&qo);
    PyBuffer_Release(&seq); PyBuffer_Release(&qual); PyBuffer_Release(&offs);
    return out;
}

/* seq_batch_cols(SeqType, fa, names, cols, buf, offs, lens, sel) -> list of Sequence objects: a RowCursor batch of
 * "SELECT ID, chrom, boff, blen, slen, llen, elen, norm, dlen FROM seq" (cols: those 8 integers, k each); the whole sequences
 * of the records sel[t] lie at buf[offs[t] : offs[t] + lens[t]] (one gather for the batch) and ride along in the objects */
static PyObject *mod_seq_batch_cols(PyObject *m, PyObject *args)
{
    PyObject *type, *fa, *names, *out = NULL;
    Py_buffer cols, buf, offs, lens, sel;
    Py_ssize_t k, i, nsel;
    (void)m;
    if (!PyArg_ParseTuple(args, "OOO!y*y*y*y*y*", &type, &fa, &PyList_Type, &names, &cols, &buf, &offs, &lens, &sel)) return NULL;
    k = PyList_GET_SIZE(names);
    nsel = sel.len / 8;
    if (!PyType_Check(type) || !PyType_IsSubtype((PyTypeObject *)type, &SeqCoreType) || cols.len < 8 * k * 8 || offs.len < nsel * 8 || lens.len < nsel * 8) {
        PyErr_SetString(PyExc_TypeError, "seq_batch_cols(SeqCore subtype, fa, names, 8 x k int64, buf, offs, lens, sel)");
        goto done;
    }
    out = PyList_New(k);
    for (i = 0; out && i < k; ++i) {
        const long long *v = (const long long *)cols.buf;
        SeqCore *q = (SeqCore *)((PyTypeObject *)type)->tp_alloc((PyTypeObject *)type, 0);
        if (!q) { Py_CLEAR(out); break; }
        if (((PyTypeObject *)type)->tp_dictoffset == 0) seq_untrack((PyObject *)q);
        PyList_SET_ITEM(out, i, (PyObject *)q);
        q->fa = Py_NewRef(fa);
        q->name = Py_NewRef(PyList_GET_ITEM(names, i));
        q->pre = NULL;
        q->id = v[i]; q->offset = v[k + i]; q->byte_len = v[2 * k + i]; q->full_len = v[3 * k + i];
        q->line_len = v[4 * k + i]; q->end_len = v[5 * k + i]; q->normal = v[6 * k + i]; q->desc_len = v[7 * k + i];
        q->start = 1; q->end = q->full_len; q->seq_len = q->full_len; q->complete = 1; q->reg = -1;
    }
    for (i = 0; out && i < nsel; ++i) {
        const int64_t t = ((const int64_t *)sel.buf)[i], o = ((const int64_t *)offs.buf)[i], l = ((const int64_t *)lens.buf)[i];
        SeqCore *q;
        if (t < 0 || t >= k || o < 0 || l < 0 || o + l > buf.len) { PyErr_SetString(PyExc_ValueError, "bad selection or offsets"); Py_CLEAR(out); break; }
        q = (SeqCore *)PyList_GET_ITEM(out, t);
        q->pre = PyUnicode_DecodeLatin1((const char *)buf.buf + o, (Py_ssize_t)l, NULL);
        if (!q->pre) { Py_CLEAR(out); break; }
    }
done:
    PyBuffer_Release(&cols); PyBuffer_Release(&buf); PyBuffer_Release(&offs); PyBuffer_Release(&lens); PyBuffer_Release(&sel);
    return out;
}

/* ------------------------------------------------------------------ PinnedBuf: owner of a block of fx_pinned_alloc
 * PinnedBuf(address, nbytes, address of fx_pinned_free): exposes the block through the buffer protocol (numpy.frombuffer
 * over it is the array a batched fetch returns) and gives it back to the library's pool when the last view is gone. */
typedef struct { PyObject_HEAD void *ptr; Py_ssize_t len; void (*release)(void *); } PinnedBuf;
static PyObject *pb_new(PyTypeObject *type, PyObject *args, PyObject *kw)
{
    unsigned long long addr, rel;
    Py_ssize_t len;
    PinnedBuf *b;
    (void)kw;
    if (!PyArg_ParseTuple(args, "KnK", &addr, &len, &rel)) return NULL;
    if (!addr || len < 0) { PyErr_SetString(PyExc_ValueError, "PinnedBuf(address, nbytes, release)"); return NULL; }
    b = (PinnedBuf *)type->tp_alloc(type, 0);
    if (!b) return NULL;
    b->ptr = (void *)(uintptr_t)addr; b->len = len; b->release = (void (*)(void *))(uintptr_t)rel;
    return (PyObject *)b;
}
static void pb_dealloc(PinnedBuf *b)
{
    if (b->ptr && b->release) b->release(b->ptr);
    Py_TYPE(b)->tp_free((PyObject *)b);
}
static int pb_getbuffer(PinnedBuf *b, Py_buffer *view, int flags) { return PyBuffer_FillInfo(view, (PyObject *)b, b->ptr, b->len, 0, flags); }
static PyBufferProcs pb_as_buffer = {(getbufferproc)pb_getbuffer, NULL};
static PyTypeObject PinnedBufType = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "pyfastx_amd._fxobj.PinnedBuf",
    .tp_basicsize = sizeof(PinnedBuf),
    .tp_dealloc = (destructor)pb_dealloc,
    .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_as_buffer = &pb_as_buffer,
    .tp_new = pb_new,
};

/* ids_of_names(names, index, out) -> -1, or the position of the first name that `index` does not hold.
 * names: list / tuple of str; index: dict name -> 0-based record id; out: writable buffer of len(names) int64.
 * The names of a query batch are drawn from few records (a genome has a few hundred) and mostly ARE the same str objects
 * again and again (qnames = [names[i] for i in ids]): a small table keyed by the object's address answers those without
 * hashing a single character; the others cost one dict probe (the hash of a str is cached in the object).  One million
 * names: ~3 ms, against ~40 ms for a join + encode + numpy split and a kernel launch (fasta.c:521-546 does one SQLite
 * probe per subscript). */
#define IDC_SLOTS 4096
static PyObject *mod_ids_of_names(PyObject *m, PyObject *args)
{
    PyObject *names, *index, *fast;
    Py_buffer out;
    Py_ssize_t n, i, bad = -1;
    static struct { PyObject *key; long long id; } cache[IDC_SLOTS];
    (void)m;
    if (!PyArg_ParseTuple(args, "OO!w*", &names, &PyDict_Type, &index, &out)) return NULL;
    fast = PySequence_Fast(names, "names: a list or tuple of str");
    if (!fast) { PyBuffer_Release(&out); return NULL; }
    n = PySequence_Fast_GET_SIZE(fast);
    if (out.len < n * 8) { PyErr_SetString(PyExc_ValueError, "ids_of_names: out is too small"); Py_DECREF(fast); PyBuffer_Release(&out); return NULL; }
    memset(cache, 0, sizeof cache);
    {
        PyObject **items = PySequence_Fast_ITEMS(fast);
        long long *o = (long long *)out.buf;
        for (i = 0; i < n; ++i) {
            PyObject *k = items[i];
            const size_t slot = (((uintptr_t)k >> 4) * 0x9E3779B97F4A7C15ull >> 40) & (IDC_SLOTS - 1);
            if (cache[slot].key == k) { o[i] = cache[slot].id; continue; }
            {
                PyObject *v = PyDict_GetItemWithError(index, k);          /* borrowed */
                if (!v) { if (!PyErr_Occurred()) bad = i; break; }
                o[i] = PyLong_AsLongLong(v);
                if (o[i] == -1 && PyErr_Occurred()) break;
                cache[slot].key = k; cache[slot].id = o[i];
            }
        }
    }
    Py_DECREF(fast);
    PyBuffer_Release(&out);
    if (PyErr_Occurred()) return NULL;
    return PyLong_FromSsize_t(bad);
}

/* pack_names(names) -> (bytes, offsets): the UTF-8 bytes of the names back to back (+ 16 zero bytes) and their n + 1 int64
 * offsets as bytes -- the (qbytes, qoff) pair of fx_names_lookup -- in one pass over the list */
static PyObject *mod_pack_names(PyObject *m, PyObject *arg)
{
    PyObject *fast = PySequence_Fast(arg, "names: a list or tuple of str / bytes"), *bytes = NULL, *offs = NULL;
    Py_ssize_t n, i, total = 0;
    (void)m;
    if (!fast) return NULL;
    n = PySequence_Fast_GET_SIZE(fast);
    offs = PyBytes_FromStringAndSize(NULL, (n + 1) * 8);
    if (!offs) goto fail;
    {
        PyObject **items = PySequence_Fast_ITEMS(fast);
        long long *o = (long long *)PyBytes_AS_STRING(offs);
        char *dst;
        for (i = 0; i < n; ++i) {                                         /* sizes first */
            Py_ssize_t l;
            o[i] = total;
            if (PyUnicode_Check(items[i])) { if (!PyUnicode_AsUTF8AndSize(items[i], &l)) { PyErr_Clear(); l = -1; } }
            else if (PyBytes_Check(items[i])) l = PyBytes_GET_SIZE(items[i]);
            else { PyErr_SetString(PyExc_TypeError, "names must be str or bytes"); goto fail; }
            if (l < 0) { PyErr_SetString(PyExc_ValueError, "pack_names: a name that is not valid UTF-8 (use the list path)"); goto fail; }
            total += l;
        }
        o[n] = total;
        bytes = PyBytes_FromStringAndSize(NULL, total + 16);
        if (!bytes) goto fail;
        dst = PyBytes_AS_STRING(bytes);
        for (i = 0; i < n; ++i) {
            Py_ssize_t l;
            const char *p = PyUnicode_Check(items[i]) ? PyUnicode_AsUTF8AndSize(items[i], &l) : (l = PyBytes_GET_SIZE(items[i]), PyBytes_AS_STRING(items[i]));
            memcpy(dst + o[i], p, (size_t)l);
        }
        memset(dst + total, 0, 16);
    }
    Py_DECREF(fast);
    return Py_BuildValue("(NN)", bytes, offs);
fail:
    Py_XDECREF(fast); Py_XDECREF(bytes); Py_XDECREF(offs);
    return NULL;
}

static PyMethodDef mod_methods[] = {
    {"set_read_type", mod_set_read_type, METH_O, "set_read_type(Read type): what fq[i] / fq[name] make"},
    {"ids_of_names", mod_ids_of_names, METH_VARARGS, "ids_of_names(names, index dict, out int64 buffer) -> -1 | position of the first unknown name"},
    {"pack_names", mod_pack_names, METH_O, "pack_names(names) -> (bytes + 16 zero bytes, int64 offsets[n + 1] as bytes)"},
    {"seq_batch_cols", mod_seq_batch_cols, METH_VARARGS, "seq_batch_cols(SeqType, fa, names, cols, buf, offs, lens, sel) -> list of Sequence objects"},
    {"read_batch_arrays", mod_read_batch_arrays, METH_VARARGS, "read_batch_arrays(ReadType, fq, first_id, names, name_offs, dlen, rlen, soff, qoff, seq, qual, offs) -> list of Read objects"},
    {"read_batch_cols", mod_read_batch_cols, METH_VARARGS, "read_batch_cols(ReadType, fq, names, cols, seq, qual, offs) -> list of Read objects"},
    {"read_batch", mod_read_batch, METH_VARARGS, "read_batch(ReadType, fq, rows, seq, qual, offs) -> list of Read objects with their strings"},
    {"fastx_batch", mod_fastx_batch, METH_VARARGS, "fastx_batch(hdr, hdr_off, seq, qual, recs, fastq, with_comment, state) -> list of tuples"},
    {"set_api", mod_set_api, METH_VARARGS, "set_api(address of fx_fetch_one, Sequence type)"},
    {"bench_fetch_one", mod_bench, METH_VARARGS, "bench_fetch_one(handle, off, blen, take, n) -> us per fx_fetch_one call from C"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_fxobj", "C base types of pyfastx_amd.Fasta / Sequence", -1, mod_methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__fxobj(void)
{
    PyObject *m;
    if (PyType_Ready(&SeqCoreType) < 0 || PyType_Ready(&FastaCoreType) < 0 || PyType_Ready(&FastxIterType) < 0 || PyType_Ready(&ReadCoreType) < 0 || PyType_Ready(&RowCursorType) < 0 || PyType_Ready(&PinnedBufType) < 0 || PyType_Ready(&FastqCoreType) < 0) return NULL;
    m = PyModule_Create(&moddef);
    if (!m) return NULL;
    Py_INCREF(&SeqCoreType); Py_INCREF(&FastaCoreType);
    PyModule_AddObject(m, "SeqCore", (PyObject *)&SeqCoreType);
    PyModule_AddObject(m, "FastaCore", (PyObject *)&FastaCoreType);
    Py_INCREF(&FastxIterType);
    PyModule_AddObject(m, "FastxIter", (PyObject *)&FastxIterType);
    Py_INCREF(&ReadCoreType);
    PyModule_AddObject(m, "ReadCore", (PyObject *)&ReadCoreType);
    Py_INCREF(&RowCursorType);
    PyModule_AddObject(m, "RowCursor", (PyObject *)&RowCursorType);
    Py_INCREF(&FastqCoreType);
    PyModule_AddObject(m, "FastqCore", (PyObject *)&FastqCoreType);
    Py_INCREF(&PinnedBufType);
    PyModule_AddObject(m, "PinnedBuf", (PyObject *)&PinnedBufType);
    return m;
}
