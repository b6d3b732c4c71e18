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

typedef int (*fetch_one_fn)(void *h, int64_t off, int64_t blen, int64_t skip, int64_t take, int flags, uint8_t *dst, int64_t *out_len);
static fetch_one_fn g_fetch_one = NULL;
static PyTypeObject *g_seq_type = NULL;          /* api.Sequence (subclass of SeqCore) */

#define FX_GETTER_CAP 65536                      /* larger answers take the Python path (its buffers) */

typedef struct {
    PyObject_HEAD
    PyObject *rows;                              /* dict: name -> (ID, chrom, boff, blen, slen, llen, elen, norm, dlen[, reg]) */
    unsigned long long handle;                   /* fx_handle* of the staged stream, 0 until it is staged */
    int upper;                                   /* Fasta(uppercase=True) */
} FastaCore;

typedef struct {
    PyObject_HEAD
    PyObject *fa, *name;
    long long id, offset, byte_len, full_len, line_len, end_len, normal, desc_len, start, end, seq_len;
    char complete;
    signed char reg;                             /* line-regular: -1 not known yet, 0, 1 */
} SeqCore;

static PyTypeObject FastaCoreType;

/* ------------------------------------------------------------------ SeqCore */
static void seq_dealloc(SeqCore *s)
{
    Py_XDECREF(s->fa);
    Py_XDECREF(s->name);
    Py_TYPE(s)->tp_free((PyObject *)s);
}

static PyObject *seq_new(PyTypeObject *type, PyObject *args, PyObject *kw)
{
    SeqCore *s = (SeqCore *)type->tp_alloc(type, 0);
    (void)args; (void)kw;
    if (s) { s->fa = Py_NewRef(Py_None); s->name = Py_NewRef(Py_None); s->reg = -1; s->start = 1; }
    return (PyObject *)s;
}

static SeqCore *seq_like(SeqCore *p, long long start, long long end, int complete)
{
    PyTypeObject *tp = Py_TYPE(p);
    SeqCore *s = (SeqCore *)tp->tp_alloc(tp, 0);
    if (!s) return NULL;
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

/* the bytes of a slice of a line-regular record, flags as in fxgpu.h (1 upper, 2 reverse, 4 complement); NULL + no error
 * set: not a case for the fast path */
static PyObject *seq_fast(SeqCore *s, int flags)
{
    FastaCore *fa;
    long long bpl, a, b, off, bl;
    int64_t got = 0;
    uint8_t buf[FX_GETTER_CAP];
    if (s->complete || s->reg != 1 || s->seq_len <= 0 || s->seq_len > FX_GETTER_CAP || !g_fetch_one) return NULL;
    if (!PyObject_TypeCheck(s->fa, &FastaCoreType)) return NULL;
    fa = (FastaCore *)s->fa;
    if (!fa->handle) return NULL;
    bpl = s->line_len - s->end_len;
    if (bpl <= 0) return NULL;
    a = s->start - 1; b = s->end;
    off = s->offset + a + s->end_len * (a / bpl);                         /* sequence.c:498-510 */
    bl = (b - a) + (b / bpl - a / bpl) * s->end_len;
    if (g_fetch_one((void *)(uintptr_t)fa->handle, off, bl, 0, s->seq_len, flags | (fa->upper ? 1 : 0), buf, &got) != 0) return NULL;
    return PyUnicode_DecodeLatin1((const char *)buf, (Py_ssize_t)got, NULL);
}

static PyObject *seq_get(SeqCore *s, int flags)
{
    PyObject *r;
    if (s->reg < 0 && !s->complete && s->seq_len > 0) {                   /* ask the Python side once per object (it caches per record) */
        PyObject *v = PyObject_CallMethod((PyObject *)s, "_line_regular", NULL);
        if (!v) return NULL;
        s->reg = (signed char)(PyObject_IsTrue(v) ? 1 : 0);
        Py_DECREF(v);
        if (PyObject_TypeCheck(s->fa, &FastaCoreType)) {                  /* ... and keep it with the row, for the next fa[name] */
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
    {"_fa", T_OBJECT_EX, offsetof(SeqCore, fa), 0, NULL}, {"_name", T_OBJECT_EX, offsetof(SeqCore, name), 0, NULL},
    {"id", T_LONGLONG, offsetof(SeqCore, id), 0, NULL}, {"_offset", T_LONGLONG, offsetof(SeqCore, offset), 0, NULL},
    {"_byte_len", T_LONGLONG, offsetof(SeqCore, byte_len), 0, NULL}, {"_full_len", T_LONGLONG, offsetof(SeqCore, full_len), 0, NULL},
    {"_line_len", T_LONGLONG, offsetof(SeqCore, line_len), 0, NULL}, {"_end_len", T_LONGLONG, offsetof(SeqCore, end_len), 0, NULL},
    {"_normal", T_LONGLONG, offsetof(SeqCore, normal), 0, NULL}, {"_desc_len", T_LONGLONG, offsetof(SeqCore, desc_len), 0, NULL},
    {"start", T_LONGLONG, offsetof(SeqCore, start), 0, NULL}, {"end", T_LONGLONG, offsetof(SeqCore, end), 0, NULL},
    {"_seq_len", T_LONGLONG, offsetof(SeqCore, seq_len), 0, NULL}, {"_complete", T_BOOL, offsetof(SeqCore, complete), 0, NULL},
    {"_reg", T_BYTE, offsetof(SeqCore, reg), 0, NULL}, {NULL, 0, 0, 0, NULL}};
static PyGetSetDef seq_getset[] = {
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
    Py_XDECREF(f->rows);
    Py_TYPE(f)->tp_free((PyObject *)f);
}

static PyObject *fasta_new(PyTypeObject *type, PyObject *args, PyObject *kw)
{
    FastaCore *f = (FastaCore *)type->tp_alloc(type, 0);
    (void)args; (void)kw;
    if (f) { f->rows = PyDict_New(); if (!f->rows) { Py_DECREF(f); return NULL; } }
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
static PyMemberDef fasta_members[] = {
    {"_rows_by_name", T_OBJECT_EX, offsetof(FastaCore, rows), 0, NULL}, {"_core_handle", T_ULONGLONG, offsetof(FastaCore, handle), 0, NULL},
    {"_core_upper", T_INT, offsetof(FastaCore, upper), 0, NULL}, {NULL, 0, 0, 0, NULL}};
static PyGetSetDef fasta_getset[] = {{"_core_tag", (getter)fasta_tag, NULL, NULL, NULL}, {NULL, NULL, NULL, NULL, NULL}};
static PyMappingMethods fasta_mapping = {NULL, (binaryfunc)fasta_subscript, NULL};     /* (__len__ stays with the Python class) */
static PyTypeObject FastaCoreType = {
    PyVarObject_HEAD_INIT(NULL, 0).tp_name = "pyfastx_amd._fxobj.FastaCore", .tp_basicsize = sizeof(FastaCore),
    .tp_dealloc = (destructor)fasta_dealloc, .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_BASETYPE, .tp_new = fasta_new,
    .tp_members = fasta_members, .tp_getset = fasta_getset, .tp_as_mapping = &fasta_mapping,
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

static PyMethodDef mod_methods[] = {
    {"set_api", mod_set_api, METH_VARARGS, "set_api(address of fx_fetch_one, Sequence type)"},
    {"bench_fetch_one", mod_bench, METH_VARARGS, "bench_fetch_one(handle, off, blen, take, n) -> us per fx_fetch_one call from C"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_fxobj", "C base types of pyfastx_amd.Fasta / Sequence", -1, mod_methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__fxobj(void)
{
    PyObject *m;
    if (PyType_Ready(&SeqCoreType) < 0 || PyType_Ready(&FastaCoreType) < 0) return NULL;
    m = PyModule_Create(&moddef);
    if (!m) return NULL;
    Py_INCREF(&SeqCoreType); Py_INCREF(&FastaCoreType);
    PyModule_AddObject(m, "SeqCore", (PyObject *)&SeqCoreType);
    PyModule_AddObject(m, "FastaCore", (PyObject *)&FastaCoreType);
    return m;
}
