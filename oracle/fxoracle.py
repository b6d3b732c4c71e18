"""ctypes binding of the CPU oracle (oracle/fx_oracle.c).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg as the *checker*.  Nothing under pyfastx_amd/
imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FASTA_REC = np.dtype([
    ("hoff", "<i8"), ("boff", "<i8"), ("blen", "<i8"), ("slen", "<i8"), ("llen", "<i8"),
    ("name_off", "<i8"), ("name_len", "<i4"), ("elen", "<i4"), ("norm", "<i4"), ("dlen", "<i4"),
], align=True)

FASTQ_REC = np.dtype([
    ("name_off", "<i8"), ("rlen", "<i8"), ("soff", "<i8"), ("qoff", "<i8"),
    ("name_len", "<i4"), ("dlen", "<i4"),
], align=True)


class FastqComp(C.Structure):
    _fields_ = [("a", C.c_int64), ("c", C.c_int64), ("g", C.c_int64), ("t", C.c_int64),
                ("n", C.c_int64), ("maxlen", C.c_int64), ("minlen", C.c_int64),
                ("minqs", C.c_int32), ("maxqs", C.c_int32), ("phred", C.c_int32)]


def build():
    """(Re)build oracle/libfxoracle.so with gcc."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "port"])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libfxoracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
        L.fxo_fasta_index.restype = i64
        L.fxo_fasta_index.argtypes = [vp, i64, i32, vp, i64, C.POINTER(i64)]
        L.fxo_fasta_comp.restype = i64
        L.fxo_fasta_comp.argtypes = [vp, i64, vp, i64]
        L.fxo_fastq_index.restype = i64
        L.fxo_fastq_index.argtypes = [vp, i64, vp, i64, C.POINTER(i64), C.POINTER(i64)]
        L.fxo_fastq_composition.restype = None
        L.fxo_fastq_composition.argtypes = [vp, i64, C.POINTER(FastqComp)]
        L.fxo_despace.restype = i64
        L.fxo_despace.argtypes = [vp, i64, i32]
        L.fxo_revcomp.restype = None
        L.fxo_revcomp.argtypes = [vp, i64, i32]
        L.fxo_slice_range.restype = None
        L.fxo_slice_range.argtypes = [i64, i64, C.c_int32, i64, i64, C.POINTER(i64), C.POINTER(i64)]
        L.fxo_fetch.restype = i64
        L.fxo_fetch.argtypes = [vp, i64, i64, i64, i64, i32, vp]
        L.fxo_kseq.restype = i64
        L.fxo_kseq.argtypes = [vp, i64, vp, i64, vp, vp, C.POINTER(C.c_int)]
        L.fxo_quali.restype = None
        L.fxo_quali.argtypes = [vp, i64, i64, i32, vp]
        _LIB = L
    return _LIB


def _buf(data):
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data, a.size


def fasta_index(data, full_name=False):
    """-> (records structured array, total seqlen)."""
    a, p, n = _buf(data)
    tot = C.c_int64(0)
    cnt = lib().fxo_fasta_index(p, n, int(full_name), None, 0, C.byref(tot))
    out = np.zeros(cnt, dtype=FASTA_REC)
    lib().fxo_fasta_index(p, n, int(full_name), out.ctypes.data, cnt, C.byref(tot))
    return out, tot.value


def fasta_comp(data, nrec):
    a, p, n = _buf(data)
    comp = np.zeros((nrec, 128), dtype=np.int64)
    lib().fxo_fasta_comp(p, n, comp.ctypes.data, nrec)
    return comp


def fastq_index(data):
    """-> (records, size, line_num)."""
    a, p, n = _buf(data)
    size, ln = C.c_int64(0), C.c_int64(0)
    cnt = lib().fxo_fastq_index(p, n, None, 0, C.byref(size), C.byref(ln))
    out = np.zeros(cnt, dtype=FASTQ_REC)
    lib().fxo_fastq_index(p, n, out.ctypes.data, cnt, C.byref(size), C.byref(ln))
    return out, size.value, ln.value


def fastq_composition(data):
    a, p, n = _buf(data)
    o = FastqComp()
    lib().fxo_fastq_composition(p, n, C.byref(o))
    return {k: getattr(o, k) for k, _ in FastqComp._fields_}


def despace(b, upper=False):
    a = np.frombuffer(bytes(b), dtype=np.uint8).copy()
    m = lib().fxo_despace(a.ctypes.data, a.size, int(upper))
    return a[:m].tobytes()


def revcomp(b, mode=3):
    a = np.frombuffer(bytes(b), dtype=np.uint8).copy()
    lib().fxo_revcomp(a.ctypes.data, a.size, mode)
    return a.tobytes()


def slice_range(boff, llen, elen, start, stop):
    off, bl = C.c_int64(0), C.c_int64(0)
    lib().fxo_slice_range(boff, llen, elen, start, stop, C.byref(off), C.byref(bl))
    return off.value, bl.value


def fetch(data, off, blen, slen, flags=0):
    a, p, n = _buf(data)
    out = np.zeros(max(int(blen), 1), dtype=np.uint8)
    m = lib().fxo_fetch(p, n, int(off), int(blen), int(slen), int(flags), out.ctypes.data)
    return out[:m].tobytes()


def fetch_batch(data, off, blen, slen, flags=0):
    """Loop of fxo_fetch -> (concatenated bytes array, offsets int64[n+1])."""
    a, p, n = _buf(data)
    off = np.asarray(off, dtype=np.int64)
    blen = np.asarray(blen, dtype=np.int64)
    slen = np.asarray(slen, dtype=np.int64)
    flags = np.broadcast_to(np.asarray(flags, dtype=np.int32), off.shape)
    cap = int(np.minimum(blen, slen).clip(min=0).sum()) + 1
    out = np.zeros(cap, dtype=np.uint8)
    offs = np.zeros(off.size + 1, dtype=np.int64)
    L = lib()
    base = out.ctypes.data
    w = 0
    for i in range(off.size):
        w += L.fxo_fetch(p, n, int(off[i]), int(blen[i]), int(slen[i]), int(flags[i]), base + w)
        offs[i + 1] = w
    return out[:w], offs


def quali(data, qoff, rlen, phred=0):
    a, p, n = _buf(data)
    out = np.zeros(int(rlen), dtype=np.int8)
    lib().fxo_quali(p, int(qoff), int(rlen), int(phred), out.ctypes.data)
    return out


KSEQ_REC = np.dtype([("name_off", "<i8"), ("name_len", "<i8"), ("com_off", "<i8"), ("com_len", "<i8"),
                     ("seq_off", "<i8"), ("seq_len", "<i8"), ("qual_off", "<i8"), ("qual_len", "<i8")])


def kseq(data):
    """kseq_read over the whole buffer -> (records, seq bytes, qual bytes, end code)."""
    a, p, n = _buf(data)
    cap = int(np.count_nonzero((a == 62) | (a == 64))) + 1
    recs = np.zeros(cap, dtype=KSEQ_REC)
    seq = np.zeros(n + 1, dtype=np.uint8)
    qual = np.zeros(n + 1, dtype=np.uint8)
    code = C.c_int(0)
    k = lib().fxo_kseq(p, n, recs.ctypes.data, cap, seq.ctypes.data, qual.ctypes.data, C.byref(code))
    return recs[:k], seq, qual, code.value


def kseq_undefined(data):
    """Does the reference read a quality buffer it never wrote on this input (see fastx_tuples)?"""
    seen = False
    for r in kseq(data)[0]:
        if r["qual_len"] >= 0:
            seen = True
        elif r["qual_len"] == -2 and not seen:
            return True
    return False


def _cstr(b):
    """Py_BuildValue "s": the bytes up to the first NUL, as text."""
    b = bytes(b)
    z = b.find(b"\0")
    return (b if z < 0 else b[:z]).decode("utf-8", "surrogateescape")


def fastx_tuples(data, fmt, uppercase=False, comment=False):
    """What iterating pyfastx.Fastx over a file of these bytes yields (fastx.c:6-30, 124-130): the tuples of the
    builder fmt ("fasta" / "fastq") selects, whatever kind of record kseq_read found -- a FASTA-style record seen
    through the FASTQ builder carries whatever the quality buffer still holds (None before the first quality string,
    kseq.c:147 resets only its length) and so does a record whose quality read met the end of the stream at once; a
    comment is None until the comment buffer exists.  (Where the reference reads a buffer it never wrote -- such a
    record before any quality string -- the value here is None; the reference's is not defined.)"""
    a, _, _ = _buf(data)
    raw = a.tobytes()
    recs, seq, qual, _ = kseq(a)
    out, have_comment, last_qual = [], False, None
    for r in recs:
        name = _cstr(raw[r["name_off"]:r["name_off"] + r["name_len"]])
        s = seq[r["seq_off"]:r["seq_off"] + r["seq_len"]].tobytes()
        if r["com_len"] >= 0:
            have_comment = True
        com = None
        if have_comment:
            com = raw[r["com_off"]:r["com_off"] + max(int(r["com_len"]), 0)].decode("utf-8", "surrogateescape")
        if r["qual_len"] >= 0:
            last_qual = _cstr(qual[r["qual_off"]:r["qual_off"] + r["qual_len"]].tobytes())
        if fmt == "fasta":
            if uppercase:
                s = s.upper()
            out.append((name, _cstr(s), com) if comment else (name, _cstr(s)))
        else:
            out.append((name, _cstr(s), last_qual, com) if comment else (name, _cstr(s), last_qual))
    return out


def index_free_tuples(data, kind, full_name=False, uppercase=False):
    """What iterating pyfastx.Fasta(path, build_index=False) (kind "fasta": index.c:609-664) or pyfastx.Fastq(path,
    build_index=False) (kind "fastq": fastq.c:598-622) yields: kseq_read's records as (name, seq) / (name, seq, qual); with
    full_name a non-empty comment is joined to the name with one space ("%s %s"), whatever the delimiter was."""
    a, _, _ = _buf(data)
    raw = a.tobytes()
    recs, seq, qual, _ = kseq(a)
    out, last_qual = [], None
    for r in recs:
        name = _cstr(raw[r["name_off"]:r["name_off"] + r["name_len"]])
        if full_name and r["com_len"] > 0:
            name = name + " " + _cstr(raw[r["com_off"]:r["com_off"] + r["com_len"]])
        s = seq[r["seq_off"]:r["seq_off"] + r["seq_len"]].tobytes()
        if r["qual_len"] >= 0:
            last_qual = _cstr(qual[r["qual_off"]:r["qual_off"] + r["qual_len"]].tobytes())
        if kind == "fasta":
            out.append((name, _cstr(s.upper() if uppercase else s)))
        else:
            out.append((name, _cstr(s), last_qual))
    return out
