"""CPU, only where oracle/_ref is built (this container; /root/reference does not exist on the GPU box): the C
restatement against the REAL reference on seeded random inputs -- every `seq` / `stat` / `comp` row of the index
file the reference writes (index.c:109-388, fasta.c:851-961), every `read` / `stat` / `base` / `meta` row
(fastq.c:8-182, 663-795), whole-record and sliced sequences with all strand flags.  The golden vectors
(test_oracle_golden.py) pin the oracle on fixed inputs everywhere; this pins it on shapes nobody wrote down."""
import glob
import os
import sqlite3
import sys

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def ref():
    if not glob.glob(os.path.join(ROOT, "oracle", "_ref", "pyfastx*.so")):
        pytest.skip("oracle/_ref not built here")
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    import pyfastx
    return pyfastx


def _fasta_text(rng, style):
    eol = b"\r\n" if style["crlf"] else b"\n"
    alpha = np.frombuffer(style["alpha"], dtype=np.uint8)
    out = [eol * int(rng.integers(0, 3))] if style["lead_blank"] else []
    for i in range(int(rng.integers(1, 40))):
        sep = (b" ", b"\t", b"  ", b"")[int(rng.integers(0, 4))]
        out.append(b">r%d_%d" % (i, int(rng.integers(0, 10 ** 6))) + sep + (b"desc |x=1" if sep else b"") + eol)
        n = int(rng.integers(0, style["maxlen"]))
        s = alpha[rng.integers(0, alpha.size, n)].tobytes()
        width = int(rng.integers(1, 90))
        p = 0
        while p < n:
            w = width if not style["ragged"] or rng.random() < 0.8 else int(rng.integers(1, width + 1))
            out.append(s[p:p + w] + eol)
            p += w
        if style["blank_between"] and rng.random() < 0.3:
            out.append(eol)
    raw = b"".join(out)
    if not style["trailing"] and raw.endswith(eol):
        raw = raw[:-len(eol)]
    return raw


_FASTA_STYLES = [
    dict(crlf=False, alpha=b"ACGT", maxlen=3000, ragged=False, trailing=True, lead_blank=False, blank_between=False),
    dict(crlf=True, alpha=b"ACGTNacgtn", maxlen=3000, ragged=False, trailing=True, lead_blank=False, blank_between=False),
    dict(crlf=False, alpha=b"ACGTNRYKMSWBDHVacgtnrykm", maxlen=800, ragged=True, trailing=False, lead_blank=True, blank_between=False),
    dict(crlf=True, alpha=b"ACGTU*-acgu", maxlen=500, ragged=True, trailing=False, lead_blank=False, blank_between=False),
    dict(crlf=False, alpha=b"ACDEFGHIKLMNPQRSTVWY", maxlen=1200, ragged=False, trailing=True, lead_blank=True, blank_between=True),
    dict(crlf=False, alpha=b"ACGT", maxlen=12, ragged=False, trailing=True, lead_blank=False, blank_between=False),
]


@pytest.mark.parametrize("seed", range(int(os.environ.get("FX_FUZZ", "60"))))
def test_fasta_rows_equal_the_reference(oracle, ref, tmp_path, seed):
    rng = np.random.default_rng(9000 + seed)
    style = _FASTA_STYLES[seed % len(_FASTA_STYLES)]
    raw = _fasta_text(rng, style)
    p = str(tmp_path / "r.fa")
    open(p, "wb").write(raw)
    full_name = bool(seed & 1)
    fa = ref.Fasta(p, full_index=True, full_name=full_name)
    db = sqlite3.connect(p + ".fxi")
    want = db.execute("SELECT * FROM seq ORDER BY ID").fetchall()
    recs, tot = oracle.fasta_index(raw, full_name=full_name)
    got = [(i + 1, raw[r["name_off"]:r["name_off"] + r["name_len"]].decode("latin-1")) +
           tuple(int(r[c]) for c in ("boff", "blen", "slen", "llen", "elen", "norm", "dlen")) for i, r in enumerate(recs)]
    assert got == want
    assert db.execute("SELECT seqnum, seqlen FROM stat").fetchone() == (len(recs), tot)
    comp = oracle.fasta_comp(raw, len(recs))
    rec, abc = np.nonzero(comp)
    mine = [(int(r) + 1, int(a), int(comp[r, a])) for r, a in zip(rec, abc)] + [(0, b, int(comp[:, b].sum())) for b in range(128)]
    assert mine == db.execute("SELECT seqid, abc, num FROM comp ORDER BY ID").fetchall()
    for i in rng.integers(0, len(recs), 25).tolist():
        r = recs[i]
        if r["slen"] <= 0 or r["boff"] + r["blen"] > len(raw):
            # blen of the last record of an unterminated file counts a newline that is not there (index.c:231): the
            # reference then despaces one uninitialised byte of its cache buffer through jump_table[] (util.c:173) --
            # found with this test under ASan; nothing to pin, and not safe to call.
            continue
        whole = fa[i].seq
        assert oracle.fetch(raw, r["boff"], r["blen"], r["slen"]).decode("latin-1") == whole
        body = raw[r["boff"]:r["boff"] + r["blen"]].split(b"\n")
        body = (body[:-1] if body[-1] == b"" else body) or [b""]
        uniform = all(len(x) + 1 == r["llen"] for x in body[:-1]) and len(body[-1].rstrip(b"\r")) <= r["llen"] - r["elen"]
        # the line arithmetic of sequence.c:498-510.  Records with ONE odd line (a short middle line, or a last line longer than the first) also carry norm=1 (index.c:342);
        # what the reference returns for a slice of those depends on what its one-entry cache holds (sequence.c:76-125),
        # so there is nothing to pin -- only truly uniform records are compared.
        if r["slen"] > 0 and r["norm"] and uniform:
            a = int(rng.integers(0, r["slen"]))
            b = int(rng.integers(a, r["slen"] + 1))
            off, bl = oracle.slice_range(int(r["boff"]), int(r["llen"]), int(r["elen"]), a, b)
            sub = fa[i][a:b]
            assert oracle.fetch(raw, off, bl, b - a, 0).decode("latin-1") == sub.seq == whole[a:b]
            assert oracle.fetch(raw, off, bl, b - a, 6).decode("latin-1") == sub.antisense
            assert oracle.fetch(raw, off, bl, b - a, 4).decode("latin-1") == sub.complement
            assert oracle.fetch(raw, off, bl, b - a, 2).decode("latin-1") == sub.reverse
    db.close()


def _fastq_text(rng, n, maxlen, crlf, plus_name, trailing, qlo, qhi):
    eol = b"\r\n" if crlf else b"\n"
    alpha = np.frombuffer(b"ACGTNacgtRY", dtype=np.uint8)
    out = []
    for i in range(n):
        k = int(rng.integers(1, maxlen + 1))
        name = b"@q%d:%d" % (i, int(rng.integers(0, 10 ** 5))) + ((b" 1:N:0:ACGT", b"\tt", b"")[i % 3])
        out += [name + eol, alpha[rng.integers(0, alpha.size if i % 7 == 0 else 5, k)].tobytes() + eol,
                (b"+" + name[1:] if plus_name and i % 2 else b"+") + eol,
                rng.integers(qlo, qhi + 1, k).astype(np.uint8).tobytes() + eol]
    raw = b"".join(out)
    return raw if trailing else raw[:-len(eol)]


@pytest.mark.parametrize("seed", range(int(os.environ.get("FX_FUZZ", "40"))))
def test_fastq_rows_equal_the_reference(oracle, ref, tmp_path, seed):
    rng = np.random.default_rng(9500 + seed)
    qlo, qhi = ((33, 73), (35, 74), (64, 104), (59, 104), (66, 100), (33, 126), (40, 40), (33, 80))[seed % 8]
    raw = _fastq_text(rng, int(rng.integers(1, 400)), (150, 9, 2000, 150, 40, 300, 1, 150)[seed % 8], crlf=bool(seed & 1),
                      plus_name=bool(seed & 2), trailing=(seed % 8 not in (3, 4)), qlo=qlo, qhi=qhi)
    p = str(tmp_path / "r.fq")
    open(p, "wb").write(raw)
    fq = ref.Fastq(p, full_index=True)
    db = sqlite3.connect(p + ".fxi")
    rq, size, ln = oracle.fastq_index(raw)
    got = [(i + 1, raw[int(r["name_off"]):int(r["name_off"]) + int(r["name_len"])].decode("latin-1")) +
           tuple(int(r[c]) for c in ("dlen", "rlen", "soff", "qoff")) for i, r in enumerate(rq)]
    assert got == db.execute("SELECT * FROM read ORDER BY ID").fetchall()
    cnt, sz, avg = db.execute("SELECT * FROM stat").fetchone()
    assert (cnt, sz) == (len(rq), size) and avg == size / len(rq)
    c = oracle.fastq_composition(raw)
    assert db.execute("SELECT * FROM base").fetchone() == (c["a"], c["c"], c["g"], c["t"], c["n"])
    assert db.execute("SELECT * FROM meta").fetchone() == (c["maxlen"], c["minlen"], c["minqs"], c["maxqs"], c["phred"])
    for i in rng.integers(0, len(rq), 20).tolist():
        r = rq[i]
        assert fq[i].seq.encode("latin-1") == raw[int(r["soff"]):int(r["soff"]) + int(r["rlen"])]
        assert fq[i].qual.encode("latin-1") == raw[int(r["qoff"]):int(r["qoff"]) + int(r["rlen"])]
    db.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("FX_FUZZ", "8"))))
def test_kseq_oracle_equals_the_reference_fastx(oracle, ref, tmp_path, seed):
    """fxo_kseq (kseq.c:138-179 restated) + the tuple builders of fastx.c against the compiled reference's Fastx, on files
    in which everything kseq tolerates happens (tests/kseq_cases.py), both builders, comment and uppercase options."""
    import random
    from kseq_cases import FIXED, gen
    rng = random.Random(4100 + seed)
    p = str(tmp_path / "t.fx")
    for data in (FIXED if seed == 0 else []) + [gen(rng) for _ in range(150)]:
        if oracle.kseq_undefined(data):          # the reference reads a buffer it never wrote: nothing to compare with
            continue
        with open(p, "wb") as f:
            f.write(data)
        for fmt in ("fasta", "fastq"):
            for kw in (dict(), dict(comment=True), dict(uppercase=True, comment=True)):
                assert list(ref.Fastx(p, format=fmt, **kw)) == oracle.fastx_tuples(data, fmt, **kw), (data, fmt, kw)


def test_index_free_iteration_oracle_equals_the_reference(oracle, ref, tmp_path):
    """fxoracle.index_free_tuples (kseq_read's records as Fasta / Fastq hand them out without an index, index.c:609-664 and
    fastq.c:598-622, full_name joining name and comment with one space) against the compiled reference."""
    import random
    from kseq_cases import FIXED, gen
    rng = random.Random(6100)
    seen = 0
    for data in list(FIXED) + [gen(rng) for _ in range(400)]:
        if oracle.kseq_undefined(data):
            continue
        first = data.lstrip()[:1]
        for kind, lead, ext in (("fasta", b">", "fa"), ("fastq", b"@", "fq")):
            if first != lead:                              # the constructors refuse anything else (fasta.c:107-110, fastq.c)
                continue
            p = str(tmp_path / ("t." + ext))
            with open(p, "wb") as f:
                f.write(data)
            for full_name in (False, True):
                if kind == "fasta":
                    for up in (False, True):
                        assert list(ref.Fasta(p, build_index=False, full_name=full_name, uppercase=up)) == \
                            oracle.index_free_tuples(data, kind, full_name, up), (data, full_name, up)
                else:
                    assert list(ref.Fastq(p, build_index=False, full_name=full_name)) == oracle.index_free_tuples(data, kind, full_name), (data, full_name)
                seen += 1
    assert seen > 300


@pytest.mark.parametrize("members", [1, 3])
def test_refshim_serves_reads_from_imported_points(ref, tmp_path, members):
    """The zran work-alike under the compiled reference (oracle/refshim/zran.c; indexed_gzip is not part of the reference
    tree): points built at deflate block boundaries, exported by the reference's own util.c:442-540, imported again by
    util.c:542-726 on the next open, and every read served by a raw inflate from the last point at or before the offset
    (bits primed, window as dictionary) -- the bytes equal the plain text, the counters say how the seeks were served."""
    import ctypes
    import gzip
    so = ctypes.CDLL(ref.__file__)
    so.fxshim_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    so.fxshim_point_hits.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32]
    rng = np.random.default_rng(11 + members)
    recs = [("c%d" % i, bytes(rng.choice(list(b"ACGTNacgt"), int(rng.integers(60_000, 300_000))).astype(np.uint8)).decode()) for i in range(24)]
    text = "".join(">%s d\n%s\n" % (n, "\n".join(s[k:k + 60] for k in range(0, len(s), 60))) for n, s in recs).encode()
    cut = [len(text) * k // members for k in range(members + 1)]
    p = str(tmp_path / "s.fa.gz")
    open(p, "wb").write(b"".join(gzip.compress(text[cut[k]:cut[k + 1]], 6) for k in range(members)))

    def stats():
        a = (ctypes.c_uint64 * 6)()
        so.fxshim_stats(a)
        return [int(x) for x in a]
    so.fxshim_reset()
    fa = ref.Fasta(p)                                          # scan + zran_build_index + export
    built = stats()[4]
    assert built >= 3 and stats()[5] == 0
    del fa
    so.fxshim_reset()
    fa = ref.Fasta(p)                                          # load_index: import, nothing is built
    assert stats()[4] == 0
    for _ in range(600):
        k = int(rng.integers(0, len(recs)))
        n, s = recs[k]
        a = int(rng.integers(0, len(s) - 1))
        b = min(len(s), a + int(rng.integers(1, 300)))
        assert fa[n][a:b].seq == s[a:b]
    assert fa[recs[-1][0]].seq == recs[-1][1]
    seeks, from_point, from_start, continued, built2, errors = stats()
    hits = (ctypes.c_uint32 * built)()
    so.fxshim_point_hits(hits, built)
    assert errors == 0 and built2 == 0 and from_start == 0 and from_point >= 300 and sum(1 for x in hits if x) >= built - 1
