"""-m gpu, where oracle/_ref travelled with the tree (it is built in the container that has /root/reference; nothing here
reads /root/reference): pyfastx_amd -- the product, HIP kernels behind the C ABI -- and the REAL reference side by side on
seeded random files.  The index files row for row, then the objects: names, whole sequences, slices, strands, fetch(),
flank(), composition, statistics, keys; reads, qualities, quality integers."""
import glob
import os
import shutil
import sqlite3
import sys

import numpy as np
import pytest

from conftest import ROOT
from test_oracle_vs_reference import _FASTA_STYLES, _fasta_text, _fastq_text

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def both():
    if not glob.glob(os.path.join(ROOT, "oracle", "_ref", "pyfastx*.so")):
        pytest.fail("oracle/_ref (the compiled reference: `make -C oracle ref` where /root/reference exists, it travels with gpurun) is "
                    "missing: the side-by-side parity tests cannot be skipped on the GPU box")
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    import pyfastx
    import pyfastx_amd
    from pyfastx_amd import _lib
    assert _lib.lib().fx_device_count() >= 1
    return pyfastx_amd, pyfastx


def _two_copies(tmp_path, name, raw):
    out = []
    for d in ("ours", "theirs"):
        os.makedirs(tmp_path / d, exist_ok=True)
        p = str(tmp_path / d / name)
        with open(p, "wb") as f:
            f.write(raw)
        out.append(p)
    return out


def _tables(path, names):
    db = sqlite3.connect(path + ".fxi")
    out = {t: db.execute("SELECT * FROM %s" % t).fetchall() for t in names}
    db.close()
    return out


@pytest.mark.parametrize("seed", range(int(os.environ.get("FX_FUZZ", "24"))))
def test_fasta_side_by_side(both, tmp_path, seed):
    fx, ref = both
    rng = np.random.default_rng(8100 + seed)
    style = dict(_FASTA_STYLES[seed % len(_FASTA_STYLES)])
    raw = _fasta_text(rng, style)
    po, pt = _two_copies(tmp_path, "r.fa", raw)
    kw = dict(full_index=True, full_name=bool(seed & 1), uppercase=bool(seed & 2))
    fa, rf = fx.Fasta(po, **kw), ref.Fasta(pt, **kw)
    a, b = _tables(po, ("seq", "comp")), _tables(pt, ("seq", "comp"))
    assert a["seq"] == b["seq"] and a["comp"] == b["comp"]
    db = sqlite3.connect(po + ".fxi"); mine = db.execute("SELECT seqnum, seqlen FROM stat").fetchone(); db.close()
    db = sqlite3.connect(pt + ".fxi"); theirs = db.execute("SELECT seqnum, seqlen FROM stat").fetchone(); db.close()
    assert mine == theirs
    n = len(rf)
    assert len(fa) == n and fa.size == rf.size and fa.composition == rf.composition and fa.type == rf.type
    if sum(fa.composition.get(c, 0) for c in "ACGTacgt") > 0:
        assert fa.gc_content == rf.gc_content
    def same(f):                                             # equal values, or the same exception with the same text
        out = []
        for o in (fa, rf):
            try:
                out.append(f(o))
            except Exception as e:
                out.append((type(e).__name__, str(e)))
        assert out[0] == out[1], out
    # (statistics: the product answers them from the device sort of the lengths -- the table is resident after the build)
    for f in (lambda o: o.mean, lambda o: o.median, lambda o: o.nl(50), lambda o: o.nl(90), lambda o: o.nl(50), lambda o: o.nl(0),
              lambda o: o.nl(100), lambda o: o.nl(37), lambda o: o.count(100), lambda o: o.count(0), lambda o: o.count(10 ** 9),
              lambda o: o.longest.name, lambda o: o.shortest.name, lambda o: o.longest.id, lambda o: o.shortest.id, lambda o: o.gc_skew):
        same(f)
    mine = sqlite3.connect(po + ".fxi").execute("SELECT * FROM stat").fetchall()
    theirs = sqlite3.connect(pt + ".fxi").execute("SELECT * FROM stat").fetchall()
    assert mine == theirs                                    # avglen / medlen / n50 / l50 as the reference caches them
    assert list(fa.keys()) == list(rf.keys()) and list(fa.keys().sort("length", reverse=True)) == list(rf.keys().sort("length", reverse=True))
    safe = [r for r in a["seq"] if r[4] > 0 and r[2] + r[3] <= len(raw)]          # not empty, not the unterminated last record (UB there)
    for row in [safe[i] for i in rng.integers(0, len(safe), min(30, len(safe))).tolist()] if safe else []:
        i, name, slen = row[0] - 1, row[1], row[4]
        s, t = fa[i], rf[i]
        # (.end of a whole sequence taken by subscript is left 0 by the reference, index.c:483 vs :522: compared with its length)
        assert (s.name, s.id, len(s), s.start, s.end) == (t.name, t.id, len(t), t.start, len(t)) and repr(s) == repr(t)
        assert fa[name].id == rf[name].id and (name in fa) and ("no such" not in fa)
        whole = t.seq                                         # warms the reference's one-entry cache: its slices below are true slices
        assert s.seq == whole and s.description == t.description and s.raw == t.raw
        assert s.antisense == t.antisense and s.complement == t.complement and s.reverse == t.reverse
        assert s.composition == t.composition
        x = int(rng.integers(0, slen)); y = int(rng.integers(x, slen + 1))
        if y > x:
            sub, tub = s[x:y], t[x:y]
            assert sub.seq == tub.seq == whole[x:y] and sub.name == tub.name and (sub.start, sub.end) == (tub.start, tub.end)
            assert sub.antisense == tub.antisense
            assert fa.fetch(name, (x + 1, y)) == rf.fetch(name, (x + 1, y))
            assert fa.fetch(name, [(x + 1, y), (1, 1)], strand="-") == rf.fetch(name, [(x + 1, y), (1, 1)], strand="-")
            assert fa.flank(name, x + 1, y, flank_length=7) == rf.flank(name, x + 1, y, flank_length=7)
        assert s[x] == t[x]
    del rf


@pytest.mark.parametrize("seed", range(int(os.environ.get("FX_FUZZ", "16"))))
def test_fastq_side_by_side(both, tmp_path, seed):
    fx, ref = both
    rng = np.random.default_rng(8500 + seed)
    qlo, qhi = ((33, 73), (35, 74), (64, 104), (59, 104), (66, 100), (33, 126), (40, 40), (33, 80))[seed % 8]
    raw = _fastq_text(rng, int(rng.integers(1, 3000)), (150, 9, 2000, 150, 40, 300, 1, 150)[seed % 8], crlf=bool(seed & 1),
                      plus_name=bool(seed & 2), trailing=(seed % 8 not in (3, 4)), qlo=qlo, qhi=qhi)
    po, pt = _two_copies(tmp_path, "r.fq", raw)
    fq, rq = fx.Fastq(po, full_index=True), ref.Fastq(pt, full_index=True)
    a, b = _tables(po, ("read", "stat", "base", "meta")), _tables(pt, ("read", "stat", "base", "meta"))
    assert a == b
    n = len(rq)
    assert len(fq) == n and fq.size == rq.size and fq.avglen == rq.avglen and fq.composition == rq.composition
    assert (fq.maxlen, fq.minlen, fq.maxqual, fq.minqual, fq.phred) == (rq.maxlen, rq.minlen, rq.maxqual, rq.minqual, rq.phred)
    assert fq.encoding_type == rq.encoding_type and fq.gc_content == rq.gc_content
    assert list(fq.keys()) == list(rq.keys())
    for i in rng.integers(0, n, 40).tolist():
        r, t = fq[i], rq[i]
        assert (r.id, r.name, len(r), r.seq, r.qual, r.quali) == (t.id, t.name, len(t), t.seq, t.qual, t.quali)
        assert r.description == t.description and repr(r) == repr(t)
        if i < n - 1 or raw.endswith(b"\n"):                 # the last read of an unterminated file: the reference's raw runs two bytes
            assert r.raw == t.raw                             # past the end of the file into whatever its buffer held (read.c:124-150)
        assert r.antisense == t.antisense and r.reverse == t.reverse and r.complement == t.complement
        assert fq[t.name].id == t.id and t.name in fq
    for k, (r, t) in enumerate(zip(fq, rq)):                 # iteration (batched fetches on our side)
        assert (r.name, r.seq, r.qual) == (t.name, t.seq, t.qual)
        if k > 300:
            break
    del rq


@pytest.mark.parametrize("seed", range(6))
def test_gzip_inputs_side_by_side(both, tmp_path, seed):
    """The same for compressed inputs: plain gzip (inflated on the host here) and BGZF (inflated on the GPU) against the
    reference reading through its gzip layer.  Offsets in the index are offsets in the inflated stream on both sides."""
    import gzip
    from pyfastx_amd import synth
    fx, ref = both
    rng = np.random.default_rng(8800 + seed)
    raw = _fasta_text(rng, dict(_FASTA_STYLES[seed % 2]))    # line-regular styles (LF / CRLF)
    while len(raw) < 150_000:                                 # several BGZF members
        raw += _fasta_text(rng, dict(_FASTA_STYLES[seed % 2])).replace(b">r", b">s%d_" % len(raw))
    payload = synth.bgzf_compress(raw, block=int(rng.integers(3000, 60000))) if seed & 1 else gzip.compress(raw, 6)
    po, pt = _two_copies(tmp_path, "r.fa.gz", payload)
    fa, rf = fx.Fasta(po, full_index=True), ref.Fasta(pt, full_index=True)
    a, b = _tables(po, ("seq", "comp")), _tables(pt, ("seq", "comp"))
    assert a == b and fa.is_gzip and rf.is_gzip and len(fa) == len(rf) and fa.size == rf.size
    rows = [r for r in a["seq"] if r[4] > 0]
    for row in [rows[i] for i in rng.integers(0, len(rows), 25).tolist()]:
        i, name, slen = row[0] - 1, row[1], row[4]
        whole = rf[i].seq
        assert fa[i].seq == whole and fa[name].antisense == rf[name].antisense
        x = int(rng.integers(0, slen)); y = int(rng.integers(x, slen + 1))
        if y > x:
            assert fa[i][x:y].seq == whole[x:y] and fa.fetch(name, (x + 1, y), strand="-") == rf.fetch(name, (x + 1, y), strand="-")
    rawq = _fastq_text(rng, 1500, 150, crlf=bool(seed & 2), plus_name=False, trailing=True, qlo=33, qhi=74)
    payload = synth.bgzf_compress(rawq, block=20000) if seed & 1 else gzip.compress(rawq, 6)
    qo, qt = _two_copies(tmp_path, "r.fq.gz", payload)
    fq, rq = fx.Fastq(qo, full_index=True), ref.Fastq(qt, full_index=True)
    assert _tables(qo, ("read", "stat", "base", "meta")) == _tables(qt, ("read", "stat", "base", "meta"))
    for i in rng.integers(0, len(rq), 30).tolist():
        assert (fq[i].name, fq[i].seq, fq[i].qual, fq[i].quali) == (rq[i].name, rq[i].seq, rq[i].qual, rq[i].quali)
    del rf, rq


@pytest.mark.parametrize("seed", range(6))
def test_fastx_side_by_side(both, tmp_path, seed):
    """Fastx (fastx.c): index-free iteration -- (name, seq[, comment]) / (name, seq, qual[, comment]) tuples of well-formed
    files, plain and gzip, with and without upper-casing, against the reference's kseq walk."""
    import gzip
    fx, ref = both
    rng = np.random.default_rng(8900 + seed)
    raw = _fasta_text(rng, dict(_FASTA_STYLES[seed % 2]))              # line-regular FASTA, LF / CRLF, lower case in style 1
    rawq = _fastq_text(rng, 700, 150, crlf=bool(seed & 1), plus_name=bool(seed & 2), trailing=(seed != 4), qlo=33, qhi=74)
    for name, payload in (("x.fa", raw), ("x.fq", rawq), ("x.fa.gz", gzip.compress(raw)), ("x.fq.gz", gzip.compress(rawq))):
        po, pt = _two_copies(tmp_path, name, payload)
        for kw in (dict(), dict(comment=True), dict(uppercase=True), dict(uppercase=True, comment=True)):
            ours, theirs = fx.Fastx(po, **kw), ref.Fastx(pt, **kw)
            assert repr(ours).split(" ")[:2] == repr(theirs).split(" ")[:2]
            assert list(ours) == list(theirs), (name, kw)
        assert not os.path.exists(po + ".fxi")                            # no index file is written
    with pytest.raises(FileExistsError):
        fx.Fastx(str(tmp_path / "nope.fa"))
    bad = tmp_path / "bad.txt"
    bad.write_text("hello\n")
    with pytest.raises(RuntimeError):
        fx.Fastx(str(bad))


def test_non_ascii_names_side_by_side(both, tmp_path):
    """UTF-8 (and invalid UTF-8) bytes in header lines: the `chrom` column, by-name access and descriptions agree with the
    reference on the bulk path, the INSERT path (an existing empty directory entry forces it via index_file in :memory:
    style is not comparable, so key_func) and after re-opening the index."""
    fx, ref = both
    raw = ">é first record\nACGTACGT\nAC\n>日本 x\nGGGG\n>plain\nTT\n".encode("utf-8")
    po, pt = _two_copies(tmp_path, "u.fa", raw)
    fa, rf = fx.Fasta(po), ref.Fasta(pt)
    assert _tables(po, ("seq",)) == _tables(pt, ("seq",))
    for name in ("é", "日本", "plain"):
        assert fa[name].seq == rf[name].seq and fa[name].description == rf[name].description and fa[name].name == rf[name].name
        assert (name in fa) and (name in rf)
    assert list(fa.keys()) == list(rf.keys())
    buf, offs = fa.fetch_many(["日本", "é"], [0, 2], [4, 9])
    assert buf.tobytes() == b"GGGG" + b"GTACGTA"
    del fa, rf
    os.unlink(po + ".fxi"); os.unlink(pt + ".fxi")
    ident = lambda h: h.split()[0]                          # noqa: E731  (the INSERT path: key_func)
    fa, rf = fx.Fasta(po, key_func=ident), ref.Fasta(pt, key_func=ident)
    assert _tables(po, ("seq",)) == _tables(pt, ("seq",))
    assert fa["é"].seq == rf["é"].seq
    del fa, rf
    fa, rf = fx.Fasta(po), ref.Fasta(pt)                    # loaded from the files written above
    assert fa["日本"].seq == rf["日本"].seq == "GGGG"


def test_batched_cli_side_by_side(both, tmp_path):
    """`subseq -r / -b / regions`, `sample`, `extract` of pyfastx_amd.cli (one GPU batch each) write the bytes
    the reference's command loops write (pyfastxcli.py:240-387, restated here over the compiled reference's objects)."""
    import random
    fx, ref = both
    from pyfastx_amd import cli
    rng = np.random.default_rng(99)
    raw = _fasta_text(rng, dict(_FASTA_STYLES[0]))
    rawq = _fastq_text(rng, 300, 120, crlf=False, plus_name=False, trailing=True, qlo=33, qhi=74)
    (po, pt), (qo, qt) = _two_copies(tmp_path, "c.fa", raw), _two_copies(tmp_path, "c.fq", rawq)
    rf, rq = ref.Fasta(pt), ref.Fastq(qt)
    names = list(rf.keys())
    regions = []
    for _ in range(200):
        nm = names[int(rng.integers(0, len(names)))]
        L = len(rf[nm])
        if L < 2:
            continue
        a = int(rng.integers(1, L))
        regions.append((nm, a, int(rng.integers(a, L + 1))))
    reg_file, bed_file, out = str(tmp_path / "r.txt"), str(tmp_path / "r.bed"), str(tmp_path / "out.txt")
    open(reg_file, "w").write("".join("%s\t%d\t%d\n" % r for r in regions))
    open(bed_file, "w").write("".join("%s\t%d\t%d\n" % (c, s - 1, e) for c, s, e in regions))
    want = "".join(">%s:%d-%d\n%s\n" % (c, s, e, rf.fetch(c, (s, e))) for c, s, e in regions)
    cli.main(["subseq", "-r", reg_file, "-o", out, po]); assert open(out).read() == want
    cli.main(["subseq", "-b", bed_file, "-o", out, po]); assert open(out).read() == want
    cli.main(["subseq", "-o", out, po] + ["%s:%d-%d" % r for r in regions[:20]])
    assert open(out).read() == "".join(">%s:%d-%d\n%s\n" % (c, s, e, rf[c][s - 1:e].seq) for c, s, e in regions[:20])
    for path_o, obj in ((po, rf), (qo, rq)):
        for seed, num in ((7, 25), (11, 3)):
            random.seed(seed)
            sel = sorted(random.sample(range(len(obj)), k=num))
            cli.main(["sample", "-n", str(num), "-s", str(seed), "-o", out, path_o])
            assert open(out).read() == "".join(obj[i].raw for i in sel), (path_o, seed)
        cli.main(["sample", "-p", "0.1", "-s", "5", "-o", out, path_o])
        random.seed(5)
        sel = sorted(random.sample(range(len(obj)), k=int(np.ceil(len(obj) * 0.1))))
        assert open(out).read() == "".join(obj[i].raw for i in sel)
    # extract: by names on the command line, by a list file (list order), --sequential-read (file order, once each)
    pick = [names[i] for i in rng.integers(0, len(names), 15)]
    cli.main(["extract", "-o", out, po] + pick); assert open(out).read() == "".join(rf[n].raw for n in pick)
    rnames = [rq[int(i)].name for i in rng.integers(0, len(rq), 40)]
    lst = str(tmp_path / "names.txt")
    open(lst, "w").write("".join(n + "\n" for n in rnames))
    cli.main(["extract", "-l", lst, "-o", out, qo]); assert open(out).read() == "".join(rq[n].raw for n in rnames)
    cli.main(["extract", "-l", lst, "--sequential-read", "-o", out, qo])
    assert open(out).read() == "".join(r.raw for r in rq if r.name in set(rnames))
    open(lst, "w").write("".join(n + "\n" for n in pick))
    cli.main(["extract", "-l", lst, "--sequential-read", "-o", out, po])
    assert open(out).read() == "".join(s.raw for s in rf if s.name in set(pick))
