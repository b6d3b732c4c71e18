"""-m gpu: the pyfastx-shaped object API (Fasta / Sequence / Fastq / Read) over
the HIP path, checked against golden vectors dumped from the real reference
(tests/golden/make_golden.py).  Reads like the reference's own tests
(tests/test_fasta.py, test_sequence.py, test_fastq.py, test_read.py)."""
import os
import shutil
import sqlite3

import numpy as np
import pytest

from conftest import DATA, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fx():
    import pyfastx_amd
    from pyfastx_amd import _lib
    assert _lib.lib().fx_device_count() >= 1
    return pyfastx_amd


@pytest.fixture()
def files(tmp_path):
    out = {}
    for fn in ("test.fa", "test.fa.gz", "test.fq", "test.fq.gz"):
        p = tmp_path / fn
        shutil.copy(os.path.join(DATA, fn), p)
        os.chmod(p, 0o644)
        out[fn] = str(p)
    return out


@pytest.mark.parametrize("fn", ["test.fa", "test.fa.gz"])
def test_fasta_index_file_matches_reference(fx, files, fn):
    g = load_golden("fasta_fixture")[fn]
    fa = fx.Fasta(files[fn], full_index=True)
    assert len(fa) == g["count"] and fa.size == g["size"]
    assert fa.is_gzip == fn.endswith(".gz") == fx.gzip_check(files[fn])
    db = sqlite3.connect(files[fn] + ".fxi")            # the .fxi itself, row for row
    assert [list(r) for r in db.execute("SELECT * FROM seq")] == g["seq"]
    assert list(db.execute("SELECT seqnum,seqlen FROM stat").fetchone()) == g["stat"]
    assert [list(r[1:]) for r in db.execute("SELECT * FROM comp")] == g["comp"]
    db.close()
    assert fa.composition == g["composition"]
    assert fa.gc_content == g["gc_content"]
    assert repr(fa) == "<Fasta> %s contains %d sequences" % (files[fn], g["count"])
    assert fa.type == "DNA"
    # reopening loads the existing index instead of rebuilding
    fb = fx.Fasta(files[fn])
    assert len(fb) == g["count"] and fb[0].seq == fa[0].seq


@pytest.mark.parametrize("fn", ["test.fa", "test.fa.gz"])
def test_sequence_getters(fx, files, fn):
    g = load_golden("fasta_fixture")[fn]
    fa = fx.Fasta(files[fn])
    for rid, rec in list(g["records"].items())[::7]:
        s = fa[int(rid) - 1]
        assert s.id == int(rid) and s.seq == rec["seq"] and str(s) == rec["seq"] and len(s) == len(rec["seq"])
        assert s.description == rec["desc"] and s.raw == rec["raw"]
        assert fa[s.name].seq == rec["seq"] and s.name in fa
        assert repr(s) == "<Sequence> %s with length of %d" % (s.name, len(s))
    for f in g["fetches"]:
        sub = fa[f["id"] - 1][f["start"]:f["stop"]]
        assert sub.seq == f["seq"] and sub.antisense == f["antisense"]
        assert sub.complement == f["complement"] and sub.reverse == f["reverse"]
        assert sub.name == f["name"] and (sub.start, sub.end) == (f["start"] + 1, f["stop"])
        if f["raw"] is not None:
            assert sub.raw == f["raw"]
    s = fa[3]
    full = s.seq
    assert s[5:10].seq == full[5:10] and s[20:].seq == full[20:] and s[-10:].seq == full[-10:]
    assert s[10:100][:20].seq == full[10:30]            # nested slices: absolute coordinates
    assert s[7] == full[7] and s[-1] == full[-1]
    assert "".join(s) == full                           # line iteration
    assert full[10:25] in s and s.search(full[10:25]) == full.find(full[10:25]) + 1


@pytest.mark.parametrize("fn", ["test.fa", "test.fa.gz"])
def test_fetch_and_flank(fx, files, fn):
    g = load_golden("fasta_fixture")[fn]
    fa = fx.Fasta(files[fn])
    for f in g["fetch"]:
        iv = tuple(f["intervals"][0]) if len(f["intervals"]) == 1 else [tuple(x) for x in f["intervals"]]
        assert fa.fetch(f["name"], iv, strand=f["strand"]) == f["seq"]
    for f in g["flank"]:
        assert list(fa.flank(f["name"], f["start"], f["end"], flank_length=f["flank"], use_cache=f["use_cache"])) == f["out"]


def test_fasta_statistics(fx, files):
    fa = fx.Fasta(files["test.fa"])
    lens = sorted((len(s) for s in fa), reverse=True)
    half, acc = sum(lens) / 2, 0
    for l50, n50 in enumerate(lens, 1):
        acc += n50
        if acc >= half:
            break
    assert fa.nl(50) == (n50, l50) == (516, 66)
    assert round(fa.mean, 3) == round(sum(lens) / len(lens), 3)
    assert fa.median == sorted(lens)[105] == 386.0
    assert fa.count(200) == sum(1 for x in lens if x >= 200)
    assert len(fa.longest) == max(lens) and len(fa.shortest) == min(lens)
    assert list(fa.keys())[:2] == [fa[0].name, fa[1].name]


def test_fasta_options(fx, files, tmp_path):
    up = fx.Fasta(files["test.fa"], uppercase=True)
    assert up[0].seq == fx.Fasta(files["test.fa"])[0].seq.upper()
    os.unlink(files["test.fa"] + ".fxi")
    kf = fx.Fasta(files["test.fa"], key_func=lambda x: x.split()[1])
    assert kf[5].name == kf[5].description.split()[1]
    os.unlink(files["test.fa"] + ".fxi")
    fn = fx.Fasta(files["test.fa"], full_name=True, memory_index=True)
    assert fn[0].name == fn[0].description and not os.path.exists(files["test.fa"] + ".fxi")
    tup = list(fx.Fasta(files["test.fa.gz"], build_index=False))
    ref = fx.Fasta(files["test.fa.gz"])
    assert [n for n, _ in tup] == list(ref.keys()) and tup[10][1] == ref[10].seq
    assert repr(fx.Fasta(files["test.fa"], build_index=False)) == "<Fasta> %s" % files["test.fa"]


def test_fasta_edge_inputs_via_api(fx, tmp_path):
    g = load_golden("fasta_edge")
    for name, case in g.items():
        up = name.endswith(":upper")
        p = tmp_path / (name.replace(":", "_") + ".fa")
        p.write_bytes(case["text"].encode())
        fa = fx.Fasta(str(p), uppercase=up, full_index=True)
        db = sqlite3.connect(str(p) + ".fxi")
        assert [list(r) for r in db.execute("SELECT * FROM seq")] == case["seq"], name
        assert [list(r[1:]) for r in db.execute("SELECT * FROM comp")] == case["comp"], name
        db.close()
        for rid, rec in case["records"].items():
            assert fa[int(rid) - 1].seq == rec["seq"], (name, rid)
            if "desc" in rec:
                assert fa[int(rid) - 1].description == rec["desc"] and fa[int(rid) - 1].raw == rec["raw"], (name, rid)
        for f in case["fetches"]:
            sub = fa[f["id"] - 1][f["start"]:f["stop"]]
            assert sub.seq == f["seq"] and sub.antisense == f["antisense"], (name, f)


def test_fasta_exceptions(fx, files, tmp_path):
    fa = fx.Fasta(files["test.fa"])
    with pytest.raises(TypeError):
        fx.Fasta(files["test.fa"], key_func=1)
    with pytest.raises(FileExistsError):
        fx.Fasta("a_file_not_exists")
    with pytest.raises(ValueError):
        fa.fetch("seq1", {"a": 1})
    with pytest.raises(NameError):
        fa.fetch("seq1", (1, 10))
    with pytest.raises(ValueError):
        fa.fetch(fa[0].name, (1, 10, 20))
    with pytest.raises(ValueError):
        fa.fetch(fa[0].name, (20, 10))
    with pytest.raises(IndexError):
        fa[len(fa)]
    with pytest.raises(KeyError):
        fa[list()]
    with pytest.raises(KeyError):
        fa["no_such_sequence"]
    with pytest.raises(ValueError):
        fa.nl(101)
    bad = tmp_path / "non.fa"
    bad.write_text("abc")
    with pytest.raises(RuntimeError):
        fx.Fasta(str(bad))
    with pytest.raises(ValueError):
        fa[0][::2]
    with pytest.raises(RuntimeError):
        iter(fa[0][2:9]).__next__()


@pytest.mark.parametrize("fn", ["test.fq", "test.fq.gz"])
def test_fastq_and_reads(fx, files, fn):
    g = load_golden("fastq_fixture")[fn]
    fq = fx.Fastq(files[fn], full_index=True)
    assert len(fq) == g["count"] and fq.size == g["size"] and fq.avglen == g["stat"][2]
    db = sqlite3.connect(files[fn] + ".fxi")
    assert [list(r) for r in db.execute("SELECT * FROM read")] == g["read"]
    assert list(db.execute("SELECT * FROM base").fetchone()) == g["base"]
    assert list(db.execute("SELECT * FROM meta").fetchone()) == g["meta"]
    db.close()
    assert fq.phred == g["phred"] == 33 and (fq.minqual, fq.maxqual) == (35, 70)
    assert (fq.minlen, fq.maxlen) == (150, 150)
    assert "Sanger Phred+33" in fq.encoding_type
    assert repr(fq) == "<Fastq> %s contains %d reads" % (files[fn], g["count"])
    for r in g["reads"]:
        rd = fq[r["i"]]
        assert (rd.name, rd.seq, rd.qual, rd.quali) == (r["name"], r["seq"], r["qual"], r["quali"])
        assert (rd.antisense, rd.complement, rd.reverse) == (r["antisense"], r["complement"], r["reverse"])
        assert rd.raw == r["raw"] and rd.description == r["desc"] and len(rd) == len(r["seq"])
        assert fq[rd.name].id == rd.id and rd.name in fq
    # the object that BUILT the index answers fq[i] from the table it wrote the index from (csrc/fxobj.c: _core_table); one that
    # loads the index file answers from the file's statements -- the same reads either way
    fq2 = fx.Fastq(files[fn])
    assert fq._core_table_rows == len(fq) == fq._core_names_rows and fq2._core_table_rows == 0 == fq2._core_names_rows
    for r in g["reads"][:40]:
        a, b = fq[r["i"]], fq2[r["i"]]
        assert (a.id, a.name, a.seq, a.qual, a.raw, a.description, repr(a)) == (b.id, b.name, b.seq, b.qual, b.raw, b.description, repr(b))
        a, b = fq[r["name"]], fq2[r["name"]]                    # by name: the builder's hash of the packed names / the file's statement
        assert (a.id, a.name, a.seq, a.qual, a.raw, a.description) == (b.id, b.name, b.seq, b.qual, b.raw, b.description) and a.id == r["i"] + 1
    assert fq[-1].name == fq2[-1].name == fq2[len(fq) - 1].name
    out = fq.fetch_many([r["i"] for r in g["reads"][:50]])
    for j, r in enumerate(g["reads"][:50]):
        a, b = out["offsets"][j], out["offsets"][j + 1]
        assert out["seq"][a:b].tobytes().decode() == r["seq"] and out["quali"][a:b].tolist() == r["quali"]
    with pytest.raises(IndexError):
        fq[len(fq)]
    with pytest.raises(KeyError):
        fq["nope"]
    tup = list(fx.Fastq(files[fn], build_index=False, index_file=files[fn] + ".none"))
    assert len(tup) == g["count"] and tup[0][1] == fq[0].seq and tup[0][2] == fq[0].qual and tup[0][0] == fq[0].name


def test_module_functions(fx):
    assert fx.reverse_complement("ATGC") == "GCAT"
    assert fx.version() == "2.3.1"
    for s, want in load_golden("misc")["reverse_complement"]:
        assert fx.reverse_complement(s) == want


def test_batched_fetch_many(fx, files):
    g = load_golden("fasta_fixture")["test.fa"]
    fa = fx.Fasta(files["test.fa"])
    f = g["fetches"]
    buf, offs = fa.fetch_many([x["id"] - 1 for x in f], [x["start"] for x in f], [x["stop"] for x in f],
                              strand=["-" if i % 2 else "+" for i in range(len(f))])
    for i, x in enumerate(f):
        want = x["antisense"] if i % 2 else x["seq"]
        assert buf[offs[i]:offs[i + 1]].tobytes().decode() == want


def test_fetch_many_layout_by_the_library(fx, files):
    """Round 4: the batched calls leave the layout to the library (fx_fasta_fetch_alloc / fx_fastq_fetch_alloc: intervals
    checked on the device, offsets by a device scan, answers by DMA into pinned blocks of fx_pinned_alloc).  Same bytes as the
    caller-allocated entry points, the reference's exception classes, blocks back in the pool when the arrays die."""
    from pyfastx_amd import _lib
    L = _lib.lib()
    fa = fx.Fasta(files["test.fa"])
    n = len(fa)
    rng = np.random.default_rng(5)
    ids = rng.integers(0, n, 5000)
    slen = np.array([len(fa[int(i)]) for i in range(n)], dtype=np.int64)
    st = (rng.random(5000) * slen[ids]).astype(np.int64)
    sp = np.minimum(slen[ids], st + rng.integers(0, 200, 5000))
    strand = rng.integers(0, 2, 5000).astype(np.uint8)
    buf, offs = fa.fetch_many(ids, st, sp, strand=strand)
    assert L.fx_pinned_holds(buf.ctypes.data, buf.nbytes) == 1 and L.fx_pinned_holds(offs.ctypes.data, offs.nbytes) == 1
    rb, ro, _ = fa._st.blob.fasta_fetch(ids, st, sp, flags_per_query=np.where(strand != 0, 6, 0).astype(np.uint8))
    assert np.array_equal(ro, offs) and np.array_equal(rb, buf)
    names = [fa[int(i)].name for i in range(n)]
    b2, o2 = fa.fetch_many([names[i] for i in ids], st, sp, strand=strand)            # names: one C pass, most of them the same objects
    assert np.array_equal(b2, buf) and np.array_equal(o2, offs)
    b3, o3 = fa.fetch_many(tuple(str(names[i]) for i in ids[:100]), st[:100], sp[:100])
    assert np.array_equal(np.diff(o3), (sp - st)[:100]) and b3.size == int(o3[-1])
    addr = buf.ctypes.data
    view = buf[10:20]
    del buf, b2
    assert L.fx_pinned_holds(addr, 1) == 1                       # a view keeps the block
    del view
    assert L.fx_pinned_holds(addr, 1) == 0                       # ... the last one gives it back to the pool
    e0, o0 = fa.fetch_many([], [], [])
    assert e0.size == 0 and o0.tolist() == [0]
    with pytest.raises(IndexError):
        fa.fetch_many([0, n], [0, 0], [1, 1])
    with pytest.raises(ValueError):
        fa.fetch_many([0, 1], [0, 5], [1, 4])
    with pytest.raises(ValueError):
        fa.fetch_many([0], [0], [int(slen[0]) + 1])
    with pytest.raises(KeyError):
        fa.fetch_many([names[0], "no such sequence"], [0, 0], [1, 1])
    # FASTQ: by id, by name list, by pre-packed names
    fq = fx.Fastq(files["test.fq"])
    m = len(fq)
    rid = rng.integers(0, m, 3000)
    out = fq.fetch_many(rid)
    t = fq._tab_host
    rs, rq, ri, rof = fq._st.blob.fastq_fetch(rid, t["rlen"][rid], phred=fq._phred)
    assert np.array_equal(out["offsets"], rof) and np.array_equal(out["seq"], rs) and np.array_equal(out["qual"], rq) and np.array_equal(out["quali"], ri)
    only = fq.fetch_many(rid, want=("qual",))
    assert only["seq"] is None and only["quali"] is None and np.array_equal(only["qual"], rq)
    rnames = [fq[int(i)].name for i in rid[:300]]
    byname = fq.fetch_many(rnames)
    assert np.array_equal(byname["seq"], rs[:int(rof[300])])
    enc = [x.encode() for x in rnames]
    po = np.zeros(len(enc) + 1, dtype=np.int64)
    np.cumsum([len(x) for x in enc], out=po[1:])
    packed = fq.fetch_many((b"".join(enc), po))
    assert np.array_equal(packed["seq"], byname["seq"]) and np.array_equal(packed["offsets"], byname["offsets"])
    with pytest.raises(KeyError, match="no such read"):
        fq.fetch_many((b"no such read", np.array([0, 12], dtype=np.int64)))
    with pytest.raises(IndexError):
        fq.fetch_many([0, m])
    assert fq.fetch_many([])["offsets"].tolist() == [0]
    # query arrays that already live in pinned memory go up without a staging copy: same answers
    pid = _lib.pinned_empty(rid.size, np.int64)
    pid[:] = rid
    again = fq.fetch_many(pid)
    assert np.array_equal(again["seq"], rs)


def test_single_getters_page_cache_and_resident_kernel_agree(fx, files, monkeypatch):
    """Round 4: on a plain file a single small getter is answered from the page cache by the C object layer (csrc/fxobj.c);
    with FX_NO_HOST_GETTERS=1 (and for every gzip input) the same getter goes to the resident kernel (fx_fetch_one).  Same
    strings either way, and both equal the goldens' (which come from the compiled reference)."""
    g = load_golden("fasta_fixture")["test.fa"]
    fa = fx.Fasta(files["test.fa"])
    fa[0][0:5].seq
    assert fa._core_fd >= 0 and fa._core_handle != 0
    monkeypatch.setenv("FX_NO_HOST_GETTERS", "1")
    fb = fx.Fasta(files["test.fa"])
    fb[0][0:5].seq
    assert fb._core_fd == -1 and fb._core_handle != 0
    for x in g["fetches"]:
        a, b = fa[x["id"] - 1][x["start"]:x["stop"]], fb[x["id"] - 1][x["start"]:x["stop"]]
        assert a.seq == b.seq == x["seq"] and a.antisense == b.antisense == x["antisense"]
        assert a.reverse == b.reverse and a.complement == b.complement
    gq = load_golden("fastq_fixture")["test.fq"]
    fq = fx.Fastq(files["test.fq"])
    monkeypatch.delenv("FX_NO_HOST_GETTERS")
    fr = fx.Fastq(files["test.fq"])
    fr[0].seq, fq[0].seq
    assert fr._core_fd >= 0 and fq._core_fd == -1
    for r in gq["reads"]:
        a, b = fr[r["i"]], fq[r["i"]]
        assert (a.seq, a.qual, a.quali) == (b.seq, b.qual, b.quali) == (r["seq"], r["qual"], r["quali"])
        assert fr[r["name"]].id == a.id == fq[r["name"]].id
    monkeypatch.setenv("FX_NO_C_SUBSCRIPT", "1")                # ... and the subscript through the sqlite3 module gives the same Reads
    fs = fx.Fastq(files["test.fq"])
    for r in gq["reads"][:20]:
        a = fs[r["i"]]
        assert (a.name, a.seq, a.id) == (r["name"], r["seq"], fr[r["i"]].id)
    fa._st.blob.close()                                          # a closed blob takes the C getters' handle with it
    assert fa._core_handle == 0 and fa._core_fd == -1


def test_fetch_many_on_loaded_index(fx, files):
    """An index that already exists on disk is LOADED (no scan, index.c:391-429); fetch_many installs its rows
    in HBM (fx_fasta_set_table) and answers by (id, start, stop) exactly like a freshly built one."""
    g = load_golden("fasta_fixture")["test.fa"]
    fx.Fasta(files["test.fa"])                       # builds and writes test.fa.fxi
    fa = fx.Fasta(files["test.fa"])                  # loads it
    f = g["fetches"]
    names = [fa[x["id"] - 1].name for x in f]
    buf, offs = fa.fetch_many(names, [x["start"] for x in f], [x["stop"] for x in f],
                              strand=["-" if i % 3 == 0 else "+" for i in range(len(f))])
    for i, x in enumerate(f):
        want = x["antisense"] if i % 3 == 0 else x["seq"]
        assert buf[offs[i]:offs[i + 1]].tobytes().decode() == want


def test_fastq_fetch_many_by_name(fx, files):
    g = load_golden("fastq_fixture")["test.fq"]
    fq = fx.Fastq(files["test.fq"])
    reads = g["reads"][:40]
    names = [fq[r["i"]].name for r in reads]
    assert fq.ids_of(names + ["no such read"]).tolist() == [r["i"] for r in reads] + [-1]
    out = fq.fetch_many(names)
    offs = out["offsets"]
    for j, r in enumerate(reads):
        assert out["seq"][offs[j]:offs[j + 1]].tobytes().decode() == r["seq"]
        assert out["qual"][offs[j]:offs[j + 1]].tobytes().decode() == r["qual"]
    with pytest.raises(KeyError):
        fq.fetch_many(["no such read"])


def _idx(path):
    db = sqlite3.connect(path)
    ok = db.execute("PRAGMA integrity_check").fetchall()
    idx = sorted(r[0] for r in db.execute("SELECT name FROM sqlite_master WHERE type='index'"))
    db.close()
    return ok, idx


def test_bulk_written_index_is_sound(fx, files, tmp_path):
    """A new .fxi is written as b-tree pages (fx_fxi_bulk_rows / fx_names_sort / fx_fxi_bulk_index): SQLite's own
    integrity_check (page structure, index entries vs rows, ordering, uniqueness) accepts it, by-name access goes
    through the index, later INSERTs (composition) land on it, and the special cases fall back as the reference does."""
    import numpy as np
    g = load_golden("fasta_fixture")["test.fa"]
    fa = fx.Fasta(files["test.fa"], full_index=True)        # comp rows INSERTed on top of the bulk-written file
    assert _idx(files["test.fa"] + ".fxi") == ([("ok",)], ["chromidx", "seqidx"])
    db = sqlite3.connect(files["test.fa"] + ".fxi")
    plan = db.execute("EXPLAIN QUERY PLAN SELECT * FROM seq WHERE chrom=?", ("x",)).fetchall()
    assert "chromidx" in plan[0][-1]
    for row in g["seq"][::9]:
        assert db.execute("SELECT ID FROM seq WHERE chrom=?", (row[1],)).fetchone()[0] == row[0]
    db.close()
    fq = fx.Fastq(files["test.fq"], full_index=True)
    assert _idx(files["test.fq"] + ".fxi") == ([("ok",)], ["readidx"])
    # many records, shuffled numeric suffixes: several leaf and interior pages in both b-trees
    rng = np.random.default_rng(5)
    n = 120000
    ids = rng.permutation(n).tolist()
    p = tmp_path / "many.fq"
    p.write_bytes(b"".join(b"@SRR8539271.%d len=%d\nACGTNACGTA\n+\nIIIIIHHHHH\n" % (i + 1, i % 97) for i in ids))
    fq = fx.Fastq(str(p))
    assert len(fq) == n and _idx(str(p) + ".fxi") == ([("ok",)], ["readidx"])
    for j in rng.integers(0, n, 30).tolist():
        r = fq["SRR8539271.%d" % (ids[j] + 1)]
        assert r.id == j + 1 and r.seq == "ACGTNACGTA"
    db = sqlite3.connect(str(p) + ".fxi")
    assert db.execute("SELECT count(*), min(ID), max(ID) FROM read").fetchone() == (n, 1, n)
    want = sorted(b"SRR8539271.%d" % (i + 1) for i in ids)
    got = [r[0].encode() for r in db.execute("SELECT name FROM read INDEXED BY readidx ORDER BY name")]
    assert got == want
    db.close()
    # duplicate names: the reference's CREATE UNIQUE INDEX fails and is ignored (index.c:363-366) -> no index
    p = tmp_path / "dup.fa"
    p.write_bytes(b">a 1\nACGT\n>b\nGG\n>a 2\nTTTT\n")
    fa = fx.Fasta(str(p))
    assert _idx(str(p) + ".fxi") == ([("ok",)], []) and len(fa) == 3 and fa["a"].id == 1
    # a name too long for an in-page index entry (REINDEX by SQLite), one too long for a table page (INSERT path)
    for L, tag in ((1500, "mid"), (5000, "big")):
        p = tmp_path / ("long_%s.fa" % tag)
        p.write_bytes(b">" + b"N" * L + b"\nACGT\n>short\nGGCC\n")
        fa = fx.Fasta(str(p))
        assert _idx(str(p) + ".fxi") == ([("ok",)], ["chromidx"])
        assert fa["N" * L].seq == "ACGT" and fa["short"].id == 2
    # non-ASCII header bytes are stored as they are in the file (sqlite3_bind_text of the raw bytes, index.c:239-251)
    p = tmp_path / "latin.fa"
    p.write_bytes(b">caf\xc3\xa9 x\nACGT\n>plain\nGG\n")
    fa = fx.Fasta(str(p))
    db = sqlite3.connect(str(p) + ".fxi")
    assert db.execute("SELECT CAST(chrom AS BLOB) FROM seq WHERE ID=1").fetchone()[0] == b"caf\xc3\xa9"
    db.close()
    assert fa["café"].seq == "ACGT"


def test_bulk_written_comp_table(fx, tmp_path, monkeypatch):
    """full_index on a file with many records: comp + seqidx are bulk-loaded from the sparse GPU composition; the
    table equals the one the INSERT path writes for the same file (threshold forced down / up)."""
    import numpy as np
    from pyfastx_amd import api
    rng = np.random.default_rng(2)
    parts = []
    for i in range(3000):
        parts.append(b">s%d d\n" % i)
        s = np.frombuffer(b"ACGTNacgt", dtype=np.uint8)[rng.integers(0, 9, int(rng.integers(0, 300)))].tobytes()
        parts += [s[p:p + 70] + b"\n" for p in range(0, len(s), 70)]
    tables = []
    for tag, thr in (("bulk", 1), ("ins", 10**9)):
        p = tmp_path / ("many_%s.fa" % tag)
        p.write_bytes(b"".join(parts))
        monkeypatch.setattr(api, "_COMP_BULK_MIN", thr)
        fa = fx.Fasta(str(p))
        keys = fa.keys()                                     # taken before the index file is re-opened by the bulk load
        assert fa.composition == fx.Fasta(str(p)).composition          # second object: loads the table from the file
        assert keys[3] == "s3" and "s7" in keys and len(keys) == 3000
        db = sqlite3.connect(str(p) + ".fxi")
        assert db.execute("PRAGMA integrity_check").fetchall() == [("ok",)]
        tables.append((db.execute("SELECT * FROM comp ORDER BY ID").fetchall(),
                       sorted(r[0] for r in db.execute("SELECT name FROM sqlite_master WHERE type='index'"))))
        db.close()
    assert tables[0] == tables[1] and tables[0][1] == ["chromidx", "seqidx"]


def test_iteration_rides_on_batched_fetches(fx, files, tmp_path, oracle):
    """`for s in fa` / `for r in fq` over an indexed file hand out objects whose sequence (and quality)
    came off the GPU in one gather per batch; what they return is what the one-by-one getters return, and what the
    oracle says -- also for a record with ONE odd line (norm = 1 all the same, index.c:342), whose whole sequence is the
    whole record despaced (sequence.c:76-98), not the line arithmetic."""
    from conftest import fixture_bytes
    raw = fixture_bytes("test.fa")
    recs, _ = oracle.fasta_index(raw)
    fa = fx.Fasta(files["test.fa"])
    seen = 0
    for i, s in enumerate(fa):
        want = oracle.fetch(raw, recs[i]["boff"], recs[i]["blen"], recs[i]["slen"]).decode()
        assert s.seq == want == fa[i].seq and s.id == i + 1 and len(s) == len(want)
        if i % 37 == 0:
            assert s.antisense == oracle.fetch(raw, recs[i]["boff"], recs[i]["blen"], recs[i]["slen"], 6).decode()
            assert s[3:20].seq == want[3:20]
        seen += 1
    assert seen == len(recs)
    up = fx.Fasta(files["test.fa"], uppercase=True)
    assert [s.seq for s in up][:5] == [fa[i].seq.upper() for i in range(5)]
    # the rows of an index file are stepped from C (_fxobj.RowCursor) and the objects of a batch made by one call; a memory
    # index has no file to open a second connection to: the sqlite3 module's rows give the same objects
    mem = fx.Fasta(files["test.fa"], index_file=files["test.fa"] + ".unused", memory_index=True)
    assert [(s.id, s.name, s.seq, s.start, s.end, len(s)) for s in mem] == [(s.id, s.name, s.seq, s.start, s.end, len(s)) for s in fa]
    # one odd line in the middle, one long last line, an empty record, a record with no trailing newline
    odd = tmp_path / "odd.fa"
    text = b">a\nACGTACGT\nAC\nGGGGTTTT\nCCCCAAAA\n>b\nACGT\nACGTACGTAC\n>e\n>c\nTTTTGGGG\nTTTTGG"
    odd.write_bytes(text)
    fo = fx.Fasta(str(odd))
    r2, _ = oracle.fasta_index(text)
    assert [int(x) for x in r2["norm"]] == [1, 1, 1, 1]
    want = [oracle.fetch(text, r["boff"], r["blen"], r["slen"]).decode() for r in r2]
    assert want[0] == "ACGTACGTACGGGGTTTTCCCCAAAA"
    assert [s.seq for s in fo] == want == [fo[i].seq for i in range(4)]
    assert fo[0].antisense == oracle.fetch(text, r2[0]["boff"], r2[0]["blen"], r2[0]["slen"], 6).decode()
    assert str(fo[0]) == want[0] and fo[1].reverse == want[1][::-1]
    # ... and their slices are slices of that, as the reference's fetch() and warm-cache slices are (fasta.c:440-461):
    # the record is recognised as not line-regular from its last line, once
    for k in (0, 1, 3):
        w = want[k]
        for a, b in ((0, len(w)), (2, 20), (9, 11), (len(w) - 3, len(w)), (5, 5)):
            b = min(b, len(w))
            if a <= b:
                assert fo[k][a:b].seq == w[a:b], (k, a, b)
        assert fo.fetch("abec"[k], (3, min(12, len(w)))) == w[2:min(12, len(w))]
        assert fo.fetch("abec"[k], [(1, 4), (6, 9)], strand="-") == oracle.revcomp((w[0:4] + w[5:9]).encode(), 3).decode()
    reg = fo._st.blob.fasta_line_regular(4).tolist()          # the device column every fetch path goes by
    assert [reg[0], reg[1], reg[3]] == [0, 0, 1]
    buf, offs = fo.fetch_many([0, 1, 3, 0], [2, 9, 0, 5], [20, 11, 14, 5])       # ... the batched path included
    assert [buf[offs[i]:offs[i + 1]].tobytes().decode() for i in range(4)] == [want[0][2:20], want[1][9:11], want[3][0:14], ""]
    # FASTQ
    rawq = fixture_bytes("test.fq")
    rq, size, ln = oracle.fastq_index(rawq)
    fq = fx.Fastq(files["test.fq"])
    n = 0
    for i, r in enumerate(fq):
        so, qo, m = int(rq["soff"][i]), int(rq["qoff"][i]), int(rq["rlen"][i])
        assert r.seq == rawq[so:so + m].decode() == fq[i].seq and r.qual == rawq[qo:qo + m].decode() == fq[i].qual
        assert r.id == i + 1 and len(r) == m
        if i % 97 == 0:
            assert r.quali == [c - 33 for c in rawq[qo:qo + m]] and r.antisense == oracle.revcomp(rawq[so:so + m], 3).decode()
        n += 1
    assert n == len(rq)
    # the rows come from a cursor stepped in C (_fxobj.RowCursor); when it cannot have its connection -- here: another
    # connection holds the index file exclusively -- the sqlite3 module's rows give the same objects
    import sqlite3
    first = [(r.id, r.name, r.seq, r.qual, r.description, len(r)) for r in fq]
    lock = sqlite3.connect(files["test.fq"] + ".fxi", isolation_level=None)
    lock.execute("PRAGMA locking_mode=EXCLUSIVE")
    lock.execute("BEGIN EXCLUSIVE")
    try:
        from pyfastx_amd import _fxobj
        with pytest.raises(RuntimeError):
            _fxobj.RowCursor(files["test.fq"] + ".fxi", "SELECT ID, name, dlen, rlen, soff, qoff FROM read ORDER BY ID").fetch(10)
    finally:
        lock.execute("COMMIT")
        lock.close()
    real = _fxobj.RowCursor
    try:
        def refuse(*a):
            raise RuntimeError("no connection")
        _fxobj.RowCursor = refuse
        assert [(r.id, r.name, r.seq, r.qual, r.description, len(r)) for r in fq] == first
    finally:
        _fxobj.RowCursor = real


def test_fetch_many_equals_slices_on_odd_line_records(fx, tmp_path):
    """the batched fetch_many and the per-object slices return the same bytes for records with one odd line
    (norm = 1, not line-regular) -- both cut the despaced record; also on an index loaded from the file."""
    from test_gpu_kernels import _odd_line_fasta
    rng = np.random.default_rng(5)
    raw = _odd_line_fasta(rng, False)
    p = str(tmp_path / "odd.fa")
    open(p, "wb").write(raw)
    for reopen in (False, True):
        fa = fx.Fasta(p)
        n = len(fa)
        ids = rng.integers(0, n, 300)
        lens = np.array([len(fa[int(i)]) for i in ids])
        st = (rng.random(300) * lens).astype(np.int64)
        sp = np.minimum(st + rng.integers(1, 200, 300), lens)
        strand = rng.integers(0, 2, 300)
        buf, offs = fa.fetch_many(ids, st, sp, strand=strand)
        for j in range(300):
            s = fa[int(ids[j])][int(st[j]):int(sp[j])]
            want = s.antisense if strand[j] else s.seq
            whole = fa[int(ids[j])].seq[int(st[j]):int(sp[j])]
            assert buf[offs[j]:offs[j + 1]].tobytes().decode() == want, (reopen, j)
            if not strand[j]:
                assert want == whole
        del fa


@pytest.mark.parametrize("kind", ["plain", "crlf", "bgzf"])
def test_fasta_over_several_devices(fx, tmp_path, oracle, kind):
    """Fasta(path, devices=[...]) (SURVEY 8e in one process): byte-range shards -- here three logical ones on the one GPU of
    the test box -- each staged with fx_open_file_range and scanned on its own, stitched, ONE .fxi: the same index file,
    the same composition, the same answers as the single-device build; records and queries that cross the cuts included."""
    from test_gpu_kernels import _rand_fasta, _odd_line_fasta
    from pyfastx_amd import synth
    rng = np.random.default_rng(17)
    raw = _rand_fasta(rng, 25, 70, crlf=(kind == "crlf"), ragged=False, trailing=True, lower=True)
    raw += _odd_line_fasta(rng, kind == "crlf") + _rand_fasta(rng, 3, 40, crlf=(kind == "crlf"), ragged=True, trailing=False)
    one, many = str(tmp_path / "one.fa"), str(tmp_path / "many.fa")
    if kind == "bgzf":
        one, many = one + ".gz", many + ".gz"
        data = synth.bgzf_compress(raw, block=3000)         # small members: every shard covers a handful of them
    else:
        data = raw
    for p in (one, many):
        open(p, "wb").write(data)
    a = fx.Fasta(one, full_index=True)
    for devs in ([0, 0, 0], [0] * 7):
        if os.path.exists(many + ".fxi"):
            os.unlink(many + ".fxi")
        b = fx.Fasta(many, full_index=True, devices=devs)
        ta, tb = (sqlite3.connect(p + ".fxi") for p in (one, many))
        for tab in ("seq", "comp", "gzindex"):
            assert ta.execute("SELECT * FROM %s" % tab).fetchall() == tb.execute("SELECT * FROM %s" % tab).fetchall(), (tab, devs)
        assert ta.execute("SELECT seqnum,seqlen FROM stat").fetchall() == tb.execute("SELECT seqnum,seqlen FROM stat").fetchall()
        n = len(a)
        assert len(b) == n and b.size == a.size
        ids = rng.integers(0, n, 500)
        lens = np.array([len(a[int(i)]) for i in ids])
        st = (rng.random(500) * lens).astype(np.int64)
        sp = np.minimum(st + rng.integers(0, 400, 500), lens)
        strand = rng.integers(0, 2, 500)
        ba, oa = a.fetch_many(ids, st, sp, strand=strand)
        bb, ob = b.fetch_many(ids, st, sp, strand=strand)
        assert (oa == ob).all() and ba.tobytes() == bb.tobytes()
        for i in range(0, n, 3):                              # per-object getters read through the shards
            assert b[i].seq == a[i].seq and b[i].name == a[i].name and b[i].description == a[i].description
            if len(a[i]) > 12:
                assert b[i][3:11].antisense == a[i][3:11].antisense and b[i].raw == a[i].raw
        assert [s.name for s in b][:10] == [s.name for s in a][:10]
        del b


@pytest.mark.parametrize("kind", ["lf", "crlf", "bgzf"])
def test_fastq_over_several_devices(fx, tmp_path, oracle, kind):
    """Fastq(path, devices=[...]) (round 4, SURVEY 8e in one process): byte-range shards -- logical ones on the one GPU of the
    test box -- staged and scanned at the same time, the line numbering from the shards' counts, ONE .fxi: the same index file,
    base / meta and answers as the single-device build; reads that cross the cuts included."""
    from test_gpu_shards import _rand_fastq
    from pyfastx_amd import synth
    rng = np.random.default_rng(23)
    raw = _rand_fastq(rng, 900, crlf=(kind == "crlf"), trailing=(kind != "crlf"))
    one, many = str(tmp_path / "one.fq"), str(tmp_path / "many.fq")
    if kind == "bgzf":
        one, many = one + ".gz", many + ".gz"
        data = synth.bgzf_compress(raw, block=3000)
    else:
        data = raw
    for p in (one, many):
        open(p, "wb").write(data)
    a = fx.Fastq(one, full_index=True)
    for devs in ([0, 0, 0], [0] * 7):
        if os.path.exists(many + ".fxi"):
            os.unlink(many + ".fxi")
        b = fx.Fastq(many, full_index=True, devices=devs)
        assert b._st.md is not None and b._st.md.windows == len(devs) and len(b._st.md.cache.lru) == len(devs)
        ta, tb = (sqlite3.connect(p + ".fxi") for p in (one, many))
        for tab in ("read", "base", "meta", "gzindex"):
            assert ta.execute("SELECT * FROM %s" % tab).fetchall() == tb.execute("SELECT * FROM %s" % tab).fetchall(), (tab, devs)
        assert ta.execute("SELECT counts, size FROM stat").fetchall() == tb.execute("SELECT counts, size FROM stat").fetchall()
        assert tb.execute("PRAGMA integrity_check").fetchone()[0] == "ok"
        n = len(a)
        assert len(b) == n and b.size == a.size and b.phred == a.phred and b.composition == a.composition
        ids = rng.integers(0, n, 600)
        ga, gb = a.fetch_many(ids), b.fetch_many(ids)
        for k in ("seq", "qual", "quali", "offsets"):
            assert np.array_equal(ga[k], gb[k]), (k, devs)
        for i in range(0, n, 37):
            ra, rb = a[i], b[i]
            assert (rb.name, rb.seq, rb.qual, rb.quali, rb.raw) == (ra.name, ra.seq, ra.qual, ra.quali, ra.raw)
            assert b[ra.name].id == ra.id and rb.antisense == ra.antisense
        assert [(r.name, r.seq, r.qual) for r in b] == [(r.name, r.seq, r.qual) for r in a]
        ra_, rb_ = a.raw_many(ids[:50]), b.raw_many(ids[:50])
        assert np.array_equal(ra_[1], rb_[1]) and ra_[0].tobytes() == rb_[0].tobytes()
        with pytest.raises(IndexError):
            b.fetch_many([n])
        del b

