"""CPU: host-side logic around the kernels -- .fxi writer, slice arithmetic,
error classes, shard stitch -- none of which needs a GPU."""
import os
import sqlite3

import numpy as np
import pytest

from conftest import DATA, fixture_bytes, load_golden
import shard_ref


def test_fxi_writer_reproduces_reference_file(oracle, tmp_path):
    """Feed the writer the (oracle) arrays -> the .fxi equals the reference's rows and schema."""
    from pyfastx_amd import fxi
    g = load_golden("fasta_fixture")["test.fa"]
    raw = fixture_bytes("test.fa")
    recs, tot = oracle.fasta_index(raw)
    names = [raw[r["name_off"]:r["name_off"] + r["name_len"]].decode() for r in recs]
    cols = {k: recs[k] for k in ("boff", "blen", "slen", "llen", "elen", "norm", "dlen")}
    p = str(tmp_path / "t.fxi")
    db = fxi.connect(p)
    fxi.write_fasta(db, names, cols, tot)
    fxi.write_fasta_comp(db, oracle.fasta_comp(raw, len(recs)))
    db.close()
    db = sqlite3.connect(p)
    assert [list(r) for r in db.execute("SELECT * FROM seq")] == g["seq"]
    assert [list(r[1:]) for r in db.execute("SELECT * FROM comp")] == g["comp"]
    assert db.execute("SELECT * FROM stat").fetchone() == (211, 86262, None, None, None, None)
    cols_of = lambda t: [r[1] for r in db.execute("PRAGMA table_info(%s)" % t)]
    assert cols_of("seq") == ["ID", "chrom", "boff", "blen", "slen", "llen", "elen", "norm", "dlen"]   # index.c:178-189
    assert cols_of("stat") == ["seqnum", "seqlen", "avglen", "medlen", "n50", "l50"]
    assert cols_of("comp") == ["ID", "seqid", "abc", "num"] and cols_of("gzindex") == ["ID", "content"]
    idx = {r[1] for r in db.execute("PRAGMA index_list(seq)")} | {r[1] for r in db.execute("PRAGMA index_list(comp)")}
    assert {"chromidx", "seqidx"} <= idx


def test_fxi_fastq_writer(oracle, tmp_path):
    from pyfastx_amd import fxi
    g = load_golden("fastq_fixture")["test.fq"]
    raw = fixture_bytes("test.fq")
    recs, size, ln = oracle.fastq_index(raw)
    names = [raw[r["name_off"]:r["name_off"] + r["name_len"]].decode() for r in recs]
    db = fxi.connect(str(tmp_path / "q.fxi"))
    fxi.write_fastq(db, names, {k: recs[k] for k in ("dlen", "rlen", "soff", "qoff")}, size)
    c = oracle.fastq_composition(raw)
    fxi.write_fastq_comp(db, [c[k] for k in "acgtn"], [c["maxlen"], c["minlen"], c["minqs"], c["maxqs"], c["phred"]])
    assert [list(r) for r in db.execute("SELECT * FROM read")] == g["read"]
    assert list(db.execute("SELECT * FROM stat").fetchone()) == g["stat"]
    assert list(db.execute("SELECT * FROM base").fetchone()) == g["base"]
    assert list(db.execute("SELECT * FROM meta").fetchone()) == g["meta"]
    assert [r[1] for r in db.execute("PRAGMA table_info(meta)")] == ["maxlen", "minlen", "minqs", "maxqs", "phred"]


def test_slice_arithmetic_matches_oracle(oracle):
    from pyfastx_amd.api import Sequence
    rng = np.random.default_rng(3)
    for _ in range(2000):
        llen = int(rng.integers(2, 200)); elen = int(rng.integers(1, 3))
        if llen - elen <= 0:
            continue
        slen = int(rng.integers(1, 5000)); boff = int(rng.integers(0, 10**9))
        s = Sequence(None, 1, "x", boff, slen * 2, slen, llen, elen, 1, 5)
        a = int(rng.integers(0, slen)); b = int(rng.integers(a, slen + 1))
        assert s._range(a, b) == oracle.slice_range(boff, llen, elen, a, b)


def test_error_classes_before_any_gpu_work(tmp_path):
    import pyfastx_amd as fx
    with pytest.raises(TypeError):
        fx.Fasta(os.path.join(DATA, "test.fa"), key_func=1)
    with pytest.raises(FileExistsError):
        fx.Fasta("a_file_not_exists")
    with pytest.raises(FileExistsError):
        fx.Fastq("a_file_not_exists")
    bad = tmp_path / "non.fa"
    bad.write_text("abc")
    with pytest.raises(RuntimeError):
        fx.Fasta(str(bad))
    with pytest.raises(RuntimeError):
        fx.Fastq(os.path.join(DATA, "test.fa"))
    assert fx.gzip_check(os.path.join(DATA, "test.fa.gz")) and not fx.gzip_check(os.path.join(DATA, "test.fa"))


def test_existing_index_is_loaded_without_gpu(oracle, tmp_path):
    """An .fxi next to the file is reused (index.c:418-429): metadata works with no device."""
    import pyfastx_amd as fx
    from pyfastx_amd import fxi
    import shutil
    p = str(tmp_path / "test.fa")
    shutil.copy(os.path.join(DATA, "test.fa"), p)
    raw = fixture_bytes("test.fa")
    recs, tot = oracle.fasta_index(raw)
    names = [raw[r["name_off"]:r["name_off"] + r["name_len"]].decode() for r in recs]
    db = fxi.connect(p + ".fxi")
    fxi.write_fasta(db, names, {k: recs[k] for k in ("boff", "blen", "slen", "llen", "elen", "norm", "dlen")}, tot)
    db.close()
    fa = fx.Fasta(p)
    assert len(fa) == 211 and fa.size == 86262 and fa.nl(50) == (516, 66) and fa.median == 386.0
    assert fa[0].name == "JZ822577.1" and len(fa[-1]) == 134 and "JZ822578.1" in fa
    assert fa[1][5:30].name == "JZ822578.1:6-30" and fa.count(200) > 0


def test_statistics_answer_from_a_read_only_index(oracle, tmp_path):
    """The reference ignores the result of the write that caches nl / mean / median in `stat` (fasta.c:651-659, 770-782,
    827-839), so an index the user cannot write still answers.  `PRAGMA query_only` makes every write fail the way a
    read-only file does (the tests run as root, for whom a chmod means nothing)."""
    import sqlite3
    import pyfastx_amd as fx
    from pyfastx_amd import fxi
    p = str(tmp_path / "ro.fa")
    raw = fixture_bytes("test.fa")
    open(p, "wb").write(raw)
    recs, tot = oracle.fasta_index(raw)
    names = [raw[r["name_off"]:r["name_off"] + r["name_len"]].decode() for r in recs]
    db = fxi.connect(p + ".fxi")
    fxi.write_fasta(db, names, {k: recs[k] for k in ("boff", "blen", "slen", "llen", "elen", "norm", "dlen")}, tot)
    db.close()
    fa = fx.Fasta(p)
    fa._db.execute("PRAGMA query_only=ON")
    with pytest.raises(sqlite3.Error):
        fa._db.execute("UPDATE stat SET medlen=1")
    assert fa.nl(50) == (516, 66) and fa.median == 386.0 and abs(fa.mean - 408.82464454976304) < 1e-9
    assert fa._db.execute("SELECT avglen, medlen, n50, l50 FROM stat").fetchone()[1:] in ((None, None, None), (0, 0, 0), (0.0, 0, 0))


def _reg_of(raw, r):
    """The line-regular bit of one oracle row: shard.line_regular_rule with the whole stream in view."""
    from pyfastx_amd.shard import line_regular_rule
    return int(bool(line_regular_rule(int(r["boff"]), int(r["blen"]), int(r["slen"]), int(r["llen"]), int(r["elen"]), int(r["norm"]),
                                      lambda p: raw[p] if 0 <= p < len(raw) else None)))


def _expect(oracle, raw, full_name=False):
    recs, _ = oracle.fasta_index(raw, full_name=full_name)
    out = {k: [int(x) for x in recs[k]] for k in ("hoff", "boff", "blen", "slen", "llen", "elen", "norm", "dlen", "name_len")}
    out["reg"] = [_reg_of(raw, r) for r in recs]            # the stitched rows carry it too (decided from the summaries there)
    return out


def test_stitch_every_cut_of_every_edge_case(oracle):
    for name, case in load_golden("fasta_edge").items():
        if name.endswith(":upper"):
            continue
        raw = case["text"].encode()
        if len(raw) > 600:
            continue
        want = _expect(oracle, raw)
        for c in range(1, len(raw)):
            assert shard_ref.stitched_rows(raw, [c]) == want, (name, c)
        for c in range(1, len(raw) - 2):
            assert shard_ref.stitched_rows(raw, [c, c + 1]) == want, (name, c)
            assert shard_ref.stitched_rows(raw, [c, c + 2], full_name=True) == _expect(oracle, raw, True), (name, c)


@pytest.mark.parametrize("seed", range(10))
def test_stitch_random(oracle, seed):
    rng = np.random.default_rng(seed)
    eol = b"\r\n" if seed & 1 else b"\n"
    parts = []
    for i in range(int(rng.integers(1, 12))):
        parts.append(b">r%d d%d" % (i, i) + eol)
        w = int(rng.integers(1, 30))
        s = bytes(rng.choice(list(b"ACGTN"), int(rng.integers(0, 400))).astype(np.uint8))
        odd = int(rng.integers(0, max(len(s) // w, 1))) if seed in (3, 6, 9) and i % 2 else -1     # ONE odd line somewhere: norm = 1, not regular
        for j, p in enumerate(range(0, len(s), w)):
            ww = w if seed % 3 else int(rng.integers(1, w + 1))
            if j == odd:
                ww = max(w - 1, 1)
            parts.append(s[p:p + ww] + eol)
    raw = b"".join(parts)
    if seed == 7:
        raw = raw.rstrip()
    want = _expect(oracle, raw)
    for g in (2, 3, 4, 8, 16):
        for _ in range(10):
            cuts = sorted(set(int(x) for x in rng.integers(1, len(raw), g - 1)))
            assert shard_ref.stitched_rows(raw, cuts) == want, (seed, cuts)


def test_stitch_header_lines_of_100_kib(oracle):
    """The host statement of the stitch (shard_ref: summaries computed in Python, shard.stitch_tail) with header lines of
    ~100 KiB across the cuts: the name may end any distance behind a cut."""
    name = ("N" + "x" * 69_999).encode()
    raw = b">a d\nACGTAC\nGT\n>" + name + b" " + b"d" * 30_000 + b"\nACGTACGT\nACGTACGT\nAC\n>" + b"y" * 100_000 + b"\nGG\n>z\nA\n"
    h1, h2 = raw.index(b">N"), raw.index(b">y")
    want, want_full = _expect(oracle, raw), _expect(oracle, raw, True)
    for c in (h1 + 1, h1 + 35_000, h1 + 70_000, h1 + 70_001, h1 + 70_002, h1 + 85_000, h1 + 100_002, h2 + 50_000, h2 + 100_001):
        assert shard_ref.stitched_rows(raw, [c]) == want, c
        assert shard_ref.stitched_rows(raw, [c], full_name=True) == want_full, c
    for a, b in ((h1 + 10, h1 + 66_000), (h1 + 69_000, h1 + 99_000), (h2 + 5, h2 + 99_999)):
        assert shard_ref.stitched_rows(raw, [a, (a + b) // 2, b]) == want, (a, b)


def test_reference_opens_our_gz_fxi(oracle, tmp_path):
    """The .fxi contract in the other direction: an index written by fxi.py for a gzip input --
    including BGZF-style gzindex points without windows -- is accepted by the REAL reference's
    loader (pyfastx_load_index + pyfastx_gzip_index_import, index.c:391-416, util.c:542-726).
    Runs only where oracle/_ref was built (needs /root/reference at build time)."""
    import glob
    import shutil
    import sys
    from conftest import ROOT
    if not glob.glob(os.path.join(ROOT, "oracle", "_ref", "pyfastx*.so")):
        pytest.skip("oracle/_ref not built here")
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    import pyfastx
    from pyfastx_amd import fxi, synth
    raw = fixture_bytes("test.fa")
    for name, payload, pts in (("plain.fa", raw, None),
                               ("bgzf.fa.gz", synth.bgzf_compress(raw, block=8000), "bgzf")):
        p = str(tmp_path / name)
        open(p, "wb").write(payload)
        recs, tot = oracle.fasta_index(raw)
        names = [raw[r["name_off"]:r["name_off"] + r["name_len"]].decode() for r in recs]
        db = fxi.connect(p + ".fxi")
        fxi.write_fasta(db, names, {k: recs[k] for k in ("boff", "blen", "slen", "llen", "elen", "norm", "dlen")}, tot)
        if pts:
            # the first deflate byte of every fourth member as restart points (bits 0, no window), as fx_gz_points reports
            # them: the reference's zran (here: the work-alike of oracle/refshim) starts a raw inflate there
            offs, uoffs, q, u = [], [], 0, 0
            while q < len(payload):
                bsize = payload[q + 16] | (payload[q + 17] << 8)
                isz = int.from_bytes(payload[q + bsize - 3:q + bsize + 1], "little")
                offs.append(q + 18); uoffs.append(u)
                q += bsize + 1; u += isz
            fxi.write_gzindex(db, len(payload), len(raw), offs[::4], uoffs[::4])
        db.close()
        fa = pyfastx.Fasta(p)                               # loads OUR index, builds nothing
        assert len(fa) == 211 and fa.size == 86262
        g = load_golden("fasta_fixture")["test.fa"]
        for f in g["fetches"][:40]:
            assert fa[f["id"] - 1][f["start"]:f["stop"]].seq == f["seq"]
        assert fa[0].name == "JZ822577.1" and fa.fetch("JZ822578.1", (1, 10)) == g["records"]["2"]["seq"][:10]
        del fa


def _fxi_rows(path):
    import sqlite3
    db = sqlite3.connect(path)
    ok = db.execute("PRAGMA integrity_check").fetchall()
    rows = db.execute("SELECT * FROM read ORDER BY ID").fetchall()
    stat = db.execute("SELECT * FROM stat").fetchall()
    idx = db.execute("SELECT name FROM sqlite_master WHERE type='index'").fetchall()
    db.close()
    return ok, rows, stat, idx


@pytest.mark.parametrize("n,maxname", [(0, 10), (1, 10), (37, 30), (5000, 60), (300000, 45), (2000, 1500), (30000, 900)])
def test_fxi_bulk_table_equals_inserts(tmp_path, n, maxname):
    """fx_fxi_bulk_rows (b-tree pages written directly, host code of libfxgpu.so) produces a database that SQLite
    reads exactly like the one made of INSERTs: same rows, integrity_check ok, unique index created on top."""
    from pyfastx_amd import fxi
    rng = np.random.default_rng(n + maxname)
    lens = rng.integers(0 if n < 100 else 1, maxname + 1, n)
    alpha = np.frombuffer(b"ACGT:_0123456789abcxyz", dtype=np.uint8)
    names = [alpha[rng.integers(0, alpha.size, int(L))].tobytes().decode() + ("_%d" % i) for i, L in enumerate(lens)]
    cols = {"dlen": rng.integers(1, 300, n), "rlen": rng.integers(0, 1 << 17, n),
            "soff": np.sort(rng.integers(0, 1 << 45, n)), "qoff": rng.integers(-5, 1 << 62, n)}
    cols = {k: v.astype(np.int64) for k, v in cols.items()}
    if n > 3:
        cols["rlen"][:4] = [0, 1, 127, 128]                 # serial types 8, 9, 1, 2
    a, b = str(tmp_path / "a.fxi"), str(tmp_path / "b.fxi")
    db = fxi.connect(a)
    fxi.write_fastq(db, names, cols, int(cols["rlen"].sum()))
    db.close()
    enc = [x.encode() for x in names]
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(x) for x in enc], out=offs[1:])
    packed = np.frombuffer(b"".join(enc), dtype=np.uint8)
    db = fxi.write_fastq_bulk(b, packed, offs, cols, int(cols["rlen"].sum()))
    db.close()
    # c: the UNIQUE INDEX b-tree written directly as well (fx_fxi_bulk_index) from the sorted order of the names
    c = str(tmp_path / "c.fxi")
    order = np.array(sorted(range(n), key=lambda i: enc[i]), dtype=np.int64)
    db = fxi.write_fastq_bulk(c, packed, offs, cols, int(cols["rlen"].sum()), order=order)
    db.close()
    import sqlite3
    ra = _fxi_rows(a)
    for other in (b, c):
        rb = _fxi_rows(other)
        assert rb[0] == [("ok",)], rb[0][:3]                # integrity_check also matches index entries with rows
        assert ra[1] == rb[1] and ra[2] == rb[2] and ra[3] == rb[3]
        db = sqlite3.connect(other)
        plan = db.execute("EXPLAIN QUERY PLAN SELECT ID FROM read WHERE name=?", ("x",)).fetchall()
        assert "readidx" in plan[0][-1]
        for i in rng.integers(0, n, min(n, 50)) if n else []:  # by-name and by-id probes through the index / rowid
            assert db.execute("SELECT ID FROM read WHERE name=?", (names[int(i)],)).fetchone()[0] == int(i) + 1
            assert db.execute("SELECT name FROM read WHERE ID=?", (int(i) + 1,)).fetchone()[0] == names[int(i)]
        assert db.execute("SELECT ID FROM read WHERE name=?", ("no such read",)).fetchone() is None
        got = [r[0] for r in db.execute("SELECT name FROM read INDEXED BY readidx ORDER BY name").fetchall()]
        assert got == [enc[int(i)].decode() for i in order]  # a full walk of the index b-tree
        if n:                                                # a written index keeps working under later changes
            db.execute("DELETE FROM read WHERE ID=?", (n // 2 + 1,))
            db.execute("INSERT INTO read VALUES (NULL,'zz new',1,2,3,4)")
            assert db.execute("PRAGMA integrity_check").fetchall() == [("ok",)]
        db.close()


def test_fxi_bulk_empty_names_and_fasta_table(tmp_path):
    """All names empty (n > 0, zero name bytes) and the FASTA flavour of the loader against write_fasta."""
    import sqlite3
    from pyfastx_amd import fxi
    n = 9
    cols = {k: (np.arange(n, dtype=np.int64) * (j + 3)) for j, k in enumerate(("boff", "blen", "slen", "llen", "elen", "norm", "dlen"))}
    for names in ([""] * n, ["chr%d" % (i * 7 % n) for i in range(n)]):
        a, b = str(tmp_path / "a.fxi"), str(tmp_path / "b.fxi")
        for p in (a, b):
            if os.path.exists(p):
                os.remove(p)
        db = fxi.connect(a)
        fxi.write_fasta(db, names, cols, 1234)
        db.close()
        enc = [x.encode() for x in names]
        offs = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(x) for x in enc], out=offs[1:])
        packed = np.frombuffer(b"".join(enc), dtype=np.uint8)
        distinct = len(set(names)) == n
        order = np.array(sorted(range(n), key=lambda i: enc[i]), dtype=np.int64) if distinct else None
        db = fxi.write_fasta_bulk(b, packed, offs, cols, 1234, order=order)
        db.close()
        for p in (a, b):
            db = sqlite3.connect(p)
            assert db.execute("PRAGMA integrity_check").fetchall() == [("ok",)]
            db.close()
        da, dbb = sqlite3.connect(a), sqlite3.connect(b)
        for q in ("SELECT * FROM seq ORDER BY ID", "SELECT seqnum, seqlen FROM stat", "SELECT name FROM sqlite_master WHERE type='index'"):
            assert da.execute(q).fetchall() == dbb.execute(q).fetchall(), q
        da.close(); dbb.close()


def test_fxi_bulk_row_too_large_falls_back(tmp_path):
    """A row that would need an overflow page: FX_ERANGE, and no half-written file is left behind."""
    from pyfastx_amd import _lib, fxi
    names = [b"N" * 5000, b"short"]
    offs = np.array([0, 5000, 5005], dtype=np.int64)
    packed = np.frombuffer(b"".join(names), dtype=np.uint8)
    cols = {k: np.array([1, 2], dtype=np.int64) for k in ("dlen", "rlen", "soff", "qoff")}
    p = str(tmp_path / "x.fxi")
    with pytest.raises(_lib.FxError) as ei:
        fxi.write_fastq_bulk(p, packed, offs, cols, 3)
    assert ei.value.code == _lib.FX_ERANGE and not os.path.exists(p)


@pytest.mark.parametrize("n,width,with_index", [(1_100_000, 1000, False), (640_000, 900, True)])
def test_fxi_bulk_steps_over_the_pending_byte_page(tmp_path, n, width, with_index):
    """Index files beyond 1 GiB: the page holding byte 2^30 is reserved by SQLite for locking and must stay unused.
    Case 1: the `read` table itself crosses it (table only); case 2: the table ends below it and the index b-tree
    crosses it."""
    import sqlite3
    from pyfastx_amd import _lib, fxi
    rng = np.random.default_rng(n)
    m = np.tile(rng.integers(65, 91, (1000, width), dtype=np.uint8), (n // 1000, 1))
    tag = np.char.zfill(np.arange(n).astype("S8"), 8)         # distinct 8-byte tail
    m[:, -8:] = np.frombuffer(tag.tobytes(), dtype=np.uint8).reshape(n, 8)
    packed = m.reshape(-1)
    offs = np.arange(n + 1, dtype=np.int64) * width
    cols = {k: np.arange(n, dtype=np.int64) + j for j, k in enumerate(("dlen", "rlen", "soff", "qoff"))}
    p = str(tmp_path / "big.fxi")
    if with_index:
        order = np.argsort(m.view("S%d" % width).reshape(n), kind="stable").astype(np.int64)
        db = fxi.write_fastq_bulk(p, packed, offs, cols, 77, order=order)
        db.close()
    else:
        db = fxi.connect(p)
        db.executescript(fxi.FASTQ_DDL)
        root = db.execute("SELECT rootpage FROM sqlite_master WHERE name='read'").fetchone()[0]
        db.close()
        _lib.fxi_bulk_rows(p, root, packed, offs, [cols[k] for k in ("dlen", "rlen", "soff", "qoff")])
    assert os.path.getsize(p) > (1 << 30)
    db = sqlite3.connect(p)
    assert db.execute("PRAGMA integrity_check").fetchall() == [("ok",)]
    assert db.execute("SELECT count(*) FROM read").fetchone()[0] == n
    for i in rng.integers(0, n, 20).tolist() + [0, n - 1]:
        nm = m[i].tobytes().decode()
        if with_index:
            assert db.execute("SELECT ID, qoff FROM read WHERE name=?", (nm,)).fetchone() == (i + 1, i + 3)
        else:
            assert db.execute("SELECT name, qoff FROM read WHERE ID=?", (i + 1,)).fetchone() == (nm, i + 3)
    db.execute("INSERT INTO read VALUES (NULL,'appended',1,2,3,4)")      # SQLite allocates past our pages
    assert db.execute("SELECT ID FROM read WHERE name='appended'").fetchone()[0] == n + 1
    db.close()


def test_comp_exchange_logic(oracle):
    """shard.comp_lead_from / comp_fold_leads (the host side of the sharded composition) against the composition
    of the whole stream, with the per-shard counting emulated on the CPU: a shard counts the bytes of the records
    whose header it holds, and -- into its lead row -- the bytes before its first header from lead_from on."""
    from pyfastx_amd import shard
    rng = np.random.default_rng(4)
    for trial in range(40):
        parts = []
        for i in range(int(rng.integers(1, 6))):
            parts.append(b">r%d %s\n" % (i, b"x" * int(rng.integers(0, 40))))
            s = bytes(rng.choice(list(b"ACGTNacgt"), int(rng.integers(0, 200))).astype(np.uint8))
            w = int(rng.integers(1, 50))
            parts += [s[p:p + w] + b"\n" for p in range(0, len(s), w)]
        raw = (b"junk\n" if trial % 5 == 0 else b"") + b"".join(parts)
        recs, _ = oracle.fasta_index(raw)
        n = len(recs)
        want = oracle.fasta_comp(raw, n)
        hoff, boff = [int(x) for x in recs["hoff"]], [int(x) for x in recs["boff"]]
        g = int(rng.integers(2, 6))
        cuts = sorted(set(int(x) for x in rng.integers(1, len(raw), g - 1)))
        bounds = [0] + cuts + [len(raw)]
        owned = [[i for i in range(n) if lo <= hoff[i] < hi] for lo, hi in zip(bounds[:-1], bounds[1:])]
        last_boffs = [boff[o[-1]] if o else -1 for o in owned]            # after stitching: the true boff
        comps, leads, nh = [], [], []
        for r, (lo, hi) in enumerate(zip(bounds[:-1], bounds[1:])):
            c = np.zeros((len(owned[r]), 128), dtype=np.int64)
            lead = np.zeros(128, dtype=np.int64)
            lf = shard.comp_lead_from(bounds[:-1], last_boffs, r)
            first_h = hoff[owned[r][0]] if owned[r] else hi
            for p in range(lo, hi):
                b = raw[p]
                if b == 10 or b >= 128:
                    continue
                if p < first_h:
                    if lf >= 0 and p >= lf:
                        lead[b] += 1
                    continue
                k = max(j for j, i in enumerate(owned[r]) if hoff[i] <= p)
                if p >= boff[owned[r][k]]:
                    c[k, b] += 1
            comps.append(c); leads.append(lead); nh.append(len(owned[r]))
        got = [shard.comp_fold_leads(comps[r], leads, nh, r) for r in range(len(comps))]
        got = np.concatenate([x for x in got if len(x)]) if n else np.zeros((0, 128), dtype=np.int64)
        np.testing.assert_array_equal(got, want, err_msg="trial %d cuts %s" % (trial, cuts))


@pytest.mark.parametrize("nrec", [0, 1, 50, 40000])
def test_fxi_bulk_comp_table_equals_inserts(tmp_path, nrec):
    """The `comp` table and its non-unique `seqidx` index written as pages (fx_fxi_bulk_rows without a TEXT column,
    fx_fxi_bulk_index_int) on top of an index file that already holds the other tables: same rows as the INSERT path
    (fasta.c:890-953), integrity_check ok, `WHERE seqid=?` served by the index."""
    import sqlite3
    from pyfastx_amd import fxi
    rng = np.random.default_rng(nrec)
    comp = np.zeros((nrec, 128), dtype=np.int64)
    for c, p in zip(b"ACGTNacgtn\r", (1, 1, 1, 1, .3, .5, .5, .5, .5, .1, .05)):
        comp[:, c] = rng.integers(0, 1 << 20, nrec) * (rng.random(nrec) < p)
    if nrec > 3:
        comp[2] = 0                                          # a record without any byte
        comp[3, 77] = 1 << 40                                # a large count, another letter
    names = ["r%d" % i for i in range(nrec)]
    cols = {k: np.arange(nrec, dtype=np.int64) + j for j, k in enumerate(("boff", "blen", "slen", "llen", "elen", "norm", "dlen"))}
    a, b = str(tmp_path / "a.fxi"), str(tmp_path / "b.fxi")
    for p in (a, b):
        db = fxi.connect(p)
        fxi.write_fasta(db, names, cols, 7)
        if p == a:
            fxi.write_fasta_comp(db, comp)
        db.close()
    db = fxi.write_fasta_comp_bulk(b, *fxi.comp_rows(comp))
    db.close()
    da, dbb = sqlite3.connect(a), sqlite3.connect(b)
    assert dbb.execute("PRAGMA integrity_check").fetchall() == [("ok",)]
    for q in ("SELECT * FROM comp ORDER BY ID", "SELECT * FROM seq ORDER BY ID", "SELECT * FROM comp WHERE seqid=0",
              "SELECT abc, num FROM comp WHERE seqid=3 ORDER BY ID", "SELECT name FROM sqlite_master WHERE type='index' ORDER BY name"):
        assert da.execute(q).fetchall() == dbb.execute(q).fetchall(), q
    plan = dbb.execute("EXPLAIN QUERY PLAN SELECT * FROM comp WHERE seqid=?", (5,)).fetchall()
    assert "seqidx" in plan[0][-1]
    dbb.execute("INSERT INTO comp VALUES (NULL, 9, 65, 1)")  # the loaded b-trees keep working
    assert dbb.execute("PRAGMA integrity_check").fetchall() == [("ok",)]
    da.close(); dbb.close()
    with pytest.raises(ValueError):
        fxi.write_fasta_comp_bulk(b, *fxi.comp_rows(comp))   # comp must be empty


@pytest.mark.parametrize("page_size", [512, 1024, 8192, 65536])
def test_fxi_bulk_loaders_other_page_sizes(tmp_path, page_size):
    """The page loaders take the page size (and the reserved bytes) from the file header: databases created with a
    page size other than SQLite's default are loaded and verified the same way."""
    import sqlite3
    from pyfastx_amd import _lib, fxi
    rng = np.random.default_rng(page_size)
    n = 20000
    names = [("read_%d_%s" % (i, "x" * int(rng.integers(0, 40)))).encode() for i in range(n)]
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(x) for x in names], out=offs[1:])
    packed = np.frombuffer(b"".join(names), dtype=np.uint8)
    cols = [rng.integers(0, 1 << 40, n).astype(np.int64) for _ in range(4)]
    order = np.array(sorted(range(n), key=names.__getitem__), dtype=np.int64)
    p = str(tmp_path / "p.fxi")
    db = sqlite3.connect(p)
    db.execute("PRAGMA page_size = %d" % page_size)
    db.executescript(fxi.FASTQ_DDL)
    db.execute("CREATE UNIQUE INDEX readidx ON read (name)")
    root = dict(db.execute("SELECT name, rootpage FROM sqlite_master").fetchall())
    assert db.execute("PRAGMA page_size").fetchone()[0] == page_size
    db.close()
    _lib.fxi_bulk_rows(p, root["read"], packed, offs, cols)
    _lib.fxi_bulk_index(p, root["readidx"], packed, offs, order)
    db = sqlite3.connect(p)
    assert db.execute("PRAGMA integrity_check").fetchall() == [("ok",)]
    assert db.execute("SELECT count(*) FROM read").fetchone()[0] == n
    for i in rng.integers(0, n, 40).tolist():
        assert db.execute("SELECT ID, dlen, qoff FROM read WHERE name=?", (names[i].decode(),)).fetchone() == (i + 1, int(cols[0][i]), int(cols[3][i]))
    db.close()


def test_fxi_bulk_integer_serial_types(tmp_path):
    """Every INTEGER serial type of the record format (0, 1, 1/2/3/4/6/8-byte two's complement, both signs, the
    extremes) read back by SQLite exactly as written by the page loader."""
    import sqlite3
    from pyfastx_amd import _lib, fxi
    vals = [0, 1, -1, 2, 127, 128, -128, -129, 32767, 32768, -32768, -32769, 8388607, 8388608, -8388608, -8388609,
            2147483647, 2147483648, -2147483648, -2147483649, 140737488355327, 140737488355328, -140737488355328,
            -140737488355329, 9223372036854775807, -9223372036854775808]
    n = len(vals)
    cols = [np.array(vals, dtype=np.int64), np.array(vals[::-1], dtype=np.int64),
            np.array(vals, dtype=np.int64) // 3, np.arange(n, dtype=np.int64)]
    names = [b"n%d" % i for i in range(n)]
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(x) for x in names], out=offs[1:])
    p = str(tmp_path / "i.fxi")
    db = fxi.connect(p)
    db.executescript(fxi.FASTQ_DDL)
    root = db.execute("SELECT rootpage FROM sqlite_master WHERE name='read'").fetchone()[0]
    db.close()
    _lib.fxi_bulk_rows(p, root, np.frombuffer(b"".join(names), dtype=np.uint8), offs, cols)
    db = sqlite3.connect(p)
    assert db.execute("PRAGMA integrity_check").fetchall() == [("ok",)]
    got = db.execute("SELECT dlen, rlen, soff, qoff FROM read ORDER BY ID").fetchall()
    assert got == [tuple(int(c[i]) for c in cols) for i in range(n)]
    db.close()


@pytest.mark.parametrize("pragma", ["PRAGMA auto_vacuum = FULL", "PRAGMA journal_mode = WAL"])
def test_fxi_bulk_refuses_files_it_cannot_extend(tmp_path, pragma):
    """Auto-vacuum databases interleave pointer-map pages and WAL databases keep their tail elsewhere: the page loaders
    refuse both (FX_EINVAL) instead of writing pages SQLite would not find; a non-database is refused as well."""
    import sqlite3
    from pyfastx_amd import _lib, fxi
    p = str(tmp_path / "v.fxi")
    db = sqlite3.connect(p)
    db.execute(pragma)
    db.executescript(fxi.FASTQ_DDL)
    root = db.execute("SELECT rootpage FROM sqlite_master WHERE name='read'").fetchone()[0]
    db.close()
    one = np.array([1], dtype=np.int64)
    with pytest.raises(_lib.FxError) as ei:
        _lib.fxi_bulk_rows(p, root, np.frombuffer(b"a", dtype=np.uint8), np.array([0, 1], dtype=np.int64), [one, one, one, one])
    assert ei.value.code == _lib.FX_EINVAL
    q = str(tmp_path / "not_a_db")
    open(q, "wb").write(b"x" * 8192)
    with pytest.raises(_lib.FxError) as ei:
        _lib.fxi_bulk_rows(q, 2, np.frombuffer(b"a", dtype=np.uint8), np.array([0, 1], dtype=np.int64), [one, one, one, one])
    assert ei.value.code == _lib.FX_EINVAL


def test_reference_opens_our_bulk_written_fxi(oracle, tmp_path):
    """The REAL reference (oracle/_ref, compiled from /root/reference) opens index files whose b-trees were written as
    pages by fx_fxi_bulk_rows / fx_fxi_bulk_index / fx_fxi_bulk_index_int -- FASTA (seq + chromidx, comp + seqidx) and
    FASTQ (read + readidx) -- and answers by id, by name (through the bulk-loaded UNIQUE INDEX) and with composition
    exactly as from its own index.  Runs only where oracle/_ref was built."""
    import glob
    import shutil
    import sys
    from conftest import ROOT, DATA
    if not glob.glob(os.path.join(ROOT, "oracle", "_ref", "pyfastx*.so")):
        pytest.skip("oracle/_ref not built here")
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    import pyfastx
    from pyfastx_amd import fxi
    # ---- FASTA
    raw = fixture_bytes("test.fa")
    recs, tot = oracle.fasta_index(raw)
    n = len(recs)
    names = [raw[r["name_off"]:r["name_off"] + r["name_len"]] for r in recs]
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(x) for x in names], out=offs[1:])
    packed = np.frombuffer(b"".join(names), dtype=np.uint8)
    order = np.array(sorted(range(n), key=names.__getitem__), dtype=np.int64)
    cols = {k: recs[k].astype(np.int64) for k in ("boff", "blen", "slen", "llen", "elen", "norm", "dlen")}
    ours, theirs = str(tmp_path / "ours.fa"), str(tmp_path / "theirs.fa")
    for p in (ours, theirs):
        open(p, "wb").write(raw)
    db = fxi.write_fasta_bulk(ours + ".fxi", packed, offs, cols, tot, order=order)
    db.close()
    db = fxi.write_fasta_comp_bulk(ours + ".fxi", *fxi.comp_rows(oracle.fasta_comp(raw, n)))
    db.close()
    fa, fb = pyfastx.Fasta(ours), pyfastx.Fasta(theirs, full_index=True)       # loads OUR index / builds its own
    assert len(fa) == len(fb) == n and fa.size == fb.size
    assert fa.composition == fb.composition and fa.gc_content == fb.gc_content
    rng = np.random.default_rng(3)
    for i in rng.integers(0, n, 60).tolist():
        nm = names[i].decode()
        assert fa[nm].id == fb[nm].id == i + 1 and fa[nm].seq == fb[nm].seq          # by name: chromidx
        assert fa[i].name == nm and fa[i][3:40].seq == fb[i][3:40].seq
        assert fa[i].composition == fb[i].composition and fa[i].gc_content == fb[i].gc_content
    assert "no such name" not in fa and names[5].decode() in fa
    assert fa.fetch(names[7].decode(), (1, 30)) == fb.fetch(names[7].decode(), (1, 30))
    del fa, fb
    # ---- FASTQ
    rawq = fixture_bytes("test.fq")
    rq, size, ln = oracle.fastq_index(rawq)
    m = len(rq)
    qn = [rawq[int(r["name_off"]):int(r["name_off"]) + int(r["name_len"])] for r in rq]
    qo = np.zeros(m + 1, dtype=np.int64)
    np.cumsum([len(x) for x in qn], out=qo[1:])
    qorder = np.array(sorted(range(m), key=qn.__getitem__), dtype=np.int64)
    qcols = {k: rq[k].astype(np.int64) for k in ("dlen", "rlen", "soff", "qoff")}
    oq, tq = str(tmp_path / "ours.fq"), str(tmp_path / "theirs.fq")
    for p in (oq, tq):
        open(p, "wb").write(rawq)
    db = fxi.write_fastq_bulk(oq + ".fxi", np.frombuffer(b"".join(qn), dtype=np.uint8), qo, qcols, size, order=qorder)
    c = oracle.fastq_composition(rawq)
    fxi.write_fastq_comp(db, [c["a"], c["c"], c["g"], c["t"], c["n"]], [c["maxlen"], c["minlen"], c["minqs"], c["maxqs"], c["phred"]])
    db.close()
    qa, qb = pyfastx.Fastq(oq), pyfastx.Fastq(tq)
    assert len(qa) == len(qb) == m and qa.size == qb.size and qa.composition == qb.composition
    for i in rng.integers(0, m, 40).tolist():
        nm = qn[i].decode()
        assert qa[nm].id == qb[nm].id == i + 1 and qa[nm].seq == qb[nm].seq and qa[i].qual == qb[i].qual


def _ref_pyfastx():
    import glob
    import sys
    from conftest import ROOT
    if not glob.glob(os.path.join(ROOT, "oracle", "_ref", "pyfastx*.so")):
        return None
    if os.path.join(ROOT, "oracle", "_ref") not in sys.path:
        sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    import pyfastx
    return pyfastx


def _indexed_pair(oracle, tmp_path):
    """test.fa / test.fq with index files written by fxi.py from the oracle's rows -> paths (no GPU involved)."""
    from pyfastx_amd import fxi
    raw = fixture_bytes("test.fa")
    pa = str(tmp_path / "k.fa")
    open(pa, "wb").write(raw)
    recs, tot = oracle.fasta_index(raw)
    names = [raw[r["name_off"]:r["name_off"] + r["name_len"]].decode() for r in recs]
    db = fxi.connect(pa + ".fxi")
    fxi.write_fasta(db, names, {k: recs[k] for k in ("boff", "blen", "slen", "llen", "elen", "norm", "dlen")}, tot)
    db.close()
    rawq = fixture_bytes("test.fq")
    pq = str(tmp_path / "k.fq")
    open(pq, "wb").write(rawq)
    rq, size, ln = oracle.fastq_index(rawq)
    qn = [rawq[int(r["name_off"]):int(r["name_off"]) + int(r["name_len"])].decode() for r in rq]
    db = fxi.connect(pq + ".fxi")
    fxi.write_fastq(db, qn, {k: rq[k] for k in ("dlen", "rlen", "soff", "qoff")}, size)
    db.close()
    return pa, names, [int(x) for x in recs["slen"]], pq, qn


def test_fasta_keys_sort_filter(oracle, tmp_path):
    """FastaKeys / FastqKeys (fakeys.c, fqkeys.c) over an existing index: the cases of the reference's
    tests/test_fakeys.py and tests/test_fqkeys.py, checked against plain Python over (name, length) pairs and -- where
    oracle/_ref is built -- against the real reference opening the same files."""
    from pyfastx_amd import api
    pa, names, lens, pq, qn = _indexed_pair(oracle, tmp_path)
    fa = api.Fasta(pa)                                       # loads the index; nothing is staged on a device
    n = len(names)
    keys = fa.keys()
    assert repr(keys) == "<FastaKeys> contains %d keys" % n and len(keys) == n
    assert list(keys) == names and list(keys) == names       # iteration restarts
    assert keys[0] == names[0] and keys[-1] == names[-1] and keys[17 - n] == names[17]
    assert keys[30:40] == names[30:40] and keys[-20:-10] == names[-20:-10] and keys[n - 3:n + 9] == names[-3:] and keys[5:2] == []
    assert names[9] in keys and "nope" not in keys and 7 not in keys
    assert list(keys.sort("id", reverse=True)) == names[::-1]
    assert list(keys.sort("name")) == sorted(names)
    by_len = [nm for nm, _ in sorted(zip(names, lens), key=lambda x: x[1])]            # stable, as sqlite's rowid order within ties
    got = list(keys.sort("length"))
    assert sorted(got) == sorted(by_len) and [lens[names.index(g)] for g in got] == sorted(lens)
    assert keys[0] == got[0] and keys[-1] == got[-1] and keys[3:6] == got[3:6]
    keys.reset()
    assert list(keys) == names
    ids = fa.keys()
    assert list(ids.filter(ids > 700)) == [nm for nm, l in zip(names, lens) if l > 700]
    assert len(ids) == sum(l > 700 for l in lens)
    assert list(ids.filter(600 <= ids <= 700)) == [nm for nm, l in zip(names, lens) if 600 <= l <= 700]
    assert list(ids.filter(ids % "JZ8226")) == [nm for nm in names if "JZ8226" in nm]
    want = sorted((nm for nm, l in zip(names, lens) if "JZ8226" in nm and l >= 300), reverse=True)
    assert list(ids.filter(ids % "JZ8226", ids >= 300).sort("name", reverse=True)) == want
    assert ids[0] == want[0] and ids[-1] == want[-1] and len(ids) == len(want)
    assert want[1] in ids and [nm for nm in names if nm not in want][0] not in ids       # `in` honours the filter
    ids.reset()
    assert len(ids) == n
    assert list(ids.filter(ids == lens[4])) == [nm for nm, l in zip(names, lens) if l == lens[4]]
    assert list(ids.filter(ids != lens[4], ids < 200)) == [nm for nm, l in zip(names, lens) if l != lens[4] and l < 200]
    k2 = fa.keys()
    with pytest.raises(IndexError):
        k2[len(k2)]
    with pytest.raises(ValueError):
        k2 % list
    with pytest.raises(ValueError):
        k2.filter()
    with pytest.raises(ValueError):
        k2.sort("sort")
    with pytest.raises(ValueError):
        k2 > list
    with pytest.raises(TypeError):
        k2[list]
    # FASTQ keys
    fq = api.Fastq(pq)
    qk = fq.keys()
    assert repr(qk) == "<FastqKeys> contains %d keys" % len(qn) and len(qk) == len(qn)
    assert list(qk) == qn and qk[0] == qn[0] and qk[-1] == qn[-1] and qk[123] == qn[123]
    assert qn[77] in qk and "zzz" not in qk and 3 not in qk
    with pytest.raises(IndexError):
        qk[len(qn)]
    # ---- the same questions put to the real reference on the same files
    ref = _ref_pyfastx()
    if ref is None:
        return
    ra = ref.Fasta(pa)
    rk, ok = ra.keys(), fa.keys()
    assert repr(rk) == repr(ok) and list(rk) == list(ok)
    for by in ("id", "name", "length"):
        for rev in (False, True):
            assert list(rk.sort(by, reverse=rev)) == list(ok.sort(by, reverse=rev)), (by, rev)
            assert rk[0] == ok[0] and rk[-2] == ok[-2] and rk[4:9] == ok[4:9]
    rk.reset(); ok.reset()
    assert (rk > 500) == (ok > 500) and (rk <= 900) == (ok <= 900) and (rk % "JZ83") == (ok % "JZ83")   # the fragments themselves
    assert list(rk.filter(rk >= 250, rk % "JZ82")) == list(ok.filter(ok >= 250, ok % "JZ82")) and len(rk) == len(ok)
    assert list(rk.filter(300 <= rk <= 400).sort("length", reverse=True)) == list(ok.filter(300 <= ok <= 400).sort("length", reverse=True))
    assert rk[1:7] == ok[1:7] and (names[0] in rk) == (names[0] in ok)
    rq_ = ref.Fastq(pq)
    assert list(rq_.keys()) == list(fq.keys()) and repr(rq_.keys()) == repr(fq.keys()) and rq_.keys()[-5] == fq.keys()[-5]


class _OracleShard:
    """Stand-in for Blob.fetch_ranges on one byte-range shard: the oracle's fetch on the bytes the shard holds, with
    the kernel's clamping of a range to them (fx_kernels.hpp, "clamp to the bytes we hold")."""

    def __init__(self, oracle, raw, base, end):
        self.o, self.raw, self.base, self.end = oracle, raw[base:end], base, end

    def fetch_ranges(self, off, blen, slen, flags=0, flags_per_query=None, skip=None):
        n = len(off)
        offs = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.maximum(slen, 0), out=offs[1:])
        buf = np.zeros(max(int(offs[-1]), 1), dtype=np.uint8)
        ol = np.zeros(n, dtype=np.int64)
        for i in range(n):
            lo, hi = max(int(off[i]), self.base), min(int(off[i] + blen[i]), self.end)
            fl = int(flags if flags_per_query is None else flags_per_query[i])
            sk = 0 if skip is None else int(skip[i])
            if sk and hi > lo:                                # fx_fetch_slices: despace, drop `skip`, keep `slen`, then reverse
                s = self.o.fetch(self.raw, lo - self.base, hi - lo, sk + int(slen[i]), fl & 5)[sk:]
                s = s[::-1] if fl & 2 else s
            else:
                s = self.o.fetch(self.raw, lo - self.base, hi - lo, int(slen[i]), fl) if hi > lo else b""
            buf[offs[i]:offs[i] + len(s)] = np.frombuffer(s, dtype=np.uint8)
            ol[i] = len(s)
        return buf[:int(offs[-1])], offs, ol


def _shard_queries(rng, recs, nq):
    ok = np.nonzero(recs["slen"] > 0)[0]
    ids = rng.choice(ok, nq)
    st = (rng.random(nq) * recs["slen"][ids]).astype(np.int64)
    sp = np.minimum(st + rng.integers(0, 400, nq), recs["slen"][ids])
    whole = rng.random(nq) < 0.1
    st[whole], sp[whole] = 0, recs["slen"][ids][whole]
    return ids, st, sp, rng.integers(0, 8, nq).astype(np.uint8)


def _expected_fetch(oracle, raw, recs, i, a, b, fl):
    r = recs[i]
    if _reg_of(raw, r):
        off, bl = oracle.slice_range(int(r["boff"]), int(r["llen"]), int(r["elen"]), a, b)
        return oracle.fetch(raw, off, bl, b - a, fl)
    s = oracle.fetch(raw, r["boff"], r["blen"], r["slen"], fl & 5)[a:b]       # despace the record, slice, then reverse
    return s[::-1] if fl & 2 else s


@pytest.mark.parametrize("seed", range(6))
def test_fetch_over_byte_range_shards(oracle, seed):
    """SURVEY 8e "Fetch": queries bucketed by the shard that holds their bytes, queries crossing a cut split and put
    together again -- shard.ShardFetcher with the per-shard kernel replaced by the oracle on the shard's bytes; the
    answers must equal the oracle's on the whole stream, for cuts anywhere (inside lines, inside CRLF, shards holding
    a fraction of one record)."""
    from pyfastx_amd import shard
    rng = np.random.default_rng(7100 + seed)
    eol = b"\r\n" if seed & 1 else b"\n"
    parts = []
    for i in range(12):
        parts.append(b">c%d d" % i + eol)
        s = bytes(rng.choice(list(b"ACGTNacgtn"), int(rng.integers(1, 4000))).astype(np.uint8))
        w = int(rng.integers(7, 80))
        ragged = i % 5 == 3
        odd_at = -1
        if i % 5 == 1 and w > 2:                              # ONE odd line in the middle, every other line full: norm = 1, not line-regular
            m = max(len(s) // w, 3)
            s = (s * (m * w // len(s) + 2))[:m * w + (w - 2)]
            odd_at = int(rng.integers(1, m))
        p = j = 0
        while p < len(s):
            k = w if not ragged else int(rng.integers(1, w + 1))
            if j == odd_at:
                k = max(w - 2, 1)
            parts.append(s[p:p + k] + eol)
            p += k
            j += 1
    raw = b"".join(parts)
    if seed == 4:
        raw = raw[:-len(eol)]
    recs, _ = oracle.fasta_index(raw)
    G = (2, 3, 5, 8, 16, 2)[seed]
    cuts = sorted(set(int(x) for x in rng.integers(1, len(raw) - 1, G - 1)))
    bases, ends = [0] + cuts, cuts + [len(raw)]
    table = {k: recs[k] for k in ("boff", "blen", "slen", "llen", "elen", "norm")}
    table["reg"] = np.array([_reg_of(raw, r) for r in recs], dtype=np.int32)
    assert ((table["reg"] == 0) & (recs["norm"] == 1) & (recs["slen"] > 0)).sum() >= 1       # the odd kind is in the mix
    ids, st, sp, fl = _shard_queries(rng, recs, 600)
    # make sure some queries cross each cut: a window around every cut that lies inside a sequence
    extra = []
    for c in cuts:
        i = int(np.searchsorted(recs["boff"], c, "right")) - 1
        if i >= 0 and recs["boff"][i] < c < recs["boff"][i] + recs["blen"][i] and recs["slen"][i] > 4:
            bpl = max(int(recs["llen"][i] - recs["elen"][i]), 1)
            mid = min(int((c - recs["boff"][i]) // int(recs["llen"][i]) * bpl), int(recs["slen"][i]) - 1)
            extra.append((i, max(mid - 150, 0), min(mid + 150, int(recs["slen"][i]))))
    if extra:
        e = np.array(extra, dtype=np.int64)
        ids, st, sp = np.concatenate([ids, e[:, 0]]), np.concatenate([st, e[:, 1]]), np.concatenate([sp, e[:, 2]])
        fl = np.concatenate([fl, rng.integers(0, 8, len(extra)).astype(np.uint8)])
    want = [_expected_fetch(oracle, raw, recs, int(i), int(a), int(b), int(f)) for i, a, b, f in zip(ids, st, sp, fl)]
    shards = {r: _OracleShard(oracle, raw, bases[r], ends[r]) for r in range(len(bases))}
    # (1) all shards in one process (G logical shards on one GPU)
    qidx, buf, offs = shard.ShardFetcher(shards, bases, ends, table).fetch(ids, st, sp, flags_per_query=fl)
    assert sorted(qidx.tolist()) == list(range(len(ids)))
    for j, qi in enumerate(qidx.tolist()):
        assert buf[offs[j]:offs[j + 1]].tobytes() == want[qi], (seed, qi)
    off, bl, _, _ = shard.slice_ranges(table, ids, st, sp)
    P = shard.route_ranges(bases, ends, off, bl)
    assert (P["cnt"] > 1).sum() >= (1 if extra else 0) and int(P["plen"].sum()) == int(np.minimum(off + bl, len(raw)).sum() - off.sum())
    # (2) one process per shard: every process fetches what it holds, the cross-cut pieces are exchanged, every query
    # is answered exactly once
    procs = [shard.ShardFetcher({r: shards[r]}, bases, ends, table, exchange=lambda mine: mine) for r in shards]
    pool = []
    for pr in procs:                                          # first round: collect what each would contribute
        pr.exchange = lambda mine, pool=pool: (pool.extend(mine), [])[1]
        pr.fetch(ids, st, sp, flags_per_query=fl)
    seen = np.zeros(len(ids), dtype=np.int64)
    for pr in procs:                                          # second round: everybody sees everybody's pieces
        pr.exchange = lambda mine, pool=pool: list(pool)
        qidx, buf, offs = pr.fetch(ids, st, sp, flags_per_query=fl)
        seen[qidx] += 1
        for j, qi in enumerate(qidx.tolist()):
            assert buf[offs[j]:offs[j + 1]].tobytes() == want[qi], (seed, qi)
    assert (seen == 1).all()


def test_line_regular_rule(oracle):
    """shard.line_regular_rule (= csrc/fx_kernels.hpp line_regular, the column every fetch path goes by): whenever it
    lets a norm=1 record through to the line arithmetic of sequence.c:498-510, the arithmetic is right at EVERY position
    of the record -- brute force over random records with odd lines anywhere, blank lines, CRLF and unterminated ends;
    and it does let the ordinary records through.  A record it turns away is sliced after despacing."""
    rng = np.random.default_rng(4)
    passed = odd = 0
    for it in range(1500):
        eol = b"\r\n" if it % 2 else b"\n"
        parts = []
        for i in range(5):
            parts.append(b">r%d" % i + eol)
            w = int(rng.integers(2, 9))
            for j in range(int(rng.integers(0, 6))):
                parts.append(b"A" * (w if rng.random() < 0.8 else int(rng.integers(1, 2 * w + 2))) + eol)
            if rng.random() < 0.2:
                parts.append(eol)
        text = b"".join(parts)
        if it % 7 == 0 and text.endswith(eol):
            text = text[:-len(eol)]
        recs, _ = oracle.fasta_index(text)
        for k, r in enumerate(recs):
            if not _reg_of(text, r):
                odd += bool(r["norm"])
                continue
            passed += 1
            n = int(r["slen"])
            full = oracle.fetch(text, r["boff"], r["blen"], r["slen"])
            for a in range(n):
                off, bl = oracle.slice_range(int(r["boff"]), int(r["llen"]), int(r["elen"]), a, n)
                assert oracle.fetch(text, off, bl, n - a, 0) == full[a:], (text, k, a)
                off, bl = oracle.slice_range(int(r["boff"]), int(r["llen"]), int(r["elen"]), 0, a + 1)
                assert oracle.fetch(text, off, bl, a + 1, 0) == full[:a + 1], (text, k, a)
    assert passed > 3000 and odd > 1000
    # well-formed records are regular: nothing ordinary takes the slow path
    for text in (b">a\nACGT\nACGT\nAC\n", b">a\r\nACGT\r\nACGT\r\nAC\r\n", b">a\nACGT\nACGT\n", b">a\nACGT\nAC", b">a\nAC\n", b">a\nACGT\nACGT\nAC"):
        recs, _ = oracle.fasta_index(text)
        assert _reg_of(text, recs[0]) == 1, text
    recs, _ = oracle.fasta_index(b">c\nAAAA\nCC\nGGGG\n")                     # the golden "one odd middle line" record: norm = 1
    assert int(recs[0]["norm"]) == 1 and _reg_of(b">c\nAAAA\nCC\nGGGG\n", recs[0]) == 0


def test_fastx_header_cut(oracle):
    """Fastx's cut of a header line into (name, comment) (api._kseq_header: kseq.c:148-149 with the one-CR rule of
    kseq.c:106) against the oracle's records, on the headers of the kseq test inputs."""
    import random
    from kseq_cases import FIXED, gen
    from pyfastx_amd.api import _kseq_header
    assert _kseq_header(b"abc def  ghi", False) == (b"abc", b"def  ghi") and _kseq_header(b"abc", False) == (b"abc", None)
    assert _kseq_header(b"a\tb c\r", False) == (b"a", b"b c") and _kseq_header(b"a\x0bb", False) == (b"a", b"b")
    assert _kseq_header(b"", False) == (b"", None) and _kseq_header(b"a \r", False) == (b"a", b"\r") and _kseq_header(b"a\r", False) == (b"a", b"")
    assert _kseq_header(b"a ", True) == (b"a", None) and _kseq_header(b"a ", False) == (b"a", b"")
    rng = random.Random(77)
    seen = 0
    for data in list(FIXED) + [gen(rng) for _ in range(200)]:
        for r in oracle.kseq(data)[0]:
            e = data.find(b"\n", int(r["name_off"]))
            raw = data[int(r["name_off"]):e if e >= 0 else len(data)]
            nm, cm = _kseq_header(raw, e < 0)
            want_c = None if r["com_len"] < 0 else data[int(r["com_off"]):int(r["com_off"] + r["com_len"])]
            assert (nm, cm) == (data[int(r["name_off"]):int(r["name_off"] + r["name_len"])], want_c), (data, raw)
            seen += 1
    assert seen > 1000


def test_fastx_batch_builds_the_reference_tuples(oracle):
    """_fxobj.fastx_batch (the C loop behind Fastx.__iter__: header cut, "s" strings, comment / quality buffer rules of
    fastx.c:6-30) fed with record tables made from the oracle's records, in batches, against fxoracle.fastx_tuples."""
    import random
    from kseq_cases import FIXED, gen
    from pyfastx_amd import _fxobj
    from pyfastx_amd._lib import KSEQ_REC
    rng = random.Random(909)
    for data in list(FIXED) + [gen(rng) for _ in range(200)]:
        recs, seq, qual, _ = oracle.kseq(data)
        t = np.zeros(len(recs), dtype=KSEQ_REC)
        hdrs = []
        for k, r in enumerate(recs):
            e = data.find(b"\n", int(r["name_off"]))
            hdrs.append(data[int(r["name_off"]):e if e >= 0 else len(data)])
            t[k] = (r["name_off"], 0, r["seq_len"], r["seq_off"], len(hdrs[-1]), 0, 0,
                    (1 if r["qual_len"] != -1 else 0) | (2 if r["qual_len"] == -2 else 0) | (4 if e < 0 else 0))
        # the quality strings at the sequence offsets, as fx_kseq_fetch lays them out
        qbuf = np.zeros(max(int(seq.size), 1), dtype=np.uint8)
        for r in recs:
            if r["qual_len"] > 0:
                qbuf[int(r["seq_off"]):int(r["seq_off"] + r["qual_len"])] = qual[int(r["qual_off"]):int(r["qual_off"] + r["qual_len"])]
        for fmt, com in (("fasta", 0), ("fasta", 1), ("fastq", 0), ("fastq", 1), ("fasta", 2), ("fastq", 2)):
            state, got, batches = [False, None], [], []
            for a in range(0, len(recs), 3):
                b = min(len(recs), a + 3)
                ho = np.zeros(b - a + 1, dtype=np.int64)
                np.cumsum([len(h) for h in hdrs[a:b]], out=ho[1:])
                base = int(t["seq_cum"][a])
                end = int(t["seq_cum"][b - 1] + t["seq_len"][b - 1])
                batches.append((np.frombuffer(b"".join(hdrs[a:b]), dtype=np.uint8), ho, seq[base:end].copy(),
                                qbuf[base:end].copy() if fmt == "fastq" else None, t[a:b]))
                got += _fxobj.fastx_batch(*batches[-1], fmt == "fastq", com, state)
            # (2: no comment element, a non-empty comment joined to the name -- the index-free iteration of Fasta / Fastq with full_name)
            want = oracle.index_free_tuples(data, fmt, True) if com == 2 else oracle.fastx_tuples(data, fmt, comment=bool(com))
            assert got == want, (data, fmt, com)
            # the iterator type Fastx.__iter__ returns, fed with the same batches (and an empty one in between)
            feed = iter(batches[:1] + [(np.zeros(0, np.uint8), np.zeros(1, np.int64), np.zeros(0, np.uint8), None, t[:0])] + batches[1:] + [None])
            it = _fxobj.FastxIter(feed.__next__, fmt == "fastq", com)
            assert iter(it) is it and list(it) == want and list(it) == []


def test_row_cursor_and_read_batch_cols(tmp_path):
    """_fxobj.RowCursor (the `read` table stepped from C through the library the sqlite3 module has loaded) against the
    sqlite3 module's rows: batches, the short last batch, names that are not valid UTF-8, an exclusive lock held by another
    connection (RuntimeError: the caller falls back); read_batch_cols makes the same objects as read_batch."""
    import sqlite3
    from pyfastx_amd import _fxobj, api, fxi
    p = str(tmp_path / "t.fxi")
    db = fxi.connect(p)
    db.execute("CREATE TABLE read (ID INTEGER PRIMARY KEY, name TEXT, dlen INTEGER, rlen INTEGER, soff INTEGER, qoff INTEGER)")
    n = 2500
    names = [b"r%d" % i if i % 7 else b"bad\xff\xfe%d" % i for i in range(n)]
    db.executemany("INSERT INTO read VALUES (?, CAST(? AS TEXT), ?, ?, ?, ?)",
                   [(i + 1, names[i], 10 + i % 3, 3, (1 << 33) + 40 * i, (1 << 33) + 40 * i + 20) for i in range(n)])
    want = db.execute("SELECT * FROM read ORDER BY ID").fetchall()
    sql = "SELECT ID, name, dlen, rlen, soff, qoff FROM read ORDER BY ID"
    cur = _fxobj.RowCursor(p, sql)
    got, objs = [], []
    seq = np.frombuffer(b"ACG" * 1024, dtype=np.uint8)
    while True:
        b = cur.fetch(1024)
        if b is None:
            break
        k, nm, raw = b
        cols = np.frombuffer(raw, dtype=np.int64).reshape(5, k)
        got += [(int(cols[0, i]), nm[i]) + tuple(int(cols[j, i]) for j in range(1, 5)) for i in range(k)]
        offs = np.arange(k + 1, dtype=np.int64) * 3
        a = _fxobj.read_batch_cols(api.Read, "fq", nm, raw, seq, seq, offs)
        b2 = _fxobj.read_batch(api.Read, "fq", want[len(objs):len(objs) + k], seq, seq, offs)
        assert [(x.id, x.name, x._desc_len, x._read_len, x._soff, x._qoff, x.seq, x.qual, len(x)) for x in a] == \
               [(x.id, x.name, x._desc_len, x._read_len, x._soff, x._qoff, x.seq, x.qual, len(x)) for x in b2]
        objs += a
    assert got == want and len(objs) == n and cur.fetch(5) is None
    lock = sqlite3.connect(p, isolation_level=None)
    lock.execute("BEGIN EXCLUSIVE")
    with pytest.raises(RuntimeError):
        _fxobj.RowCursor(p, sql).fetch(1)
    lock.execute("COMMIT")
    with pytest.raises(RuntimeError):
        _fxobj.RowCursor(str(tmp_path / "none" / "x.fxi"), sql)


def test_kseq_line_model_equals_the_oracle(oracle):
    """tools/kseq_line_model.py -- the executable model k_kq_walk (fx_kseq.hpp) transliterates: kseq_read over a line
    table with its two 64-line steps and the parallel passes over the regular prefix -- against the byte-level oracle, with and
    without them."""
    import random
    import sys
    from conftest import ROOT
    from kseq_cases import FIXED, gen
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kseq_line_model as M
    rng = random.Random(505)
    for data in list(FIXED) + [gen(rng) for _ in range(250)]:
        recs, seq, qual, code = oracle.kseq(data)
        want = []
        for r in recs:
            com = None if r["com_len"] < 0 else data[int(r["com_off"]):int(r["com_off"] + r["com_len"])]
            q = None if r["qual_len"] == -1 else bytes(qual[int(r["qual_off"]):int(r["qual_off"]) + max(int(r["qual_len"]), 0)])
            want.append((data[int(r["name_off"]):int(r["name_off"] + r["name_len"])], com, bytes(seq[int(r["seq_off"]):int(r["seq_off"] + r["seq_len"])]), q,
                         bool(r["qual_len"] == -2)))
        for fast, pre in ((True, True), (True, False), (False, False)):
            got, gcode = M.materialise(data, fast, pre)
            have = [M.header_parts(hdr, bool(fl & M.F_HDR_UNTERM)) + (s, q, bool(fl & M.F_UNTOUCHED)) for hdr, fl, s, q in got]
            assert have == want and gcode == code, (data, fast, pre)


def test_shard_fetcher_degenerate_batches(oracle):
    """ShardFetcher: an empty batch, a batch of zero-length and past-the-end intervals, one shard only, a shard that
    holds a single byte."""
    from pyfastx_amd import shard
    raw = b">a\nACGTACGT\nACGT\n>b\nTTTT\n>e\n>c\nGGGGCCCC\nGG"
    recs, _ = oracle.fasta_index(raw)
    table = {k: recs[k] for k in ("boff", "blen", "slen", "llen", "elen", "norm")}
    table["reg"] = np.array([_reg_of(raw, r) for r in recs], dtype=np.int32)
    for cuts in ([], [1], [5, 6], [len(raw) - 1], list(range(3, len(raw) - 1, 7))):
        bases, ends = [0] + cuts, cuts + [len(raw)]
        f = shard.ShardFetcher({r: _OracleShard(oracle, raw, bases[r], ends[r]) for r in range(len(bases))}, bases, ends, table)
        q, buf, offs = f.fetch([], [], [])
        assert q.size == 0 and buf.size == 0 and offs.tolist() == [0]
        ids = np.array([0, 0, 1, 2, 3, 3, 3], dtype=np.int64)
        st = np.array([0, 5, 4, 0, 0, 9, 10], dtype=np.int64)
        sp = np.array([0, 5, 4, 0, 10, 10, 10], dtype=np.int64)
        fl = np.array([0, 6, 2, 4, 6, 0, 1], dtype=np.uint8)
        q, buf, offs = f.fetch(ids, st, sp, flags_per_query=fl)
        assert sorted(q.tolist()) == list(range(7))
        for j, qi in enumerate(q.tolist()):
            assert buf[offs[j]:offs[j + 1]].tobytes() == _expected_fetch(oracle, raw, recs, int(ids[qi]), int(st[qi]), int(sp[qi]), int(fl[qi])), (cuts, qi)


def test_names_are_stored_verbatim_by_every_writer(tmp_path):
    """the INSERT writers bind the name bytes as they are in the file (CAST(? AS TEXT)), like sqlite3_bind_text
    in the reference (index.c:239-251) and like the page loader -- a UTF-8 name does not turn into mojibake, invalid UTF-8
    survives, and fxi.connect reads both back the way names_lookup encodes queries (utf-8 / surrogateescape)."""
    from pyfastx_amd import fxi
    names = ["é".encode("utf-8"), b"plain", b"\xff\xfebad", "日本".encode("utf-8")]
    cols = {k: np.arange(4, dtype=np.int64) for k in ("boff", "blen", "slen", "llen", "elen", "norm", "dlen", "rlen", "soff", "qoff")}
    for kind in ("fa", "fq", "fa_bulk"):
        p = str(tmp_path / (kind + ".fxi"))
        if kind == "fa":
            db = fxi.connect(p); fxi.write_fasta(db, names, cols, 10)
        elif kind == "fq":
            db = fxi.connect(p); fxi.write_fastq(db, names, cols, 10)
        else:
            offs = np.zeros(5, dtype=np.int64); np.cumsum([len(x) for x in names], out=offs[1:])
            db = fxi.write_fasta_bulk(p, np.frombuffer(b"".join(names), dtype=np.uint8), offs, cols, 10)
        tab, col = ("read", "name") if kind == "fq" else ("seq", "chrom")
        raw = [bytes(r[0]) for r in db.execute("SELECT CAST(%s AS BLOB) FROM %s ORDER BY ID" % (col, tab))]
        assert raw == names, kind
        assert [r[0] for r in db.execute("SELECT typeof(%s) FROM %s" % (col, tab))] == ["text"] * 4
        back = [r[0] for r in db.execute("SELECT %s FROM %s ORDER BY ID" % (col, tab))]
        assert back[0] == "é" and back[3] == "日本" and back[2].encode("utf-8", "surrogateescape") == names[2]
        # by-name probe with a str key, as Fasta.__getitem__ does (SQLite compares the UTF-8 bytes)
        assert db.execute("SELECT ID FROM %s WHERE %s=?" % (tab, col), ("é",)).fetchone() == (1,)
        db.close()
    # str names (key_func results) are encoded the same way
    db = fxi.connect(str(tmp_path / "k.fxi")); fxi.write_fasta(db, ["é", "x"], {k: v[:2] for k, v in cols.items()}, 3)
    assert bytes(db.execute("SELECT CAST(chrom AS BLOB) FROM seq WHERE ID=1").fetchone()[0]) == "é".encode("utf-8")


def test_merge_index_parts_completes_names_cut_by_a_shard_boundary():
    """shard.merge_index_parts (the merged .fxi of a sharded build): rows in shard order, and a name whose header line
    crosses a cut -- the owner sees only its first bytes -- is completed from the next shards' first bytes, tiny shards
    in between included."""
    from pyfastx_amd import shard
    raw = b">alpha_long_name desc\nACGT\n>beta\nGG\n"
    cuts = [5, 7, 8, 30]                                    # the first name is cut three times
    bounds = [0] + cuts + [len(raw)]
    parts = []
    for i in range(len(bounds) - 1):
        lo, hi = bounds[i], bounds[i + 1]
        rows = {k: np.zeros(0, dtype=np.int64) for k in ("hoff", "boff", "blen", "slen", "llen", "elen", "norm", "dlen", "name_len", "reg")}
        names = []
        if lo == 0:
            rows = {k: np.array([v], dtype=np.int64) for k, v in dict(hoff=0, boff=22, blen=5, slen=4, llen=5, elen=1, norm=1, dlen=20, name_len=15, reg=1).items()}
            names = [raw[1:hi]]                              # what a fetch clamped to the shard's bytes returns
        if lo <= 27 < hi:
            rows = {k: np.array([v], dtype=np.int64) for k, v in dict(hoff=27, boff=33, blen=3, slen=2, llen=3, elen=1, norm=1, dlen=4, name_len=4, reg=1).items()}
            names = [raw[28:min(32, hi)]]
        parts.append((lo, hi - lo, rows, names, raw[lo:min(hi, lo + shard.HEAD_BYTES)]))
    t = shard.merge_index_parts(parts)
    assert t["names"] == [b"alpha_long_name", b"beta"] and t["boff"].tolist() == [22, 33] and t["seq_len"] == 6
    assert t["bases"] == bounds[:-1] and t["ends"] == bounds[1:]


def test_gzindex_rows_round_trip(tmp_path):
    """fxi.write_gzindex / read_gzindex: the zran row layout of util.c:461-529 with bits and windows (single-stream gzip)
    and without (BGZF member boundaries)."""
    from pyfastx_amd import fxi
    rng = np.random.default_rng(1)
    db = fxi.connect(str(tmp_path / "g.fxi"))
    db.executescript(fxi.FASTA_DDL)
    cmp_, unc = [10, 5000, 9000], [0, 1048576, 2200000]
    bits, has = [0, 3, 0], [0, 1, 1]
    win = rng.integers(0, 256, 2 * 32768, dtype=np.uint8)
    fxi.write_gzindex(db, 12345, 3000000, cmp_, unc, bits=bits, has_data=has, windows=win)
    rows = [bytes(r[0]) for r in db.execute("SELECT content FROM gzindex ORDER BY ID")]
    assert rows[0] == b"GZIDX" and len(rows) == 8 + 4 * 3 + 2 and len(rows[-1]) == 32768
    g = fxi.read_gzindex(db)
    assert g["compressed_size"] == 12345 and g["uncompressed_size"] == 3000000
    assert g["cmp"].tolist() == cmp_ and g["uncmp"].tolist() == unc and g["bits"].tolist() == bits and g["has"].tolist() == has
    assert g["windows"].tobytes() == win.tobytes()
    db2 = fxi.connect(str(tmp_path / "b.fxi"))
    db2.executescript(fxi.FASTA_DDL)
    fxi.write_gzindex(db2, 99, 500, [0, 40], [0, 300])       # BGZF: no bits, no windows
    g2 = fxi.read_gzindex(db2)
    assert g2["cmp"].tolist() == [0, 40] and int(g2["has"].sum()) == 0 and g2["windows"].size == 0
    db3 = fxi.connect(str(tmp_path / "e.fxi"))
    db3.executescript(fxi.FASTA_DDL)
    assert fxi.read_gzindex(db3) is None


def test_cli_arguments_match_the_reference_commands():
    """pyfastx_amd.cli: the options of `pyfastx subseq / sample / extract` (pyfastxcli.py argument definitions) parse, the
    region syntax is the reference's, and nothing touches a GPU before a file is opened."""
    import re
    from pyfastx_amd import cli
    with pytest.raises(SystemExit):
        cli.main(["sample", "x.fa"])                         # -n or -p is required, as in the reference
    assert re.split("[:-]", "chr1:10-20") == ["chr1", "10", "20"]
    with pytest.raises((FileExistsError, FileNotFoundError, OSError)):
        cli.main(["subseq", "/nonexistent.fa", "chr1:1-5"])
    with pytest.raises((FileExistsError, FileNotFoundError, OSError)):
        cli.main(["extract", "--reverse-complement", "--out-fasta", "-l", "names.txt", "/nonexistent.fq"])


@pytest.mark.parametrize("seed", range(3))
def test_shard_route_equals_the_vectorised_statement(seed):
    """fx_shard_route (the library's one-pass routing of a query batch) against shard.slice_ranges + shard.route_ranges:
    byte ranges, slices after despacing for records that are not line-regular, answering shard, shards touched, and the
    routed order (by shard, then by position in the batch); a record id outside the table is an error."""
    from pyfastx_amd import _lib, shard
    rng = np.random.default_rng(9200 + seed)
    nrec = 40
    slen = rng.integers(0, 30_000, nrec)
    bpl, elen = int(rng.integers(1, 80)), 1 + (seed & 1)
    blen = slen + (slen + bpl - 1) // bpl * elen
    boff = np.cumsum(np.concatenate([[30], blen[:-1] + 30]))
    reg = (rng.random(nrec) > 0.3).astype(np.uint8)
    table = {"boff": boff, "blen": blen, "llen": np.full(nrec, bpl + elen), "elen": np.full(nrec, elen), "reg": reg}
    if seed == 2:
        table["llen"][:5] = elen                              # bytes per line 0: never the line arithmetic
    total = int(boff[-1] + blen[-1])
    G = (1, 3, 7)[seed]
    cuts = sorted(set(int(x) for x in rng.integers(1, total - 1, G - 1)))
    bases, ends = [0] + cuts, cuts + [total]
    n = 70_000 * (seed + 1)                                   # past 65536: several threads
    ids = rng.integers(0, nrec, n)
    a = (rng.random(n) * np.maximum(slen[ids], 1)).astype(np.int64)
    b = np.minimum(a + rng.integers(0, 3000, n), slen[ids] + rng.integers(0, 3, n))
    b[:50] = a[:50] - rng.integers(0, 2, 50)                  # empty and inverted intervals
    fl = rng.integers(0, 8, n).astype(np.uint8)
    cols = {k: np.ascontiguousarray(table[k], dtype=np.int64) for k in ("boff", "blen", "llen", "elen")}
    cols["reg"] = reg
    R = _lib.shard_route(ids, a, b, cols, bases, ends, 0, fl)
    off, ln, sk, tk = shard.slice_ranges(table, ids, a, b)
    P = shard.route_ranges(bases, ends, off, ln)
    o = R["order"]
    assert (np.sort(o) == np.arange(n)).all()
    for k, w in (("off", off), ("len", ln), ("skip", sk), ("take", tk), ("fl", fl), ("cnt", P["cnt"])):
        assert (R[k] == w[o]).all(), k
    assert R["shard_start"][0] == 0 and R["shard_start"][-1] == n
    for r in range(len(bases)):
        seg = o[R["shard_start"][r]:R["shard_start"][r + 1]]
        assert (P["first"][seg] == r).all() and (np.diff(seg) > 0).all()
    assert (R["cnt"] > 1).sum() > 0 or G == 1
    ids[n // 2] = nrec
    with pytest.raises(_lib.FxError):
        _lib.shard_route(ids, a, b, cols, bases, ends)


def _gz_text(rng, n):
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.choice(4, n, p=[.295, .205, .205, .295])]
    low = np.repeat(rng.random(n // 5000 + 1) < 0.4, 5000)[:n]
    letters = np.where(low, letters + 32, letters).astype(np.uint8)
    letters[n // 3:n // 3 + 200_000] = ord("N")
    rows = letters[:n - n % 60].reshape(-1, 60)
    return b">chr1 first\n" + b"\n".join(r.tobytes() for r in rows) + b"\n"


@pytest.mark.parametrize("shape", ["level6", "level1", "level9", "stored_inside", "fastq"])
def test_parallel_gunzip_equals_zlib(shape):
    """fx_pgzip.hpp (the first open of a single gzip stream on all cores; host code, no device): block starts searched behind
    the cuts, pieces decoded with markers for the 32 KiB in front of them, resolved in order -- the bytes are zlib's, for
    three compression levels, a stream with stored blocks in the middle (the pieces decode through them), FASTQ text."""
    import zlib
    from pyfastx_amd import _lib
    rng = np.random.default_rng(len(shape))
    if shape == "fastq":
        n = 150_000
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n, 150))]
        qual = rng.integers(35, 71, (n, 150)).astype(np.uint8)
        nl = np.full((n, 1), 10, dtype=np.uint8)
        hdr = np.frombuffer(b"".join(b"@SYN:1:FC:%07d 1:N:0\n" % i for i in range(n)), dtype=np.uint8).reshape(n, -1)
        plus = np.tile(np.frombuffer(b"+\n", dtype=np.uint8), (n, 1))
        raw = np.concatenate([hdr, seq, nl, plus, qual, nl], axis=1).tobytes()
    else:
        raw = _gz_text(rng, 36_000_000)
    if shape == "stored_inside":                              # incompressible bytes in the middle: zlib emits stored blocks
        raw = raw[:9_000_000] + bytes(rng.integers(0, 256, 3_000_000, dtype=np.uint8)) + raw[9_000_000:]
    level = {"level1": 1, "level9": 9}.get(shape, 6)
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    gz = co.compress(raw) + co.flush()
    assert len(gz) > 8 << 20                                  # several pieces of >= 4 MiB
    got = _lib.gunzip_parallel(gz, threads=6)
    assert got is not None, shape
    out, npts = got
    assert out == raw and npts >= 1 + len(raw) // (4 << 20)


def test_parallel_gunzip_declines_what_it_is_not_sure_of():
    """Small inputs, several members, a damaged trailer or a flipped bit: None (the caller inflates serially, and zlib names
    the error), never wrong bytes."""
    import zlib
    from pyfastx_amd import _lib
    rng = np.random.default_rng(9)
    raw = _gz_text(rng, 30_000_000)
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    gz = co.compress(raw) + co.flush()
    assert _lib.gunzip_parallel(gz[:3 << 20] + gz[-8:], 4) is None           # too small / cut short
    two = zlib.compressobj(6, zlib.DEFLATED, 31)
    second = two.compress(b">x\nACGT\n") + two.flush()
    assert _lib.gunzip_parallel(gz + second, 4) is None                     # two members: gzread semantics are the serial path's
    bad = bytearray(gz)
    bad[-6] ^= 0x40                                                          # the CRC-32 of the trailer
    assert _lib.gunzip_parallel(bytes(bad), 4) is None
    bad = bytearray(gz)
    bad[len(gz) // 2] ^= 0x10                                                # a bit of the deflate data
    got = _lib.gunzip_parallel(bytes(bad), 4)
    assert got is None or got[0] == raw                                      # (only if the flipped bit happened to change nothing)


# ---------------------------------------------------------------------------------- round 4: the object layer's C paths
def _tiny_fastq(tmp_path, crlf=False):
    import sqlite3
    nl = "\r\n" if crlf else "\n"
    recs = [("r%d" % i, "ACGTN" * 3 + "A" * i, "".join(chr(40 + ((i + j) % 30)) for j in range(15 + i))) for i in range(7)]
    path = str(tmp_path / "t.fq")
    rows, off = [], 0
    with open(path, "w", newline="") as f:
        for i, (n, s, q) in enumerate(recs):
            h = "@%s some words" % n
            f.write(h + nl + s + nl + "+" + nl + q + nl)
            soff = off + len(h) + len(nl)
            qoff = soff + len(s) + len(nl) + 1 + len(nl)
            rows.append((i + 1, n, len(h) + len(nl) - 1, len(s), soff, qoff))
            off = qoff + len(q) + len(nl)
    db = sqlite3.connect(path + ".fxi")
    db.execute("CREATE TABLE read (ID INTEGER PRIMARY KEY, name TEXT, dlen INTEGER, rlen INTEGER, soff INTEGER, qoff INTEGER)")
    db.executemany("INSERT INTO read VALUES (?,?,?,?,?,?)", rows)
    db.execute("CREATE UNIQUE INDEX readidx ON read (name)")
    db.commit()
    db.close()
    return path, recs


@pytest.mark.parametrize("crlf", [False, True])
def test_fastq_subscript_and_read_getters_in_c(tmp_path, crlf):
    """csrc/fxobj.c FastqCore: fq[i] / fq[name] from prepared statements (fastq.c:454-545), Read.seq / .qual / .quali of a
    plain file from the page cache (read.c:152-167, 237-278) -- no device needed for either, so they are checked here."""
    import pyfastx_amd  # noqa: F401  (binds the Read type)
    from pyfastx_amd import _fxobj
    path, recs = _tiny_fastq(tmp_path, crlf)

    class FQ(_fxobj.FastqCore):
        def _getitem_slow(self, key):
            return ("slow", key)

    fq = FQ()
    fq._counts, fq._phred = len(recs), 0
    assert fq[0] == ("slow", 0)                                  # nothing bound yet: the subclass answers
    assert fq._core_open(path + ".fxi") is True
    for i, (n, s, q) in enumerate(recs):
        r = fq[i]
        assert (r.id, r.name, len(r)) == (i + 1, n, len(s)) and type(r).__name__ == "Read"
        assert fq[n].id == i + 1 and fq[i - len(recs)].name == n
    with pytest.raises(IndexError, match="index out of range"):
        fq[len(recs)]
    with pytest.raises(IndexError):
        fq[-len(recs) - 1]
    with pytest.raises(KeyError, match="nope does not exist in fastq file"):
        fq["nope"]
    assert fq[2.0] == ("slow", 2.0) and fq[np.int64(3)] == ("slow", np.int64(3))
    # the getters stay with the Python methods until the stream is staged (no device here: a made-up handle)
    r = fq[3]
    with pytest.raises(AttributeError):
        r.seq                                                    # _seq_slow -> the Fastq's staged stream, which FQ does not have
    fq._core_stage(1, path)
    for i, (n, s, q) in enumerate(recs):
        r = fq[n]
        assert r.seq == s and r.qual == q and r.quali == [ord(c) - 33 for c in q]
    fq._phred = 64
    assert fq[1].quali == [ord(c) - 64 for c in recs[1][2]]
    # the table this process built, kept on the host (_core_table): fq[i] is six array elements, the name is read from the
    # file when somebody asks -- same objects as the statements give
    import sqlite3
    rows = sqlite3.connect(path + ".fxi").execute("SELECT name, dlen, rlen, soff, qoff FROM read ORDER BY ID").fetchall()
    raw = open(path, "rb").read()
    name_off = np.array([raw.index(b"@" + r[0].encode() + b" ") + 1 for r in rows], dtype=np.int64)
    cols = (name_off, np.array([len(r[0]) for r in rows], dtype=np.int32), np.array([r[1] for r in rows], dtype=np.int32),
            np.array([r[2] for r in rows], dtype=np.int64), np.array([r[3] for r in rows], dtype=np.int64), np.array([r[4] for r in rows], dtype=np.int64))
    with pytest.raises(ValueError):
        fq._core_table(cols[0], cols[1].astype(np.int16)[:5], *cols[2:])   # a column of another width
    assert fq._core_table_rows == 0
    fq._phred = 0
    fq._core_table(*cols)
    assert fq._core_table_rows == len(recs)
    for i, (n, s, q) in enumerate(recs):
        r = fq[i - len(recs)] if i % 2 else fq[i]
        assert type(r).__name__ == "Read" and (r.id, len(r), r._desc_len, r._soff, r._qoff) == (i + 1, len(s), rows[i][1], rows[i][3], rows[i][4])
        assert r.seq == s and r.qual == q and r.quali == [ord(c) - 33 for c in q]
        assert r.name == n and r.name is r.name and repr(r) == "<Read> %s with length of %d" % (n, len(s))
        r.name = "other"
        assert r.name == "other" and fq[n].id == i + 1                 # by name: still the statement
    with pytest.raises(IndexError, match="index out of range"):
        fq[len(recs)]
    # ... and with the names as they were packed for the index file (_core_names), fq[name] is a hash look-up into that table
    packed = np.frombuffer("".join(n for n, _, _ in recs).encode(), dtype=np.uint8)
    offs = np.cumsum([0] + [len(n) for n, _, _ in recs]).astype(np.int64)
    with pytest.raises(ValueError):
        fq._core_names(packed, offs[::-1].copy())
    fq._core_names(packed, offs)
    assert fq._core_names_rows == len(recs)
    import sqlite3 as _sq
    _sq.connect(path + ".fxi").execute("UPDATE read SET soff = -1").connection.commit()    # whoever asks the file now gets nonsense
    for k in range(70):                                           # (the id table is made once fq[name] has been used 64 times and once per 90 reads)
        fq[recs[k % len(recs)][0]]
    for i, (n, s, q) in enumerate(recs):
        key = "".join(n)                                          # (another str object than the one in recs)
        r = fq[key]
        assert (r.id, r.seq, r.qual, r._soff) == (i + 1, s, q, rows[i][3]) and r.name is key
    with pytest.raises(KeyError, match="nope does not exist in fastq file"):
        fq["nope"]
    with pytest.raises(KeyError):
        fq["r"]                                                  # a prefix of every name
    _sq.connect(path + ".fxi").executemany("UPDATE read SET soff = ? WHERE ID = ?", [(rows[i][3], i + 1) for i in range(len(rows))]).connection.commit()
    fq._core_names()
    assert fq._core_names_rows == 0 and fq[recs[2][0]].seq == recs[2][1]
    fq._core_table()
    assert fq._core_table_rows == 0 and fq[1].name == recs[1][0]
    # ... and an object that only has the index file reads the four integer columns from it in one pass once fq[i] has been
    # asked for often enough (64 times and 1/22 of the reads); the name of such a read comes from the file's statement
    fq._core_table_cap = 1000
    for k in range(70):
        assert fq[k % len(recs)].id == k % len(recs) + 1
    assert fq._core_table_rows == len(recs)
    for i, (n, s, q) in enumerate(recs):
        r = fq[i]
        assert (r.id, r._name_len, len(r), r._desc_len, r._soff, r._qoff) == (i + 1, 0, len(s), rows[i][1], rows[i][3], rows[i][4])
        assert r.seq == s and r.qual == q and r.name == n and r.name is r.name
    fq._core_open(path + ".fxi")                                 # bound again: what was known of the file goes
    assert fq._core_table_rows == 0
    # ... and asked by NAME often enough, it reads the names too (one more pass) and answers from a table of ids of its own
    for k in range(70):
        assert fq[recs[k % len(recs)][0]].id == k % len(recs) + 1
    assert fq._core_names_rows == len(recs) == fq._core_table_rows
    _sq.connect(path + ".fxi").execute("UPDATE read SET rlen = 1").connection.commit()      # (the file is not asked any more)
    for i, (n, s, q) in enumerate(recs):
        r = fq["".join(n)]
        assert (r.id, len(r), r.seq, r.qual) == (i + 1, len(s), s, q)
    with pytest.raises(KeyError, match="zzz does not exist in fastq file"):
        fq["zzz"]
    _sq.connect(path + ".fxi").executemany("UPDATE read SET rlen = ? WHERE ID = ?", [(rows[i][2], i + 1) for i in range(len(rows))]).connection.commit()
    fq._core_open(path + ".fxi")
    assert fq._core_names_rows == 0 == fq._core_table_rows
    fq._core_stage(0)                                            # Blob.close(): back to the Python methods
    with pytest.raises(AttributeError):
        fq[0].seq
    assert fq._core_open(None) is False and fq[0] == ("slow", 0)


def test_sequence_getters_from_the_page_cache(tmp_path):
    """csrc/fxobj.c seq_fast on a plain file: pread + despace / upper / complement / reverse in C (util.c:157-269), equal to
    the oracle's getters for every slice of a line-regular record -- LF and CRLF, lower case, IUPAC codes, a byte of 200."""
    import pyfastx_amd
    from pyfastx_amd import _fxobj, api
    import fxoracle
    fxoracle.lib()
    oracle_rc = fxoracle.revcomp                              # mode: 1 reverse, 2 complement (oracle/fx_oracle.c: fxo_revcomp)
    rng = np.random.default_rng(3)
    for el, nl in ((1, b"\n"), (2, b"\r\n")):
        bases = bytes(rng.choice(list(b"ACGTacgtNnRYKMBDHVUu\xc8"), 333).astype(np.uint8))
        bpl = 50
        lines = [bases[i:i + bpl] for i in range(0, len(bases), bpl)]
        hdr = b">s1 test" + nl
        path = str(tmp_path / ("t%d.fa" % el))
        with open(path, "wb") as f:
            f.write(hdr + nl.join(lines) + nl)

        class FA(_fxobj.FastaCore):
            pass

        for upper in (0, 1):
            fa = FA()
            fa._core_upper = upper
            s = api.Sequence(fa, 1, "s1", len(hdr), len(bases) + len(lines) * el, len(bases), bpl + el, el, 1, len(hdr) - 1 - el)
            s._reg = 1
            fa._core_stage(1, path)
            assert fa._core_fd >= 0
            want = bases.upper() if upper else bases
            # (bytes.upper leaves 0xC8 alone, as remove_space_uppercase leaves bytes >= 128 alone here)
            for a, b in [(0, 1), (0, 50), (49, 51), (3, 333), (100, 250), (332, 333), (17, 18)] + [tuple(sorted(rng.integers(0, 334, 2))) for _ in range(60)]:
                if a == b or (a == 0 and b == 333):
                    continue
                sl = s[a:b]
                w = want[a:b]
                assert sl.seq.encode("latin-1") == w
                assert sl.reverse.encode("latin-1") == w[::-1]
                assert sl.complement.encode("latin-1") == oracle_rc(w, 2)
                assert sl.antisense.encode("latin-1") == oracle_rc(w, 3)
            with pytest.raises(TypeError):
                del s._fa
            fa._core_stage(0)
            assert fa._core_fd == -1 and fa._core_handle == 0


def test_pinned_buffer_owner_and_name_helpers():
    """csrc/fxobj.c: PinnedBuf gives its block back when the last view dies; ids_of_names resolves names through a dict with
    an identity cache in front; pack_names packs a list for fx_names_lookup."""
    import ctypes as C
    from pyfastx_amd import _fxobj
    freed = []
    FREE = C.CFUNCTYPE(None, C.c_void_p)
    cb = FREE(lambda p: freed.append(p))
    raw = C.create_string_buffer(64)
    pb = _fxobj.PinnedBuf(C.addressof(raw), 64, C.cast(cb, C.c_void_p).value)
    a = np.frombuffer(pb, dtype=np.uint8)
    a[:] = np.arange(64)
    v = a[10:20]
    del a, pb
    assert not freed and v.tolist() == list(range(10, 20))
    del v
    assert freed == [C.addressof(raw)]
    names = ["chr%d" % i for i in range(50)]
    index = {n: i for i, n in enumerate(names)}
    ids = np.random.default_rng(0).integers(0, 50, 10_000)
    q = [names[i] for i in ids] + ["".join(["chr", "7"])]         # same objects again and again, and one equal but distinct
    out = np.empty(len(q), dtype=np.int64)
    assert _fxobj.ids_of_names(q, index, out) == -1 and out[:-1].tolist() == ids.tolist() and out[-1] == 7
    assert _fxobj.ids_of_names(q[:5] + ["nope"] + q[5:], index, np.empty(len(q) + 1, dtype=np.int64)) == 5
    with pytest.raises(ValueError):
        _fxobj.ids_of_names(q, index, np.empty(3, dtype=np.int64))
    b, o = _fxobj.pack_names(["ab", b"cde", "", "é"])
    assert b == b"abcde\xc3\xa9" + b"\0" * 16 and np.frombuffer(o, dtype=np.int64).tolist() == [0, 2, 5, 5, 7]
    with pytest.raises(ValueError):
        _fxobj.pack_names(["ok", "\udcff"])


# ---------------------------------------------------------------------------------- round 4: windows of one device (out of core)
class _FakeWindow:
    """What windows.WindowedFasta asks of a staged window, answered by the numpy restatement of the shard kernels
    (shard_ref.local_scan) over the bytes [lo, hi) of `raw`: the host logic of the windowed build on the CPU."""

    staged = []

    def __init__(self, raw, lo, hi):
        self.raw, self.lo, self.hi = raw, lo, hi
        _FakeWindow.staged.append((lo, hi))

    def fasta_build(self, full_name=False):
        self.rows, self.S = shard_ref.local_scan(self.raw, self.lo, self.hi, bool(full_name))
        self._n_fasta = len(self.rows["hoff"])

        class R:
            n_seq = self._n_fasta
        return R

    def shard_summary(self):
        return self.S

    def fasta_table(self, n):
        return {k: np.array(v[:n], dtype=np.int64) for k, v in self.rows.items() if k != "reg"}

    def fasta_line_regular(self, n):
        return np.array(self.rows["reg"][:n], dtype=np.int32)

    def read_bytes(self, off, n):
        a, b = max(int(off), self.lo), min(int(off) + int(n), self.hi)
        got = self.raw[a:b] if b > a else b""
        return got + b"\0" * (max(int(n), 0) - len(got)) if int(off) >= self.lo else got

    def fetch_ranges(self, off, blen, slen, flags=0, **kw):
        assert flags == 8
        outs = [self.raw[max(int(o), self.lo):min(int(o) + int(l), self.hi)] for o, l in zip(off, blen)]
        offs = np.zeros(len(outs) + 1, dtype=np.int64)
        np.cumsum([int(x) for x in slen], out=offs[1:])
        buf = np.zeros(max(int(offs[-1]), 1), dtype=np.uint8)
        for i, b in enumerate(outs):
            buf[offs[i]:offs[i] + len(b)] = np.frombuffer(b, dtype=np.uint8)
        return buf, offs, np.array([len(b) for b in outs], dtype=np.int64)

    def close(self):
        pass


def test_windowed_fasta_build_equals_the_whole_file(oracle, tmp_path, monkeypatch):
    """pyfastx_amd/windows.py: a FASTA file built window after window -- summaries kept, every window's last record finished
    from the windows behind it, names that cross a cut put together -- gives the rows and names of the whole-file build, for
    every edge input at every number of windows (the kernels' part played by shard_ref on the CPU)."""
    from pyfastx_amd import _lib, windows
    cases = [(k, v["text"].encode()) for k, v in load_golden("fasta_edge").items() if not k.endswith(":upper")]
    rng = np.random.default_rng(11)
    long_hdr = b">" + bytes(rng.choice(list(b"abcdefgh"), 300).astype(np.uint8)) + b" d\nACGT\nAC\n>z\nA\n"
    cases += [("long_name", long_hdr), ("long_name_full", long_hdr)]
    for name, raw in cases:
        if len(raw) < 8:
            continue
        p = str(tmp_path / "w.fa")
        open(p, "wb").write(raw)
        full = name.endswith("_full")
        want = _expect(oracle, raw, full)
        recs, _ = oracle.fasta_index(raw, full_name=full)
        want_names = [raw[r["name_off"]:r["name_off"] + r["name_len"]] for r in recs]
        monkeypatch.setattr(_lib.Blob, "from_file_range", classmethod(lambda cls, path, off, length, halo=0, device=0: _FakeWindow(raw, off, off + length)))
        for nwin in (1, 2, 3, 5, 9, 17):
            if nwin > len(raw):
                continue
            _FakeWindow.staged = []
            wf = windows.WindowedFasta(p, device=0, full_name=full, window=-(-len(raw) // nwin), capacity=2)
            got = {k: [int(x) for x in wf.table[k]] for k in want}
            assert got == want, (name, nwin)
            assert [bytes(x) for x in wf.table["names"]] == want_names, (name, nwin)
            assert len(_FakeWindow.staged) == wf.windows and len(wf.cache.lru) <= 2      # one pass, two windows resident at most


def test_window_plan_and_budget(monkeypatch, tmp_path):
    from pyfastx_amd import windows
    assert windows.parse_size("64M") == 64 << 20 and windows.parse_size("1.5G") == int(1.5 * (1 << 30)) and windows.parse_size("4096") == 4096
    assert windows.parse_size("200GiB") == 200 << 30 and windows.parse_size("8k") == 8192
    p = str(tmp_path / "f.fa")
    open(p, "wb").write(b">a\n" + b"ACGT\n" * 100000)
    monkeypatch.setenv("FX_HBM_BUDGET", "1G")
    assert windows.plan(p, 0, 1.15) is None                     # fits
    monkeypatch.setenv("FX_HBM_BUDGET", "256K")
    size, kind, win, cap = windows.plan(p, 0, 1.0)
    assert size == 500003 and kind == 0 and win == 65536 and cap == 4
    # a BGZF file: taken to fit from the ratio of its first members when that estimate is under a quarter of the budget
    # (no walk over all members: _lib.stream_size is not asked); else, and for exact=True, the exact size decides
    import gzip
    from pyfastx_amd import _lib, synth
    raw = b">a\n" + b"ACGTTGCA" * 40000 + b"\n>b\n" + bytes(np.random.default_rng(5).integers(65, 85, 300000, dtype=np.uint8)) + b"\n"
    z = str(tmp_path / "f.fa.gz")
    open(z, "wb").write(synth.bgzf_compress(raw))
    r = windows.bgzf_head_ratio(z)
    assert r is not None and abs(r - len(raw) / os.path.getsize(z)) < 0.35 * r
    g = str(tmp_path / "g.fa.gz")
    open(g, "wb").write(gzip.compress(raw))
    assert windows.bgzf_head_ratio(g) is None and windows.bgzf_head_ratio(p) is None and windows.bgzf_head_ratio(str(tmp_path / "absent")) is None
    asked = []
    real = _lib.stream_size
    monkeypatch.setattr(_lib, "stream_size", lambda path: (asked.append(path), real(path))[1])
    monkeypatch.setenv("FX_HBM_BUDGET", "1G")
    assert windows.plan(z, 0, 1.15) is None and asked == []     # estimate x 4 under the budget
    assert windows.plan(z, 0, 1.15, exact=True) is None and asked == [z]
    monkeypatch.setenv("FX_HBM_BUDGET", "2M")                   # estimate x 4 over the budget, the stream itself under it: the exact size is asked for
    assert windows.plan(z, 0, 1.0) is None and asked == [z, z]
    monkeypatch.setenv("FX_HBM_BUDGET", "256K")
    size, kind, win, cap = windows.plan(z, 0, 1.0)
    assert size == len(raw) and kind == 1 and win == 65536


def test_native_framing_helpers_write_the_bytes_of_the_python_ones():
    """csrc/libfxsynth.so (set-up of the C4 inputs: BGZF members / one gzip stream deflated by plain threads) against
    synth.bgzf_compress / synth.gzip_single_stream, byte for byte, and against gzip itself."""
    import gzip
    from pyfastx_amd import synth
    if synth._native() is None:
        pytest.skip("libfxsynth.so not built")
    rng = np.random.default_rng(3)
    raw = np.frombuffer(b"ACGTNacgt\n", dtype=np.uint8)[rng.integers(0, 10, 700_000)]
    for n in (0, 1, 65279, 65280, 65281, 700_000):
        r = raw[:n]
        a = bytes(synth.bgzf_compress_parallel(r))
        assert a == synth.bgzf_compress(r.tobytes()) and gzip.decompress(a) == r.tobytes(), n
        g = bytes(synth.gzip_single_stream_parallel(r, piece=1 << 18))
        assert g == bytes(synth.gzip_single_stream(r, piece=1 << 18)) and gzip.decompress(g) == r.tobytes(), n


@pytest.mark.parametrize("shape", ["illumina", "short_names", "crlf_long"])
def test_fxi_size_estimate_from_the_head_of_a_fastq(tmp_path, shape):
    """fxi.estimate_fastq_index_bytes (room set aside for the index file while the input is staged) against the size
    of the index file the page loaders really write for the same records: within 3 %."""
    from pyfastx_amd import fxi
    rng = np.random.default_rng(4)
    n = 120_000
    if shape == "illumina":
        recs = [b"@SYN:1:FC:1:%04d:%05d:%09d 1:N:0:ACGT\n%s\n+\n%s\n" % (i % 9999, (i * 7919) % 100000, i, b"A" * 150, b"I" * 150) for i in range(n)]
    elif shape == "short_names":
        recs = [b"@r%d\n%s\n+\n%s\n" % (i, b"ACGT" * 9, b"IIII" * 9) for i in range(n)]
    else:
        recs = [b"@%s/%d some comment here\r\n%s\r\n+\r\n%s\r\n" % (b"x" * 80, i, b"ACGT" * 60, b"IIII" * 60) for i in range(n)]
    raw = b"".join(recs)
    p = tmp_path / "e.fq"
    p.write_bytes(raw)
    est = fxi.estimate_fastq_index_bytes(str(p), head=1 << 20)
    # the real thing: rows as the index builder gives them, written by the host page loaders
    names, soff, qoff, dlen, rlen = [], [], [], [], []
    pos = 0
    eol = 2 if shape == "crlf_long" else 1
    for r in recs:
        h = r.index(b"\n")
        line = r[1:h + 1 - eol]
        names.append(line.split()[0])
        dlen.append(len(line))
        s2 = r.index(b"\n", h + 1)
        rlen.append(s2 - h - eol)
        soff.append(pos + h + 1)
        q = r.index(b"\n", s2 + 1)
        qoff.append(pos + q + 1)
        pos += len(r)
    cols = {k: np.asarray(v, dtype=np.int64) for k, v in (("dlen", dlen), ("rlen", rlen), ("soff", soff), ("qoff", qoff))}
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(x) for x in names], out=offs[1:])
    order = np.array(sorted(range(n), key=lambda i: names[i]), dtype=np.int64)
    out = str(tmp_path / "e.fxi")
    fxi.write_fastq_bulk(out, np.frombuffer(b"".join(names), dtype=np.uint8), offs, cols, int(cols["rlen"].sum()), order=order).close()
    real = os.path.getsize(out)
    assert abs(est - real) <= 0.03 * real, (est, real)
    # a head that is not made of four-line records gives no estimate
    q = tmp_path / "odd.fq"
    q.write_bytes(b"@a\nAC\nGT\n+\nII\nII\n" * 1000)
    assert fxi.estimate_fastq_index_bytes(str(q)) is None


def test_bench_stdout_carries_the_json_line_only():
    """bench.py's contract with the driver: ONE JSON line on stdout.  What libraries print there from C (gloo's "[Gloo] Rank r is
    connected to 7 peer ranks" of every rank, RCCL under NCCL_DEBUG) and what children inherit goes to stderr once
    _stdout_for_the_line_only() has run; _emit() writes the line to the real stdout.  Also the instruction-issue roofline's arithmetic."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, %r); import bench\n"
            "bench._stdout_for_the_line_only(); print('python noise'); os.system('echo child noise')\n"
            "os.write(1, b'C-level noise\\n'); bench._emit({'metric': 'm', 'value': 1.5})\n") % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout == '{"metric": "m", "value": 1.5}\n'
    for noise in ("python noise", "child noise", "C-level noise"):
        assert noise in out.stderr
    sys.path.insert(0, root)
    try:
        import bench
        r = bench._issue_roofline("k_fastq_lines", 4096 * 1000, 1.0)
        # instructions per granule x 4 cycles of a SIMD / (1024 SIMDs x 2.4 GHz): 1000 granules of 473 instructions = 0.77 us
        assert r["granules"] == 1000 and r["bound"] == "valu issue"
        assert abs(r["floor_ms"] - 1000 * 473 * 4 / (1024 * 2.4e9) * 1e3) < 1e-3
        assert 0 < r["frac"] <= 1
    finally:
        sys.path.remove(root)


def test_large_idle_blocks_policy():
    """ScratchPool's rule for the blobs of closed streams (fx_scratch_policy, csrc/fxgpu.hip): a block
    stays if it fits under the cap beside the others; the smallest go first to make room; a newcomer smaller than everything
    that would have to go is the one that goes; nothing stays with a cap of 0."""
    import ctypes as C
    from pyfastx_amd import _lib
    L = _lib.lib()
    G = 1 << 30

    def ask(idle, cap, keep):
        a = (C.c_int64 * max(len(idle), 1))(*idle)
        ev = (C.c_int32 * max(len(idle), 1))()
        r = L.fx_scratch_policy(a, len(idle), cap, keep, ev)
        return r, [int(ev[i]) for i in range(len(idle))]
    assert ask([], 35 * G, 144 * G) == (1, [])                                  # the first large blob of a process is kept
    assert ask([35 * G], 40 * G, 144 * G) == (1, [0])                           # open / close / open of two 35 GB-class sizes: both stay, no hipFree at all
    assert ask([35 * G, 40 * G, 45 * G], 50 * G, 144 * G) == (1, [1, 0, 0])     # over the cap: the smallest goes
    assert ask([35 * G, 40 * G, 60 * G], 30 * G, 144 * G) == (0, [0, 0, 0])     # the newcomer is the smallest: it goes
    assert ask([20 * G, 25 * G, 90 * G], 100 * G, 144 * G) == (1, [1, 1, 1])    # every idle block is smaller than the newcomer and all must go: they go
    assert ask([20 * G, 25 * G, 120 * G], 100 * G, 144 * G) == (0, [0, 0, 0])   # the 120 GB block would have to go as well, and it is larger: the newcomer goes
    assert ask([35 * G], 35 * G, 0) == (0, [0])                                 # FX_SCRATCH_KEEP_BIG_MB=0: nothing idles
    assert ask([35 * G], 200 * G, 144 * G) == (0, [0])                          # larger than the cap itself
