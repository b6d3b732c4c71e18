"""-m gpu: the `.fxi` b-trees formatted ON THE DEVICE (csrc/fx_fxi_dev.hpp, fx_fxi_dev_sort / fx_fxi_dev_write) against
SQLite's own integrity check, the host page loaders (fx_fxi_bulk_rows / fx_fxi_bulk_index) and the golden rows of the
reference (fastq.c:29-60, 136-171; index.c:178-207, 239-251, 363)."""
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
def dev_route(monkeypatch):
    monkeypatch.setenv("FX_FXI_DEV_MIN", "0")
    monkeypatch.delenv("FX_FXI_HOST", raising=False)


def _check(path):
    db = sqlite3.connect(path)
    ok = db.execute("PRAGMA integrity_check").fetchall()
    idx = sorted(r[0] for r in db.execute("SELECT name FROM sqlite_master WHERE type='index'"))
    db.close()
    return ok, idx


def _rows(path, table):
    db = sqlite3.connect(path)
    db.text_factory = bytes
    rows = db.execute("SELECT * FROM %s ORDER BY ID" % table).fetchall()
    db.close()
    return rows


def test_fixture_files_through_the_device_route(fx, tmp_path, dev_route):
    for fn in ("test.fa", "test.fq"):
        shutil.copy(os.path.join(DATA, fn), tmp_path / fn)
    g = load_golden("fasta_fixture")["test.fa"]
    fa = fx.Fasta(str(tmp_path / "test.fa"), full_index=True)
    assert fa.index_phases is not None                       # the device route was taken
    assert _check(str(tmp_path / "test.fa.fxi")) == ([("ok",)], ["chromidx", "seqidx"])
    db = sqlite3.connect(str(tmp_path / "test.fa.fxi"))
    assert [list(r) for r in db.execute("SELECT * FROM seq")] == g["seq"]
    assert "chromidx" in db.execute("EXPLAIN QUERY PLAN SELECT * FROM seq WHERE chrom=?", ("x",)).fetchall()[0][-1]
    for row in g["seq"][::9]:
        assert db.execute("SELECT ID FROM seq WHERE chrom=?", (row[1],)).fetchone()[0] == row[0]
    db.close()
    gq = load_golden("fastq_fixture")["test.fq"]
    fq = fx.Fastq(str(tmp_path / "test.fq"), full_index=True)
    assert fq.index_phases is not None
    assert _check(str(tmp_path / "test.fq.fxi")) == ([("ok",)], ["readidx"])
    db = sqlite3.connect(str(tmp_path / "test.fq.fxi"))
    assert [list(r) for r in db.execute("SELECT * FROM read")] == gq["read"]
    db.close()
    assert fq[7].seq == fx.Fastq(str(tmp_path / "test.fq"))[7].seq


def _many_fastq(n, seed, short=False):
    rng = np.random.default_rng(seed)
    ids = rng.permutation(n).tolist()
    if short:                                                # tiny cells: more than 64 rows on a page, names of 1..7 bytes
        return ids, b"".join(b"@%x\nA\n+\nI\n" % (i + 1) for i in ids)
    return ids, b"".join(b"@SRR8539271.%d len=%d\n%s\n+\n%s\n" % (i + 1, i % 97, b"ACGTNACGTA" * (1 + i % 3), b"IIIIIHHHHH" * (1 + i % 3)) for i in ids)


@pytest.mark.parametrize("shape", ["usual", "short", "slabs", "pwrite"])
def test_device_pages_equal_the_host_loader_row_for_row(fx, tmp_path, monkeypatch, shape):
    """The same file indexed through the device route and through the host page loaders: SQLite accepts both, every row
    of `read` is the same, the index gives the names in order, by-name access works."""
    n = 150_000 if shape != "short" else 300_000
    ids, raw = _many_fastq(n, 5, short=(shape == "short"))
    rows = {}
    for route in ("dev", "host"):
        p = tmp_path / ("many_%s.fq" % route)
        p.write_bytes(raw)
        monkeypatch.setenv("FX_FXI_DEV_MIN", "0")
        if route == "host":
            monkeypatch.setenv("FX_FXI_HOST", "1")
        else:
            monkeypatch.delenv("FX_FXI_HOST", raising=False)
        if shape == "slabs":
            monkeypatch.setenv("FX_FXI_SLAB_MB", "1")         # 256 pages per slab: dozens of slabs per tree
        if shape == "pwrite":
            monkeypatch.setenv("FX_FXI_NO_MMAP", "1")         # no mapping of the file (what a file system without room for it gets): every page through pwrite
        fq = fx.Fastq(str(p))
        assert (fq.index_phases is not None) == (route == "dev")
        assert len(fq) == n and _check(str(p) + ".fxi") == ([("ok",)], ["readidx"])
        rows[route] = _rows(str(p) + ".fxi", "read")
        db = sqlite3.connect(str(p) + ".fxi")
        db.text_factory = bytes
        got = [r[0] for r in db.execute("SELECT name FROM read INDEXED BY readidx ORDER BY name")]
        assert got == sorted(r[1] for r in rows[route])
        assert db.execute("SELECT count(*), min(ID), max(ID) FROM read").fetchone() == (n, 1, n)
        db.close()
        rng = np.random.default_rng(11)
        for j in rng.integers(0, n, 25).tolist():
            name = ("%x" % (ids[j] + 1)) if shape == "short" else "SRR8539271.%d" % (ids[j] + 1)
            assert fq[name].id == j + 1
    assert rows["dev"] == rows["host"]


def test_fasta_table_with_every_integer_width(fx, tmp_path, dev_route):
    """`seq` has seven integer columns; records of 0, 1, 127, 128, 40 000 bases put 0 / 1 (serial types 8 / 9) and one-,
    two- and three-byte integers into them.  Rows equal the INSERT path's."""
    rng = np.random.default_rng(3)
    parts = []
    lens = [0, 1, 127, 128, 40_000] + rng.integers(0, 600, 20_000).tolist()
    for i, L in enumerate(lens):
        parts.append(b">r%d some description %d\n" % (i, i))
        s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, L)].tobytes()
        parts += [s[k:k + 60] + b"\n" for k in range(0, L, 60)]
    raw = b"".join(parts)
    out = {}
    for route in ("dev", "ins"):
        p = tmp_path / ("w_%s.fa" % route)
        p.write_bytes(raw)
        fa = fx.Fasta(str(p), key_func=(None if route == "dev" else (lambda h: h.split()[0])))      # key_func: the INSERT path
        assert (fa.index_phases is not None) == (route == "dev")
        assert _check(str(p) + ".fxi") == ([("ok",)], ["chromidx"])
        out[route] = _rows(str(p) + ".fxi", "seq")
        assert fa["r3"].seq == fa[3].seq and len(fa) == len(lens)
    assert out["dev"] == out["ins"]


def test_special_cases_fall_back(fx, tmp_path, dev_route):
    # duplicate names: the reference's CREATE UNIQUE INDEX fails and is ignored (index.c:363-366) -> table only, no index
    p = tmp_path / "dup.fa"
    p.write_bytes(b">a 1\nACGT\n>b\nGG\n>a 2\nTTTT\n")
    fa = fx.Fasta(str(p))
    assert fa.index_phases is not None and fa.index_phases["index_to_file"] == 0.0
    assert _check(str(p) + ".fxi") == ([("ok",)], []) and len(fa) == 3 and fa["a"].id == 1
    # a name too long for an in-page index entry / for a table page: the device route declines, the host routes take over
    for L, tag in ((1500, "mid"), (5000, "big")):
        p = tmp_path / ("long_%s.fa" % tag)
        p.write_bytes(b">" + b"N" * L + b"\nACGT\n>short\nGGCC\n")
        fa = fx.Fasta(str(p))
        assert fa.index_phases is None
        assert _check(str(p) + ".fxi") == ([("ok",)], ["chromidx"])
        assert fa["N" * L].seq == "ACGT" and fa["short"].id == 2
    # non-ASCII header bytes are stored as they are in the file
    p = tmp_path / "latin.fa"
    p.write_bytes(b">caf\xc3\xa9 x\nACGT\n>plain\nGG\n")
    fa = fx.Fasta(str(p))
    db = sqlite3.connect(str(p) + ".fxi")
    assert db.execute("SELECT CAST(chrom AS BLOB) FROM seq WHERE ID=1").fetchone()[0] == b"caf\xc3\xa9"
    db.close()
    assert fa["café"].seq == "ACGT"
    # an empty FASTQ-like input cannot be opened at all (fastq.c:300-304); one read: everything in the root pages
    p = tmp_path / "one.fq"
    p.write_bytes(b"@only\nAC\n+\nII\n")
    fq = fx.Fastq(str(p))
    assert _check(str(p) + ".fxi") == ([("ok",)], ["readidx"]) and fq["only"].seq == "AC"


def test_index_file_that_crosses_the_pending_byte_page(fx, tmp_path, dev_route):
    """More than 2^30 bytes of pages: the page that holds byte 2^30 must stay untouched (SQLite's locking page) in both
    trees' page sequences.  560 k records with 960-byte names: four cells per page, 140 k + 140 k leaves = 1.15 GB."""
    n = 560_000
    pad = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789", dtype=np.uint8)
    rng = np.random.default_rng(9)
    body = pad[rng.integers(0, pad.size, (n, 950))]
    rec = np.empty((n, 950 + 12 + 6), dtype=np.uint8)       # ">" + 10 digits + "_" + 950 letters + "\nACGT\n"
    rec[:, 0] = ord(">")
    idx = rng.permutation(n)
    for k in range(10):
        rec[:, 10 - k] = 48 + (idx // 10 ** k) % 10
    rec[:, 11] = ord("_")
    rec[:, 12:962] = body
    rec[:, 962:] = np.frombuffer(b"\nACGT\n", dtype=np.uint8)
    p = tmp_path / "wide.fa"
    rec.tofile(str(p))
    fa = fx.Fasta(str(p))
    assert fa.index_phases is not None and len(fa) == n
    assert os.path.getsize(str(p) + ".fxi") > (1 << 30) + (64 << 20)
    assert _check(str(p) + ".fxi") == ([("ok",)], ["chromidx"])
    for j in (0, 1, n // 2, n - 1):
        name = rec[j, 1:962].tobytes().decode()
        assert fa[name].id == j + 1 and fa[j].name == name


def test_room_set_aside_while_staging(fx, tmp_path, dev_route, monkeypatch):
    """fxi.presize_fastq: the index file is created before the input is staged and grown to its estimated size in the
    background; the build fills it and cuts it to what it used -- the same rows as without, no slack behind the database,
    and an over- or under-estimate changes nothing but time."""
    from pyfastx_amd import fxi
    n = 200_000
    ids, raw = _many_fastq(n, 8)
    rows = {}
    for tag, scale in (("plain", None), ("presized", 1.0), ("too_small", 0.3), ("too_large", 2.5)):
        p = tmp_path / ("pre_%s.fq" % tag)
        p.write_bytes(raw)
        if scale is None:
            monkeypatch.setenv("FX_FXI_NO_PRESIZE", "1")
        else:
            monkeypatch.delenv("FX_FXI_NO_PRESIZE", raising=False)
            monkeypatch.setenv("FX_FXI_PRESIZE_MIN", "0")
            real = fxi.estimate_fastq_index_bytes
            monkeypatch.setattr(fxi, "estimate_fastq_index_bytes", lambda path, full_name=False, _r=real, _s=scale: int(_r(path, full_name) * _s))
        fq = fx.Fastq(str(p))
        assert fq.index_phases is not None and len(fq) == n
        assert _check(str(p) + ".fxi") == ([("ok",)], ["readidx"])
        db = sqlite3.connect(str(p) + ".fxi")
        assert db.execute("PRAGMA page_count").fetchone()[0] * 4096 == os.path.getsize(str(p) + ".fxi")
        db.close()
        rows[tag] = _rows(str(p) + ".fxi", "read")
        assert fq["SRR8539271.%d" % (ids[77] + 1)].id == 78
        monkeypatch.undo()
        monkeypatch.setenv("FX_FXI_DEV_MIN", "0")
    assert rows["plain"] == rows["presized"] == rows["too_small"] == rows["too_large"]
    # the build that declines the device route (a name too long for an index entry) leaves no early file behind
    monkeypatch.setenv("FX_FXI_PRESIZE_MIN", "0")
    p = tmp_path / "long.fq"
    p.write_bytes(b"".join(b"@%s_%d\nACGT\n+\nIIII\n" % (b"N" * 1500, i) for i in range(50)))
    fq = fx.Fastq(str(p))
    assert fq.index_phases is None and _check(str(p) + ".fxi") == ([("ok",)], ["readidx"]) and len(fq) == 50
    # an input that is no FASTQ at all: the constructor raises before anything is created; a truncated one builds what is there
    p = tmp_path / "cut.fq"
    p.write_bytes(raw[: len(raw) // 2 + 7])
    fq = fx.Fastq(str(p))
    assert _check(str(p) + ".fxi")[0] == [("ok",)] and len(fq) > 0


def test_memory_mapped_right_after_the_build_stays(fx, tmp_path, dev_route, monkeypatch):
    """The index file's mappings are taken down by a thread of the library after the build has returned, piece by piece
    (a munmap of the whole file at once holds the process's mm lock for as long as it runs).  Every address is given back
    ONCE: arrays the caller maps while that thread is still at work -- the allocator hands out the addresses just
    freed -- keep their pages.  (The first piecewise form went over the whole area once more at the end and took a numpy
    array of bench.py with it: a SIGSEGV in the next copy into it.)"""
    import mmap
    import time
    monkeypatch.setenv("FX_FXI_UNMAP_PAUSE_US", "3000")      # the library's thread pauses between its pieces: it is at work while the arrays below are made
    n = 1_500_000                                            # an index file of ~150 MB: some twenty mappings of 8 MiB
    ids, raw = _many_fastq(n, 8)
    for rep in range(3):
        p = tmp_path / ("unmap_%d.fq" % rep)
        p.write_bytes(raw)
        fq = fx.Fastq(str(p))
        maps = []
        for k in range(300):                                 # anonymous mappings of 2 MiB, made while the thread is at work: they fit the holes it leaves
            m = mmap.mmap(-1, 2 << 20)
            m[0:8] = m[-8:] = (rep * 1000 + k).to_bytes(8, "little")
            maps.append(m)
            time.sleep(0.0003)
        assert fq.index_phases is not None and len(fq) == n
        time.sleep(0.2)
        assert [int.from_bytes(m[0:8], "little") + int.from_bytes(m[-8:], "little") for m in maps] == [2 * (rep * 1000 + k) for k in range(300)]
        for m in maps:
            m.close()
        del fq
