"""-m gpu: ONE `.fxi` from SEVERAL handles with every page formatted on a device (fx_fxi_part_* / fx_fxi_join_*, round 6):
the byte-range shards of a FASTQ file as logical ranks of one process (shard.write_fastq_index_parts), as one process per
rank over a gloo process group on one GPU (ShardedFastq.write_index: the shape the 8-GPU run has, RCCL in place of gloo),
and as windows of a stream larger than its HBM budget (windows.WindowedFastq.write_index).  Every file must pass SQLite's
integrity check and hold, row for row, what the single-device file of the same input holds (itself pinned against the
host loader and the reference's golden rows in test_gpu_fxi_dev.py); the names come out of `readidx` in order.
Replaces fastq.c:29-60, 136-171 for a sharded build."""
import multiprocessing as mp
import os
import socket
import sqlite3
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def fx():
    import pyfastx_amd
    from pyfastx_amd import _lib
    assert _lib.lib().fx_device_count() >= 1
    return pyfastx_amd


def _fastq(n, seed, dup=False):
    rng = np.random.default_rng(seed)
    ids = rng.permutation(n).tolist()
    if dup:
        ids[n // 2] = ids[3]                                    # one name twice: no UNIQUE INDEX (fastq.c:152-156)
    return b"".join(b"@SRR8539271.%d len=%d\n%s\n+\n%s\n" % (i + 1, i % 97, b"ACGTNACGTA" * (1 + i % 3), b"IIIIIHHHHH" * (1 + i % 3)) for i in ids)


def _db(path):
    db = sqlite3.connect(path)
    db.text_factory = bytes
    return db


def _whole(path):
    db = _db(path)
    out = {"check": db.execute("PRAGMA integrity_check").fetchall(),
           "index": sorted(r[0].decode() for r in db.execute("SELECT name FROM sqlite_master WHERE type='index'")),
           "read": db.execute("SELECT * FROM read ORDER BY ID").fetchall(),
           "stat": db.execute("SELECT * FROM stat").fetchall()}
    if out["index"]:
        out["by_name"] = [r[0] for r in db.execute("SELECT name FROM read INDEXED BY readidx ORDER BY name")]
    db.close()
    return out


def _single(fx, tmp_path, raw, monkeypatch):
    p = tmp_path / "single.fq"
    p.write_bytes(raw)
    monkeypatch.setenv("FX_FXI_DEV_MIN", "0")
    fq = fx.Fastq(str(p))
    assert fq.index_phases is not None
    return _whole(str(p) + ".fxi")


def _logical_ranks(path, world):
    from pyfastx_amd import _lib, shard
    size, _ = _lib.stream_size(path)
    cores = []
    for r in range(world):
        lo, hi = size * r // world, size * (r + 1) // world
        b = _lib.Blob.from_file_range(path, lo, hi - lo, 0)
        cores.append(b.fastq_scan())
        b.close()
    table = np.array(cores, dtype=np.int64)
    return [shard.ShardedFastq(path, r, world, gather=lambda mine, t=table: t) for r in range(world)]


@pytest.mark.parametrize("world,n", [(2, 150_000), (3, 150_000), (3, 5), (7, 40), (4, 3000)])
def test_logical_ranks_write_the_single_device_rows(fx, tmp_path, monkeypatch, world, n):
    """n = 5 / 40: one leaf in all (it lives in the root page) and ranks without a read; 3000: a few leaves per rank."""
    from pyfastx_amd import shard
    raw = _fastq(n, world)
    want = _single(fx, tmp_path, raw, monkeypatch)
    p = tmp_path / "parts.fq"
    p.write_bytes(raw)
    ranks = _logical_ranks(str(p), world)
    assert sum(r.n_local for r in ranks) == n
    out = str(p) + ".fxi"
    assert shard.write_fastq_index_parts(ranks[::-1], out) == n          # (any order of the list: the parts are taken by rank)
    got = _whole(out)
    assert got["check"] == [(b"ok",)] and got["index"] == ["readidx"]
    assert got["read"] == want["read"] and got["stat"] == want["stat"]
    assert got["by_name"] == sorted(r[1] for r in got["read"]) == want["by_name"]
    db = _db(out)
    for j in np.random.default_rng(1).integers(0, n, 20).tolist():
        name = got["read"][j][1].decode()
        assert db.execute("SELECT ID FROM read WHERE name=?", (name,)).fetchone()[0] == j + 1
        assert b"readidx" in db.execute("EXPLAIN QUERY PLAN SELECT ID FROM read WHERE name=?", (name,)).fetchall()[0][-1]
    db.close()
    # the index file serves the object API like any other
    fq = fx.Fastq(str(p))
    assert len(fq) == n and fq[n // 2].name == got["read"][n // 2][1].decode()
    for r in ranks:
        r.blob.close()


def test_fasta_table_through_the_parts_route(fx, tmp_path, monkeypatch):
    """kind 0 (`seq`, `chromidx`; index.c:178-207, 239-251, 363): one handle as the only part of a PartsWriter writes what the
    single-device route writes -- seven integer columns of every width, names one byte behind the header offsets."""
    from pyfastx_amd import _lib, fxi
    rng = np.random.default_rng(8)
    lens = [0, 1, 127, 128, 40_000] + rng.integers(0, 600, 30_000).tolist()
    raw = b"".join(b">rec%d some description %d\n" % (i, i) + b"\n".join(bytes(rng.choice(list(b"ACGTN"), min(60, L - a)).astype(np.uint8)) for a in range(0, L, 60)) + (b"\n" if L else b"")
                   for i, L in enumerate(lens))
    monkeypatch.setenv("FX_FXI_DEV_MIN", "0")
    p1 = tmp_path / "one.fa"
    p1.write_bytes(raw)
    fa = fx.Fasta(str(p1))
    assert fa.index_phases is not None
    db = _db(str(p1) + ".fxi")
    want = db.execute("SELECT * FROM seq ORDER BY ID").fetchall()
    want_names = [r[0] for r in db.execute("SELECT chrom FROM seq INDEXED BY chromidx ORDER BY chrom")]
    db.close()
    p2 = tmp_path / "parts.fa"
    p2.write_bytes(raw)
    b = _lib.Blob.from_file(str(p2))
    s = b.fasta_build()
    w = fxi.PartsWriter(str(p2) + ".fxi", 0, 0)
    assert w.add_local(b) == s.n_seq == len(lens)
    out = w.finish()
    out.execute("INSERT INTO stat (seqnum,seqlen) VALUES (?,?)", (int(s.n_seq), int(s.seq_len)))
    out.commit()
    out.close()
    b.close()
    db = _db(str(p2) + ".fxi")
    assert db.execute("PRAGMA integrity_check").fetchall() == [(b"ok",)]
    assert db.execute("SELECT * FROM seq ORDER BY ID").fetchall() == want
    assert [r[0] for r in db.execute("SELECT chrom FROM seq INDEXED BY chromidx ORDER BY chrom")] == want_names
    db.close()
    fb = fx.Fasta(str(p2))
    assert len(fb) == len(lens) and fb["rec4"].seq == fa["rec4"].seq and fb[len(lens) - 1].name == fa[len(lens) - 1].name


def test_duplicate_names_leave_no_index(fx, tmp_path, monkeypatch):
    from pyfastx_amd import shard
    raw = _fastq(20_000, 9, dup=True)
    want = _single(fx, tmp_path, raw, monkeypatch)
    assert want["index"] == []
    p = tmp_path / "dup.fq"
    p.write_bytes(raw)
    ranks = _logical_ranks(str(p), 3)
    shard.write_fastq_index_parts(ranks, str(p) + ".fxi")
    got = _whole(str(p) + ".fxi")
    assert got["check"] == [(b"ok",)] and got["index"] == [] and got["read"] == want["read"] and got["stat"] == want["stat"]


def test_a_row_that_needs_an_overflow_page_is_refused(fx, tmp_path):
    """A name of 5 000 bytes does not fit a leaf: FX_ERANGE, nothing left on disk (the callers then use the host loaders)."""
    from pyfastx_amd import _lib, shard
    raw = b"@a\nACGT\n+\nIIII\n@" + b"n" * 5000 + b"\nACGT\n+\nIIII\n" + _fastq(3000, 2)
    p = tmp_path / "long.fq"
    p.write_bytes(raw)
    ranks = _logical_ranks(str(p), 2)
    with pytest.raises(_lib.FxError) as e:
        shard.write_fastq_index_parts(ranks, str(p) + ".fxi")
    assert e.value.code == _lib.FX_ERANGE and not os.path.exists(str(p) + ".fxi")


def test_windows_of_a_stream_larger_than_its_budget(fx, tmp_path, monkeypatch):
    raw = _fastq(400_000, 4)
    want = _single(fx, tmp_path, raw, monkeypatch)
    p = tmp_path / "win.fq"
    p.write_bytes(raw)
    monkeypatch.setenv("FX_HBM_BUDGET", "16M")
    fq = fx.Fastq(str(p))
    wq = fq._st.md
    assert wq is not None and wq.windows >= 4 and "index_kernels" in getattr(wq, "index_laps", {})      # in windows, pages from the device
    assert wq.cache.staged == wq.windows                          # every range staged ONCE: its leaves were written while it was resident for the build
    got = _whole(str(p) + ".fxi")
    assert got["check"] == [(b"ok",)] and got["index"] == ["readidx"]
    assert got["read"] == want["read"] and got["stat"] == want["stat"] and got["by_name"] == want["by_name"]
    assert fq[123_456].seq == fx.Fastq(str(tmp_path / "single.fq"))[123_456].seq


def test_ranges_of_a_file_that_is_still_being_staged(fx, tmp_path, monkeypatch):
    """Fastq(path) of a large plain file: staged in the background (fx_open_file_async), indexed range by range through views of
    the blob as the bytes land, the table leaves of a range on their way to the index file while the next ranges arrive.  Small
    ranges here (1 MiB; 4 GiB by default) and one read of 300 KB, longer than the 64 KiB halo of a range: its range is taken
    again with a larger one.  Rows, stat, names in order and reads equal the one-blob build's."""
    big = b"@long one\n" + b"ACGT" * 75_000 + b"\n+\n" + b"I" * 300_000 + b"\n"
    raw = _fastq(60_000, 21) + big + _fastq(60_000, 22).replace(b"@SRR8539271.", b"@SRR8539272.")
    monkeypatch.setenv("FX_FXI_DEV_MIN", "0")
    monkeypatch.setenv("FX_FXI_PRESIZE_MIN", "0")
    p1 = tmp_path / "one.fq"
    p1.write_bytes(raw)
    monkeypatch.setenv("FX_FQ_NO_PIPELINE", "1")
    one = fx.Fastq(str(p1))
    assert "pipelined_ranges" not in one.build_phases
    want = _whole(str(p1) + ".fxi")
    monkeypatch.delenv("FX_FQ_NO_PIPELINE")
    monkeypatch.setenv("FX_FQ_PIPELINE", "1")                   # (off by default: no gain on the bench host)
    monkeypatch.setenv("FX_FQ_PIPELINE_RANGE", str(1 << 20))
    p2 = tmp_path / "ranges.fq"
    p2.write_bytes(raw)
    fq = fx.Fastq(str(p2))
    assert fq.build_phases.get("pipelined_ranges", 0) >= 8 and fq.index_phases is not None
    got = _whole(str(p2) + ".fxi")
    assert got["check"] == [(b"ok",)] and got["index"] == ["readidx"]
    assert got["read"] == want["read"] and got["stat"] == want["stat"] and got["by_name"] == want["by_name"]
    assert len(fq) == len(one) == 120_001
    for i in (0, 59_999, 60_000, 60_001, 120_000):
        assert fq[i].seq == one[i].seq and fq[i].qual == one[i].qual and fq[i].name == one[i].name
    assert fq["long"].id == 60_001 and len(fq["long"].seq) == 300_000


# ---------------------------------------------------------------------------------- one process per rank, gloo, one GPU
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, path, out, q):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from pyfastx_amd import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sq = shard.ShardedFastq(path, rank, world, device=0)
        n = sq.write_index(out)
        q.put((rank, n, sq.n_local, getattr(sq, "index_laps", None)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 120_000), (3, 120_000), (4, 7)])
def test_one_process_per_rank_over_gloo(fx, tmp_path, monkeypatch, world, n):
    raw = _fastq(n, 10 + world)
    want = _single(fx, tmp_path, raw, monkeypatch)
    p = tmp_path / "ranks.fq"
    p.write_bytes(raw)
    out = str(p) + ".fxi"
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, str(p), out, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted(q.get() for _ in range(world))
    for pr in procs:
        pr.join(120)
        assert pr.exitcode == 0
    assert res[0][1] == n and all(r[1] is None for r in res[1:]) and sum(r[2] for r in res) == n
    got = _whole(out)
    assert got["check"] == [(b"ok",)] and got["index"] == ["readidx"]
    assert got["read"] == want["read"] and got["stat"] == want["stat"] and got["by_name"] == want["by_name"]


def test_large_idle_blocks_come_and_go(tmp_path):
    """The scratch pool's rule for the blobs of closed streams, on the device (csrc/fxgpu.hip: ScratchPool; the rule itself is pinned on
    the CPU: test_large_idle_blocks_policy).  With a 1 MiB pool limit every blob of these files is "large": four sizes opened and
    closed in turn under a cap of 64 MiB (the 48 MiB block pushes smaller ones out, a 100 MiB one is never kept), a time to live of
    50 ms, then everything given back -- every byte read back equals the file's, before and after."""
    import subprocess
    code = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
from pyfastx_amd import _lib
rng = np.random.default_rng(3)
files = []
for k, mb in enumerate((8, 16, 48, 100, 12)):
    p = os.path.join(%r, "f%%d.fa" %% k)
    body = rng.integers(65, 91, mb << 20, dtype=np.uint8)
    body[60::61] = 10
    raw = b">r%%d\n" %% k + body.tobytes() + b"\n"
    open(p, "wb").write(raw)
    files.append((p, raw))
for rnd in range(3):
    for p, raw in files:
        b = _lib.Blob.from_file(p)
        n = len(raw)
        assert b.size == n and b.read_bytes(0, 4096) == raw[:4096] and b.read_bytes(n - 5000, 5000) == raw[-5000:]
        assert b.fasta_build().n_seq == 1
        b.close()
    if rnd == 1:
        time.sleep(0.12)                                    # past the time to live: the next call into the pool gives the idle blocks back
_lib.lib().fx_release_scratch()
b = _lib.Blob.from_file(files[2][0]); assert b.read_bytes(100, 100) == files[2][1][100:200]; b.close()
print("ok")
''' % (os.path.dirname(HERE), str(tmp_path))
    env = dict(os.environ, FX_SCRATCH_CACHE_MB="1", FX_SCRATCH_KEEP_BIG_MB="64", FX_SCRATCH_BIG_TTL_S="0.05", FX_TRACE_ALLOC="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]
    assert "scratch full: hipFree" in out.stderr             # blocks did leave the pool (cap, time to live)
