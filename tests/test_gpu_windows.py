"""-m gpu: streams larger than the HBM they may use (pyfastx_amd/windows.py).  FX_HBM_BUDGET=64M makes a
300 MB FASTA / FASTQ file "too large": the index is built window after window on the one device with the multi-GPU
machinery (fx_open_file_range, boundary summaries, stitch / running line count), ONE .fxi is written, and fetches are routed
to windows that are staged on demand.  Rows, composition and 10 000 fetched intervals / reads equal the CPU oracle's on the
whole file (the reference streams such a file through a 1 MiB buffer: index.c:229-372, fastq.c:8-182)."""
import os
import sqlite3

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fx():
    import pyfastx_amd
    from pyfastx_amd import _lib
    assert _lib.lib().fx_device_count() >= 1
    return pyfastx_amd


def _big_fasta(rng, target):
    """~target bytes: records of very different sizes (a few tens of MB down to one line), three line widths, one CRLF-free
    stream, lower case and N runs, an odd-line record, a header line of 3 KB -- record boundaries fall anywhere in a window."""
    parts, size, i = [], 0, 0
    letters = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)
    while size < target:
        n = 400_000 if i == 5 else int(rng.choice([37, 5_000, 400_000, 9_000_000, 31_000_000]))
        w = int(rng.choice([60, 70, 80]))
        hdr = b">rec%d %s\n" % (i, b"x" * 3000 if i == 3 else b"len=%d" % n)
        s = letters[rng.integers(0, 9, n)]
        full = n // w
        body = np.full((full, w + 1), 10, dtype=np.uint8)
        body[:, :w] = s[:full * w].reshape(full, w)
        tail = s[full * w:].tobytes()
        blk = hdr + body.tobytes() + (tail + b"\n" if tail else b"")
        if i == 5:                                              # ONE odd line in the middle: norm = 1, not line-regular
            blk = hdr + body[:3].tobytes() + b"ACG\n" + body[3:].tobytes() + (tail + b"\n" if tail else b"")
        parts.append(blk)
        size += len(blk)
        i += 1
    return b"".join(parts)


def test_fasta_larger_than_the_budget(fx, tmp_path, oracle, monkeypatch):
    from pyfastx_amd import shard
    rng = np.random.default_rng(41)
    raw = _big_fasta(rng, 300_000_000)
    p = str(tmp_path / "big.fa")
    open(p, "wb").write(raw)
    monkeypatch.setenv("FX_HBM_BUDGET", "64M")
    fa = fx.Fasta(p, full_index=True)
    md = fa._st.md
    assert md is not None and md.windows >= 16 and len(md.cache.lru) <= 4        # built in windows, a budget's worth resident
    recs, tot = oracle.fasta_index(raw)
    db = sqlite3.connect(p + ".fxi")
    rows = db.execute("SELECT chrom, boff, blen, slen, llen, elen, norm, dlen FROM seq ORDER BY ID").fetchall()
    assert len(rows) == len(recs) == len(fa) and fa.size == tot
    for r, row in zip(recs, rows):
        assert row[0].encode() == raw[r["name_off"]:r["name_off"] + r["name_len"]]
        assert row[1:] == tuple(int(r[k]) for k in ("boff", "blen", "slen", "llen", "elen", "norm", "dlen"))
    comp = oracle.fasta_comp(raw, len(recs))
    got = np.zeros_like(comp)
    for sid, abc, num in db.execute("SELECT seqid, abc, num FROM comp WHERE seqid > 0"):
        got[sid - 1, abc] = num
    assert (got == comp).all()
    db.close()
    # 10 000 intervals all over the file, both strands, through the routed fetch: windows are staged as they are needed
    n = len(recs)
    w = recs["slen"].astype(np.float64)
    ids = rng.choice(n, 10_000, p=w / w.sum())
    st = (rng.random(10_000) * recs["slen"][ids]).astype(np.int64)
    sp = np.minimum(st + rng.integers(1, 300, 10_000), recs["slen"][ids])
    strand = rng.integers(0, 2, 10_000).astype(np.uint8)
    before = md.cache.staged
    buf, offs = fa.fetch_many(ids, st, sp, strand=strand)
    assert md.cache.staged > before and len(md.cache.lru) <= 4
    table = {k: recs[k].astype(np.int64) for k in ("boff", "blen", "llen", "elen", "norm")}
    table["reg"] = md.table["reg"]
    off, ln, skip, take = shard.slice_ranges(table, ids, st, sp)
    for j in range(10_000):
        want = oracle.fetch(raw, int(off[j]), int(ln[j]), int(skip[j]) + int(take[j]), flags=0)[int(skip[j]):]
        if strand[j]:
            want = oracle.revcomp(want, 3)
        assert buf[offs[j]:offs[j + 1]].tobytes() == want, j
    # per-object getters and iteration read through the windows too
    for i in (0, 3, 5, n - 1):
        s = fa[i]
        whole = oracle.fetch(raw, int(recs["boff"][i]), int(recs["blen"][i]), int(recs["slen"][i]))
        if len(whole) < 2_000_000:
            assert s.seq.encode() == whole
        assert s[2:30].seq.encode() == whole[2:30] and s.name.encode() == raw[recs["name_off"][i]:recs["name_off"][i] + recs["name_len"][i]]
    assert [s.name for _, s in zip(range(5), fa)] == [rows[i][0] for i in range(5)]
    # the index that exists is loaded by the next object; its first touch builds the windows again (no second index file)
    fb = fx.Fasta(p)
    b2, o2 = fb.fetch_many(ids[:500], st[:500], sp[:500], strand=strand[:500])
    assert np.array_equal(o2, offs[:501]) and b2.tobytes() == buf[:int(offs[500])].tobytes()


def test_fastq_larger_than_the_budget(fx, tmp_path, oracle, monkeypatch):
    from pyfastx_amd import synth
    import torch
    n = 860_000                                                 # ~300 MB
    blob, cols = synth.fastq_generate(n, torch.device("cuda", 0))
    raw = blob[:cols["n_bytes"]].cpu().numpy()
    del blob
    torch.cuda.empty_cache()
    p = str(tmp_path / "big.fq")
    raw.tofile(p)
    monkeypatch.setenv("FX_HBM_BUDGET", "64M")
    fq = fx.Fastq(p, full_index=True)
    wq = fq._st.md
    assert wq is not None and wq.windows >= 16 and len(wq.cache.lru) <= 4
    recs, size, _ = oracle.fastq_index(raw)
    assert len(fq) == len(recs) == n and fq.size == size
    db = sqlite3.connect(p + ".fxi")
    assert db.execute("PRAGMA integrity_check").fetchone()[0] == "ok"
    rows = np.array(db.execute("SELECT dlen, rlen, soff, qoff FROM read ORDER BY ID").fetchall(), dtype=np.int64)
    for j, k in enumerate(("dlen", "rlen", "soff", "qoff")):
        assert (rows[:, j] == recs[k]).all(), k
    names = [r[0] for r in db.execute("SELECT name FROM read ORDER BY ID")]
    rb = raw.tobytes()
    for i in list(range(0, n, 9973)) + [n - 1]:
        assert names[i].encode() == rb[recs["name_off"][i]:recs["name_off"][i] + recs["name_len"][i]]
    oc = oracle.fastq_composition(raw)
    assert db.execute("SELECT a, c, g, t, n FROM base").fetchone() == tuple(oc[k] for k in ("a", "c", "g", "t", "n"))
    assert db.execute("SELECT maxlen, minlen, minqs, maxqs, phred FROM meta").fetchone() == tuple(oc[k] for k in ("maxlen", "minlen", "minqs", "maxqs", "phred"))
    db.close()
    rng = np.random.default_rng(8)
    ids = rng.integers(0, n, 10_000)
    out = fq.fetch_many(ids)
    o = out["offsets"]
    for j in range(0, 10_000, 7):
        i = int(ids[j])
        s, q, l = int(recs["soff"][i]), int(recs["qoff"][i]), int(recs["rlen"][i])
        assert out["seq"][o[j]:o[j + 1]].tobytes() == rb[s:s + l] and out["qual"][o[j]:o[j + 1]].tobytes() == rb[q:q + l]
        assert (out["quali"][o[j]:o[j + 1]] == oracle.quali(raw, q, l, fq.phred)).all()
    assert len(wq.cache.lru) <= 4
    r = fq[n // 2]
    s, l = int(recs["soff"][n // 2]), int(recs["rlen"][n // 2])
    assert r.seq.encode() == rb[s:s + l] and fq[r.name].id == r.id and r.quali == oracle.quali(raw, int(recs["qoff"][n // 2]), l, fq.phred).tolist()
    got = [(x.name, x.seq) for _, x in zip(range(40_000), fq)]          # the iterator's batches are gathered through the windows
    assert got[-1][1].encode() == rb[int(recs["soff"][39_999]):int(recs["soff"][39_999]) + 150] and got[0][0] == names[0]
    with pytest.raises(IndexError):
        fq.fetch_many([n])
