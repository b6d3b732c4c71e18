"""-m gpu: Fastx -- kseq_read on the device (pyfastx_amd/csrc/fx_kseq.hpp: line table, k_kq_walk, k_kq_gather) against
the golden vectors dumped from the reference's Fastx (tests/golden/fastx.json), against the oracle (oracle/fx_oracle.c:
fxo_kseq) on seeded files in which everything kseq tolerates happens, and side by side with the compiled reference
(oracle/_ref) on large files that drive the walker's 64-line steps, its ring and the long-line gather."""
import glob
import gzip
import os
import random
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden
from kseq_cases import FIXED, gen

pytestmark = pytest.mark.gpu

OPTIONS = [("fasta", False, False), ("fasta", True, True), ("fastq", False, False), ("fastq", False, True)]


@pytest.fixture(scope="module")
def fx():
    import pyfastx_amd
    from pyfastx_amd import _lib
    assert _lib.lib().fx_device_count() >= 1
    return pyfastx_amd


@pytest.fixture(scope="module")
def ref():
    if not glob.glob(os.path.join(ROOT, "oracle", "_ref", "pyfastx*.so")):
        pytest.skip("oracle/_ref not built")
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    import pyfastx
    return pyfastx


def _put(tmp_path, name, data):
    p = str(tmp_path / name)
    with open(p, "wb") as f:
        f.write(data)
    return p


def test_fastx_golden(fx, tmp_path):
    """Every golden case, every builder / option the vectors hold."""
    cases = load_golden("fastx")
    for k, case in enumerate(cases):
        p = _put(tmp_path, "g.fx", case["text"].encode("latin-1"))
        for key, want in case["out"].items():
            fmt, up, com = key.split(":")
            got = [list(t) for t in fx.Fastx(p, format=fmt, uppercase=bool(int(up)), comment=bool(int(com)))]
            assert got == want, (k, case["text"], key)


@pytest.mark.parametrize("seed", range(4))
def test_fastx_equals_the_oracle(fx, oracle, tmp_path, seed):
    rng = random.Random(9300 + seed)
    for data in [gen(rng) for _ in range(120)]:
        p = _put(tmp_path, "f.fx", data)
        for fmt, up, com in OPTIONS:
            got = list(fx.Fastx(p, format=fmt, uppercase=up, comment=com))
            assert got == oracle.fastx_tuples(data, fmt, uppercase=up, comment=com), (data, fmt, up, com)


def test_kseq_records_through_the_c_abi(fx, oracle):
    """fx_kseq_scan / fx_kseq_records / fx_kseq_fetch on a resident buffer: record table, end code and strings against
    fxo_kseq -- including the inputs whose iteration the reference ends with -2 (truncated quality)."""
    from pyfastx_amd import _lib
    rng = random.Random(31)
    codes = set()
    for data in list(FIXED) + [gen(rng) for _ in range(150)]:
        recs, seq, qual, code = oracle.kseq(data)
        blob = _lib.Blob.from_bytes(data)
        n_rec, n_lines, seq_bytes, end = blob.kseq_scan()
        assert (n_rec, end) == (len(recs), code), data
        assert n_lines == data.count(b"\n") + (1 if data and not data.endswith(b"\n") else 0)
        codes.add(end)
        if n_rec:
            t = blob.kseq_records(0, n_rec)
            assert t["hdr_off"].tolist() == recs["name_off"].tolist() and t["seq_len"].tolist() == recs["seq_len"].tolist(), data
            assert t["seq_cum"].tolist() == recs["seq_off"].tolist() and seq_bytes == int(recs["seq_off"][-1] + recs["seq_len"][-1])
            assert ((t["flags"] & 1) != 0).tolist() == (recs["qual_len"] != -1).tolist()
            assert ((t["flags"] & 2) != 0).tolist() == (recs["qual_len"] == -2).tolist()
            s, q = blob.kseq_fetch(0, n_rec, seq_bytes)
            assert bytes(s) == bytes(seq[:seq_bytes]), data
            for r, tr in zip(recs, t):
                if r["qual_len"] >= 0:
                    o = int(tr["seq_cum"])
                    assert bytes(q[o:o + int(r["qual_len"])]) == bytes(qual[int(r["qual_off"]):int(r["qual_off"] + r["qual_len"])]), data
            # a window of records in the middle: offsets relative to its first record
            if n_rec >= 3:
                a, b = 1, n_rec - 1
                nb = int(t["seq_cum"][b - 1] + t["seq_len"][b - 1] - t["seq_cum"][a])
                s2, _ = blob.kseq_fetch(a, b - a, nb, want_qual=False)
                assert bytes(s2) == bytes(seq[int(t["seq_cum"][a]):int(t["seq_cum"][a]) + nb])
        blob.close()
    assert codes == {-1, -2}


def test_index_free_iteration_of_fasta_and_fastq(fx, oracle, ref, tmp_path):
    """Fasta(path, build_index=False) and Fastq(path, build_index=False) iterate with kseq_read as well (index.c:609-664,
    fastq.c:598-622): the same device path, names joined with the comment by one space under full_name -- against the oracle
    on the kseq inputs whose first character lets the constructor through, and side by side with the reference."""
    rng = random.Random(77)
    seen = 0
    for data in list(FIXED) + [gen(rng) for _ in range(150)]:
        if oracle.kseq_undefined(data):
            continue
        first = data.lstrip()[:1]
        for kind, lead, ext in (("fasta", b">", "fa"), ("fastq", b"@", "fq")):
            if first != lead:
                continue
            p = _put(tmp_path, "i." + ext, data)
            for full_name in (False, True):
                for up in ((False, True) if kind == "fasta" else (False,)):
                    if kind == "fasta":
                        got = list(fx.Fasta(p, build_index=False, full_name=full_name, uppercase=up))
                        theirs = list(ref.Fasta(p, build_index=False, full_name=full_name, uppercase=up))
                    else:
                        got = list(fx.Fastq(p, build_index=False, full_name=full_name))
                        theirs = list(ref.Fastq(p, build_index=False, full_name=full_name))
                    assert got == oracle.index_free_tuples(data, kind, full_name, up) == theirs, (data, kind, full_name, up)
                    seen += 1
            assert not os.path.exists(p + ".fxi")
    assert seen > 200


def test_regular_files_need_no_walk(fx):
    """A file of four-line records, or of plain FASTA lines, is taken by the parallel passes alone (fx_kseq_prefix_lines);
    the first line that is neither hands over to the walk."""
    from pyfastx_amd import _lib
    fq = b"".join(b"@r%d\nACGTN\n+\nIIIII\n" % i for i in range(5000))
    for data, want in ((fq, 20000), (fq[:-1], 20000), (fq + b"@x\nAC\nGT\n+\nIIII\n" + fq, 20000), (b"junk\n" + fq, 0),
                       (b">a\nAC\nGT\n\n>b\nTT\n", 6), (b">a\nAC\nGT\n+\nIIII\n>b\nTT\n", 3), (b">a\nAC\nGT", 2)):
        blob = _lib.Blob.from_bytes(data)
        blob.kseq_scan()
        assert blob.kseq_prefix_lines() == want, data[:40]
        blob.close()


@pytest.mark.skipif(os.environ.get("FX_KSEQ_WALK_ONLY") == "1", reason="already the walk-only run")
def test_everything_through_the_walk():
    """The same comparisons with the parallel prefix passes switched off (FX_KSEQ_WALK_ONLY=1): every line goes through
    k_kq_walk, whose 64-line steps the default path reaches only behind a file's first irregular line."""
    env = dict(os.environ, FX_KSEQ_WALK_ONLY="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.abspath(__file__), "-k",
                        "golden or equals_the_oracle or c_abi or large_fastq_side_by_side or large_fasta"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def _big_fastq(rng, n, crlf, odd_every):
    """n records, mostly the four-line form (the walker's 16-records-at-once step), every odd_every-th one spread over
    several lines or followed by junk (its one-line step)."""
    e = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(n):
        ln = int(rng.integers(1, 160))
        s = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), ln))
        q = bytes(rng.integers(33, 75, ln, dtype=np.uint8))
        nm = b"@r%d len=%d" % (i, ln) if i % 3 else b"@r%d" % i
        if odd_every and i % odd_every == odd_every - 1:
            k = ln // 2
            out.append(nm + e + s[:k] + e + e + s[k:] + e + b"+" + e + q[:k] + e + q[k:] + e + (b"between records" + e if i % 2 else b""))
        else:
            out.append(nm + e + s + e + b"+" + e + q + e)
    return b"".join(out)


@pytest.mark.parametrize("crlf,odd_every", [(False, 0), (True, 0), (False, 1000), (True, 37)])
def test_large_fastq_side_by_side(fx, ref, tmp_path, crlf, odd_every):
    """300 k records = 1.2 M lines: hundreds of ring turns of k_kq_walk, its wide FASTQ step interrupted where records take
    more than four lines; several gather batches."""
    rng = np.random.default_rng(17 + odd_every + crlf)
    data = _big_fastq(rng, 300000, crlf, odd_every)
    p = _put(tmp_path, "big.fq", data)
    ours = fx.Fastx(p, comment=True)
    mine = list(ours)
    theirs = list(ref.Fastx(p, comment=True))
    assert len(mine) == len(theirs) == 300000 and ours.end_code == -1
    assert mine == theirs
    assert list(fx.Fastx(p, format="fasta")) == [t[:2] for t in theirs]


def test_large_fasta_side_by_side(fx, ref, tmp_path):
    """A FASTA file of wrapped records (the walker's header / sequence run step, records closing inside and across
    windows), one record on a single 40 MB line (a window with a line of KQ_BIG or more takes the one-line steps;
    k_kq_gather_long copies it) and one 1 MB record of 70 kB lines (the long-line list of an ordinary batch), lower case
    turned to upper case on the way."""
    rng = np.random.default_rng(5)
    alpha = np.frombuffer(b"ACGTacgtNn", dtype=np.uint8)
    out = []
    for i in range(3000):
        ln = int(rng.integers(0, 4000))
        s = bytes(rng.choice(alpha, ln))
        w = int(rng.integers(1, 120))
        out.append(b">s%d some text\n" % i + b"".join(s[k:k + w] + b"\n" for k in range(0, ln, w)) + (b"\n" if i % 7 == 0 else b""))
    out.append(b">huge\n" + bytes(rng.choice(alpha, 40 << 20)) + b"\n")
    big = bytes(rng.choice(alpha, 1 << 20))
    out.append(b">wide\n" + b"".join(big[k:k + 70000] + b"\n" for k in range(0, len(big), 70000)))
    out.append(b">last\nACGT")
    data = b"".join(out)
    p = _put(tmp_path, "big.fa", data)
    for kw in (dict(), dict(uppercase=True, comment=True)):
        assert list(fx.Fastx(p, **kw)) == list(ref.Fastx(p, **kw)), kw


def test_fastx_reads_gzip_and_bgzf(fx, oracle, tmp_path):
    rng = random.Random(2)
    data = b"".join(gen(rng) for _ in range(30))
    want = oracle.fastx_tuples(data, "fastq", comment=True)
    p = _put(tmp_path, "z.fq.gz", gzip.compress(data))
    assert list(fx.Fastx(p, format="fastq", comment=True)) == want
    # BGZF: members of at most 60000 bytes with the 'BC' extra field
    import struct
    import zlib
    mem = []
    for k in range(0, max(len(data), 1), 60000):
        chunk = data[k:k + 60000]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        mem.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body +
                   struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    mem.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    p2 = _put(tmp_path, "b.fq.gz", b"".join(mem))
    assert list(fx.Fastx(p2, format="fastq", comment=True)) == want
