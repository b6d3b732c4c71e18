"""-m gpu: HIP path (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

from conftest import fixture_bytes, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from pyfastx_amd import _lib
    _lib.lib()
    assert _lib.lib().fx_device_count() >= 1, "no HIP device: the GPU tests must run on the MI355X box"
    return _lib


def fasta_rows(blob_cls, raw, full_name=False):
    b = blob_cls.from_bytes(raw)
    s = b.fasta_build(full_name)
    t = b.fasta_table(s.n_seq)
    return b, s, t


def assert_fasta_equal(oracle, L, raw, full_name=False):
    recs, tot = oracle.fasta_index(raw, full_name=full_name)
    b, s, t = fasta_rows(L.Blob, raw, full_name)
    assert s.n_seq == len(recs)
    assert s.seq_len == tot
    for col in ("hoff", "boff", "blen", "slen", "llen", "elen", "norm", "dlen", "name_len"):
        np.testing.assert_array_equal(t[col], recs[col].astype(t[col].dtype), err_msg=col)
    comp = b.fasta_comp(s.n_seq)
    np.testing.assert_array_equal(comp, oracle.fasta_comp(raw, len(recs)))
    # the same with the counters riding on the index scan (fx_fasta_build, FX_BUILD_COMP): one read of the stream
    s2 = b.fasta_build(full_name, comp=True)
    assert (s2.n_seq, s2.seq_len) == (s.n_seq, s.seq_len)
    t2 = b.fasta_table(s2.n_seq)
    for col in t:
        np.testing.assert_array_equal(t2[col], t[col], err_msg="fused build: " + col)
    np.testing.assert_array_equal(b.fasta_comp(s.n_seq), comp, err_msg="fused build: comp")
    from test_host_logic import _reg_of                     # the line-regular column: the same rule, on the device
    np.testing.assert_array_equal(b.fasta_line_regular(s.n_seq), np.array([_reg_of(raw, r) for r in recs], dtype=np.int32), err_msg="reg")
    return b, recs, t


@pytest.mark.parametrize("fn", ["test.fa", "test.fa.gz"])
def test_fasta_fixture_index(oracle, L, fn):
    raw = fixture_bytes(fn)
    b, recs, t = assert_fasta_equal(oracle, L, raw)
    g = load_golden("fasta_fixture")[fn]
    # against the committed reference rows as well
    for i, row in enumerate(g["seq"]):
        name = raw[t["hoff"][i] + 1: t["hoff"][i] + 1 + t["name_len"][i]].decode()
        got = [name] + [int(t[c][i]) for c in ("boff", "blen", "slen", "llen", "elen", "norm", "dlen")]
        assert got == row[1:]
    # fetches by (id, start, stop)
    f = g["fetches"]
    ids = np.array([x["id"] - 1 for x in f]); st = np.array([x["start"] for x in f]); sp = np.array([x["stop"] for x in f])
    for key, fl in (("seq", 0), ("reverse", L.FX_REVERSE), ("complement", L.FX_COMPLEMENT),
                    ("antisense", L.FX_REVERSE | L.FX_COMPLEMENT)):
        buf, offs, ol = b.fasta_fetch(ids, st, sp, flags=fl)
        np.testing.assert_array_equal(ol, sp - st)
        for j, x in enumerate(f):
            assert buf[offs[j]:offs[j + 1]].tobytes().decode() == x[key], (key, j)


def test_fasta_edge_cases(oracle, L):
    g = load_golden("fasta_edge")
    for name, case in g.items():
        if name.endswith(":upper"):
            continue
        raw = case["text"].encode()
        b, recs, t = assert_fasta_equal(oracle, L, raw)
        for i, row in enumerate(case["seq"]):
            nm = raw[t["hoff"][i] + 1: t["hoff"][i] + 1 + t["name_len"][i]].decode()
            got = [nm] + [int(t[c][i]) for c in ("boff", "blen", "slen", "llen", "elen", "norm", "dlen")]
            assert got == row[1:], (name, got, row)
        # whole records through fetch_ranges, plain and upper
        for up, key in ((0, name), (1, name + ":upper")):
            for rid, rec in g[key]["records"].items():
                k = int(rid) - 1
                buf, offs, ol = b.fetch_ranges([t["boff"][k]], [t["blen"][k]], [max(t["slen"][k], 0)], flags=up)
                assert buf[:ol[0]].tobytes().decode("latin-1") == rec["seq"], (key, rid)
            for fx in g[key]["fetches"]:
                buf, offs, ol = b.fasta_fetch([fx["id"] - 1], [fx["start"]], [fx["stop"]], flags=up)
                assert buf[:ol[0]].tobytes().decode("latin-1") == fx["seq"], (key, fx)
                buf, offs, ol = b.fasta_fetch([fx["id"] - 1], [fx["start"]], [fx["stop"]], flags=up | 6)
                assert buf[:ol[0]].tobytes().decode("latin-1") == fx["antisense"], (key, fx)


def _comp_both_ways(b):
    """base / meta of an indexed FASTQ stream: counted from the read table in a second pass (k_fastq_comp) AND counted while
    the index is built, in the one read of the stream (fx_fastq_build_comp: k_fastq_lines_comp + k_fastq_comp_reduce, the
    line-of-four guessed per run and verified against the prefix; fastq.c:715-774 is one loop over the reads) -- equal."""
    base, meta = b.fastq_comp()
    s0 = b.fastq_build(comp=True)
    base1, meta1 = b.fastq_comp()
    assert base1.tolist() == base.tolist() and meta1.tolist() == meta.tolist(), (base1, base, meta1, meta)
    s1 = b.fastq_build()                                     # ... and the plain build after it forgets the stored counts
    assert (s0.n_reads, s0.size, s0.n_lines) == (s1.n_reads, s1.size, s1.n_lines)
    base2, meta2 = b.fastq_comp()
    assert base2.tolist() == base.tolist() and meta2.tolist() == meta.tolist()
    return base, meta


@pytest.mark.parametrize("fn", ["test.fq", "test.fq.gz"])
def test_fastq_fixture(oracle, L, fn):
    raw = fixture_bytes(fn)
    recs, size, ln = oracle.fastq_index(raw)
    b = L.Blob.from_bytes(raw)
    s = b.fastq_build()
    assert (s.n_reads, s.size, s.n_lines) == (len(recs), size, ln)
    t = b.fastq_table(s.n_reads)
    for col in ("name_off", "name_len", "dlen", "rlen", "soff", "qoff"):
        np.testing.assert_array_equal(t[col], recs[col].astype(t[col].dtype), err_msg=col)
    base, meta = _comp_both_ways(b)
    c = oracle.fastq_composition(raw)
    assert base.tolist() == [c["a"], c["c"], c["g"], c["t"], c["n"]]
    assert meta.tolist() == [c["maxlen"], c["minlen"], c["minqs"], c["maxqs"], c["phred"]]
    g = load_golden("fastq_fixture")[fn]
    assert base.tolist() == g["base"] and meta.tolist() == g["meta"]
    ids = np.array([r["i"] for r in g["reads"]])
    seq, qual, qi, offs = b.fastq_fetch(ids, t["rlen"][ids], phred=g["phred"])
    for j, r in enumerate(g["reads"]):
        assert seq[offs[j]:offs[j + 1]].tobytes().decode() == r["seq"]
        assert qual[offs[j]:offs[j + 1]].tobytes().decode() == r["qual"]
        assert qi[offs[j]:offs[j + 1]].tolist() == r["quali"]


def test_fastq_edge_cases(oracle, L):
    for name, case in load_golden("fastq_edge").items():
        raw = case["text"].encode()
        b = L.Blob.from_bytes(raw)
        s = b.fastq_build()
        t = b.fastq_table(s.n_reads)
        assert s.n_reads == case["count"], name
        assert s.size == case["stat"][1], name
        for i, row in enumerate(case["read"]):
            nm = raw[t["name_off"][i]: t["name_off"][i] + t["name_len"][i]].decode()
            got = [nm, int(t["dlen"][i]), int(t["rlen"][i]), int(t["soff"][i]), int(t["qoff"][i])]
            assert got == row[1:], (name, got, row)
        base, meta = _comp_both_ways(b)
        assert base.tolist() == case["base"], name
        assert meta.tolist() == case["meta"], name


def test_revcomp(L):
    for s, want in load_golden("misc")["reverse_complement"]:
        assert L.revcomp_bytes(s.encode()).decode() == want


def _rand_fasta(rng, nrec, width, crlf=False, ragged=False, trailing=True, lower=False):
    eol = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(nrec):
        out.append(b">seq%d some desc\tx" % i + eol)
        n = int(rng.integers(0, 5000))
        alpha = np.frombuffer(b"ACGTNacgtn" if lower else b"ACGTN", dtype=np.uint8)
        s = alpha[rng.integers(0, alpha.size, n)].tobytes()
        p = 0
        while p < n:
            w = width if not ragged else int(rng.integers(1, width + 1))
            out.append(s[p:p + w] + eol)
            p += w
    raw = b"".join(out)
    if not trailing and raw.endswith(eol):
        raw = raw[:-len(eol)]
    return raw


@pytest.mark.parametrize("seed", range(6))
def test_fasta_random(oracle, L, seed):
    rng = np.random.default_rng(seed)
    raw = _rand_fasta(rng, int(rng.integers(1, 60)), int(rng.integers(5, 120)), crlf=bool(seed & 1),
                      ragged=(seed % 3 == 2), trailing=(seed != 4), lower=(seed > 2))
    b, recs, t = assert_fasta_equal(oracle, L, raw)
    ok = np.nonzero(recs["slen"] > 0)[0]
    nq = 500
    ids = rng.choice(ok, nq)
    st = (rng.random(nq) * recs["slen"][ids]).astype(np.int64)
    sp = np.minimum(st + rng.integers(0, 300, nq), recs["slen"][ids])
    fl = rng.integers(0, 8, nq).astype(np.uint8)
    buf, offs, ol = b.fasta_fetch(ids, st, sp, flags_per_query=fl)
    for j in range(nq):
        r = recs[ids[j]]
        from test_host_logic import _reg_of
        if _reg_of(raw, r):                                   # line-regular: the arithmetic; else (the odd-line record included) the true slice
            off, bl = oracle.slice_range(int(r["boff"]), int(r["llen"]), int(r["elen"]), int(st[j]), int(sp[j]))
            want = oracle.fetch(raw, off, bl, int(sp[j] - st[j]), int(fl[j]))
        else:
            full = oracle.fetch(raw, int(r["boff"]), int(r["blen"]), 1 << 60, int(fl[j]) & 1)
            want = full[st[j]:sp[j]]
            if fl[j] & 4:
                want = oracle.revcomp(want, 2)
            if fl[j] & 2:
                want = want[::-1]
        assert buf[offs[j]:offs[j] + ol[j]].tobytes() == want, (seed, j, fl[j])


def _pad_to(raw_parts, target):
    """bytes of sequence lines (width 60) so that the next byte written lands exactly at offset `target`"""
    cur = sum(len(p) for p in raw_parts)
    need = target - cur
    assert need > 2
    body = (b"ACGT" * ((need // 4) + 2))[:need - 1]
    out, p = [], 0
    while p < len(body):
        out.append(body[p:p + 60]); p += 60
    s = b"\n".join(out)
    s = s[:need - 1] + b"\n"
    return s


@pytest.mark.parametrize("crlf", [False, True])
def test_fasta_granule_shapes(oracle, L, crlf):
    """Shapes that matter to the 4 KiB granule machinery: header lines that start exactly at, one byte before /
    after, or run across a granule boundary; lines much longer than a granule (unwrapped FASTA); a record with
    one odd middle line (the norm = 1 quirk); '>' inside sequence lines; blank lines; no trailing newline."""
    rng = np.random.default_rng(11 + crlf)
    alpha = np.frombuffer(b"ACGTNacgtnRYKM", dtype=np.uint8)
    parts = [b">first record\n"]
    for target in (4096, 8191, 12289, 16384 - 7, 20480 + 4090):      # header '>' lands at / around granule boundaries
        parts.append(_pad_to(parts, target))
        parts.append(b">hdr_at_%d with a fairly long description to cross things %s\n" % (target, b"x" * int(rng.integers(0, 40))))
    # unwrapped records: one line of 3 ... 40 KiB
    for n in (3000, 4096 - 20, 9000, 40000):
        parts.append(b">unwrapped_%d\n" % n + alpha[rng.integers(0, alpha.size, n)].tobytes() + b"\n")
    # regular 60-column record over many granules, then the same with ONE odd middle line (still norm = 1) and with two (norm = 0)
    for odd in (0, 1, 2):
        body = alpha[rng.integers(0, 5, 30000)].tobytes()
        lines = [body[p:p + 60] for p in range(0, len(body), 60)]
        for k in range(odd):
            lines[100 + 37 * k] = lines[100 + 37 * k][:41]
        parts.append(b">odd%d\n" % odd + b"\n".join(lines) + b"\n")
    parts.append(b">gt_inside\nACGT>ACGT\n>ACGT is a header\nAC\n\n\nGT\n>last no newline\nACGTAC")
    raw = b"".join(parts)
    if crlf:
        raw = raw.replace(b"\n", b"\r\n")
    b, recs, t = assert_fasta_equal(oracle, L, raw)
    assert len(recs) >= 15
    ok = np.nonzero(recs["slen"] > 0)[0]
    nq = 800
    ids = rng.choice(ok, nq)
    st = (rng.random(nq) * recs["slen"][ids]).astype(np.int64)
    sp = np.minimum(st + rng.integers(0, 700, nq), recs["slen"][ids])
    fl = rng.integers(0, 8, nq).astype(np.uint8)
    buf, offs, ol = b.fasta_fetch(ids, st, sp, flags_per_query=fl)
    for j in range(nq):
        r = recs[ids[j]]
        from test_host_logic import _reg_of
        if _reg_of(raw, r):                                   # line-regular: the arithmetic; else (the odd-line record included) the true slice
            off, bl = oracle.slice_range(int(r["boff"]), int(r["llen"]), int(r["elen"]), int(st[j]), int(sp[j]))
            want = oracle.fetch(raw, off, bl, int(sp[j] - st[j]), int(fl[j]))
        else:
            full = oracle.fetch(raw, int(r["boff"]), int(r["blen"]), 1 << 60, int(fl[j]) & 1)
            want = full[st[j]:sp[j]]
            if fl[j] & 4:
                want = oracle.revcomp(want, 2)
            if fl[j] & 2:
                want = want[::-1]
        assert buf[offs[j]:offs[j] + ol[j]].tobytes() == want, (crlf, j, int(ids[j]), int(st[j]), int(sp[j]), int(fl[j]))


def _rand_fastq(rng, nrec, maxlen, crlf=False, plus_name=False, trailing=True):
    eol = b"\r\n" if crlf else b"\n"
    out = []
    alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)
    for i in range(nrec):
        n = int(rng.integers(1, maxlen + 1))
        name = b"@read%d/%d" % (i, n) + (b" extra words here" if i % 3 else b"") + (b"_" * int(rng.integers(0, 30)))
        out.append(name + eol)
        out.append(alpha[rng.integers(0, alpha.size, n)].tobytes() + eol)
        out.append((b"+" + name[1:] if plus_name and i % 2 else b"+") + eol)
        out.append(rng.integers(33, 75, n).astype(np.uint8).tobytes() + eol)
    raw = b"".join(out)
    if not trailing:
        raw = raw[:-len(eol)]
    return raw


@pytest.mark.parametrize("seed", range(6))
def test_fastq_random(oracle, L, seed):
    """Short reads (many newlines per granule, several compaction rounds), long reads (lines longer than a
    granule), CRLF, '+name' lines, unterminated last line -- index, composition and read fetches vs the oracle."""
    rng = np.random.default_rng(500 + seed)
    maxlen = (8, 150, 150, 30000, 3000, 2)[seed]
    nrec = (4000, 1500, 1500, 40, 300, 6000)[seed]
    raw = _rand_fastq(rng, nrec, maxlen, crlf=bool(seed & 1), plus_name=(seed >= 2), trailing=(seed != 4))
    recs, size, ln = oracle.fastq_index(raw)
    b = L.Blob.from_bytes(raw)
    s = b.fastq_build()
    assert (s.n_reads, s.size, s.n_lines) == (len(recs), size, ln)
    t = b.fastq_table(s.n_reads)
    for col in ("name_off", "name_len", "dlen", "rlen", "soff", "qoff"):
        np.testing.assert_array_equal(t[col], recs[col].astype(t[col].dtype), err_msg=col)
    base, meta = _comp_both_ways(b)
    c = oracle.fastq_composition(raw)
    assert base.tolist() == [c["a"], c["c"], c["g"], c["t"], c["n"]]
    assert meta.tolist() == [c["maxlen"], c["minlen"], c["minqs"], c["maxqs"], c["phred"]]
    ids = rng.integers(0, s.n_reads, 300)
    for flags in (0, L.FX_REVERSE | L.FX_COMPLEMENT):
        seq, qual, qi, offs = b.fastq_fetch(ids, t["rlen"][ids], phred=33, seq_flags=flags)
        for j, k in enumerate(ids):
            so, qo, n = int(recs["soff"][k]), int(recs["qoff"][k]), int(recs["rlen"][k])
            want = raw[so:so + n]
            if flags:
                want = oracle.revcomp(want, 3)
            assert seq[offs[j]:offs[j + 1]].tobytes() == want, (seed, j)
            assert qual[offs[j]:offs[j + 1]].tobytes() == raw[qo:qo + n], (seed, j)
            assert qi[offs[j]:offs[j + 1]].tolist() == [q - 33 for q in raw[qo:qo + n]], (seed, j)


def test_names_lookup(oracle, L):
    """fx_names_build / fx_names_lookup: every name resolves to its record, absent names to -1, duplicate
    names to the FIRST record (what `SELECT ... WHERE chrom=? LIMIT 1` returns, index.c:527-566)."""
    rng = np.random.default_rng(3)
    raw = fixture_bytes("test.fa")
    b, s, t = fasta_rows(L.Blob, raw)
    names = [raw[t["hoff"][i] + 1: t["hoff"][i] + 1 + t["name_len"][i]].decode() for i in range(s.n_seq)]
    b.names_build(0)
    perm = rng.permutation(s.n_seq)
    got = b.names_lookup([names[i] for i in perm] + ["nope", "", names[0] + "x", names[5][:-1]])
    assert got[:s.n_seq].tolist() == perm.tolist()
    assert got[s.n_seq:].tolist() == [-1, -1, -1, -1]
    # duplicates, empty names, names of very different lengths, full_name
    raw = b">dup 1\nAC\n>x\nGT\n>dup 2\nTT\n>\nAA\n>" + b"L" * 300 + b" tail\nCC\n>dup\nGG\n"
    b, s, t = fasta_rows(L.Blob, raw)
    b.names_build(0)
    assert b.names_lookup(["dup", "x", "", "L" * 300, "L" * 299, "dup 1"]).tolist() == [0, 1, 3, 4, -1, -1]
    b, s, t = fasta_rows(L.Blob, raw, full_name=True)
    b.names_build(0)
    assert b.names_lookup(["dup 1", "dup 2", "dup", "L" * 300 + " tail"]).tolist() == [0, 2, 5, 4]
    # FASTQ read names
    raw = _rand_fastq(rng, 5000, 60, crlf=True, plus_name=True)
    recs, size, ln = oracle.fastq_index(raw)
    fq = L.Blob.from_bytes(raw)
    sq = fq.fastq_build()
    fq.names_build(1)
    rn = [raw[int(recs["name_off"][i]): int(recs["name_off"][i]) + int(recs["name_len"][i])].decode() for i in range(sq.n_reads)]
    first = {}
    for i, nme in enumerate(rn):
        first.setdefault(nme, i)
    pick = rng.integers(0, sq.n_reads, 2000)
    assert fq.names_lookup([rn[i] for i in pick]).tolist() == [first[rn[i]] for i in pick]


def _py_order(names):
    """BINARY collation of SQLite = Python's bytes ordering; equal names in id order (sorted() is stable)."""
    return sorted(range(len(names)), key=names.__getitem__)


def test_names_sort(oracle, L):
    """fx_names_sort: the order `CREATE UNIQUE INDEX ... (name)` sorts into (index.c:363, fastq.c:152), computed on
    the GPU over the names in the resident stream; n_dup = number of adjacent equal names."""
    rng = np.random.default_rng(11)
    raw = fixture_bytes("test.fa")
    b, s, t = fasta_rows(L.Blob, raw)
    names = [raw[t["hoff"][i] + 1: t["hoff"][i] + 1 + t["name_len"][i]] for i in range(s.n_seq)]
    order, ndup = b.names_sort(0, s.n_seq)
    assert order.tolist() == _py_order(names) and ndup == 0
    # duplicates, the empty name, prefixes of each other, a NUL byte, lengths across several 8-byte chunks
    heads = [b"dup 1", b"x", b"dup 2", b"", b"L" * 300 + b" tail", b"dup", b"abcdefgh", b"abcdefghi", b"abcdefg",
             b"abcdefgh\x01", b"ab\x00c", b"ab", b"\xff\xfe", b"L" * 300, b"L" * 299 + b"M", b"abcdefghabcdefgh", b"abcdefghabcdefg",
             b"Z" * 2100 + b"b", b"Z" * 2100 + b"a", b"Z" * 2100]       # more than 256 chunks of 8 bytes
    raw = b"".join(b">" + h + b"\nACGT\n" for h in heads)
    for full in (False, True):
        b, s, t = fasta_rows(L.Blob, raw, full_name=full)
        names = [raw[t["hoff"][i] + 1: t["hoff"][i] + 1 + t["name_len"][i]] for i in range(s.n_seq)]
        order, ndup = b.names_sort(0, s.n_seq)
        assert order.tolist() == _py_order(names), full
        srt = [names[i] for i in order]
        assert ndup == sum(srt[i] == srt[i + 1] for i in range(len(srt) - 1))
        assert (ndup == 0) == full                          # first tokens: "dup" three times; whole headers: distinct
    # names of more than 4 KiB (the sort then keeps the bits that differ in global words, not in a workgroup's LDS)
    heads = [b"Q" * 5000 + b"x", b"short", b"Q" * 5000 + b"w", b"Q" * 4999, b"Q" * 5000 + b"x"]
    raw = b"".join(b">" + h + b"\nACGT\n" for h in heads)
    b, s, t = fasta_rows(L.Blob, raw)
    order, ndup = b.names_sort(0, s.n_seq)
    assert order.tolist() == _py_order(heads) and ndup == 1
    # FASTQ read names: random (duplicates likely with short names) and sequencer-style (long common prefix)
    for n, maxlen in ((5000, 6), (20000, 60)):
        raw = _rand_fastq(rng, n, maxlen, crlf=bool(n & 1), plus_name=True)
        recs, size, ln = oracle.fastq_index(raw)
        fq = L.Blob.from_bytes(raw)
        sq = fq.fastq_build()
        rn = [raw[int(recs["name_off"][i]): int(recs["name_off"][i]) + int(recs["name_len"][i])] for i in range(sq.n_reads)]
        order, ndup = fq.names_sort(1, sq.n_reads)
        assert order.tolist() == _py_order(rn)
        assert ndup == len(rn) - len(set(rn))
    # names that differ in ONE bit per byte over 400 bytes (the sort packs the differing bits of all chunks: 400 runs of one bit,
    # more than a round's key takes), and names where every bit of every byte differs (whole chunks as keys), duplicates in both
    for alphabet, ln_, n in ((b"ac", 400, 3000), (bytes(range(33, 256)), 21, 20000), (b"01", 9, 4000)):
        a = np.frombuffer(alphabet, dtype=np.uint8)[rng.integers(0, len(alphabet), (n, ln_))]
        a[n // 2] = a[n // 3]
        rn = [a[i].tobytes() for i in range(n)]
        raw = b"".join(b"@" + x + b"\nAC\n+\nII\n" for x in rn)
        fq = L.Blob.from_bytes(raw)
        sq = fq.fastq_build()
        assert sq.n_reads == n
        order, ndup = fq.names_sort(1, n)
        assert order.tolist() == _py_order(rn), (alphabet[:4], ln_)
        assert ndup == len(rn) - len(set(rn)) and ndup >= 1
    n = 30000
    ids = rng.permutation(n)
    raw = b"".join(b"@SRR8539271.%d %d/1\nACGTN\n+\nIIIII\n" % (i + 1, i) for i in ids.tolist())
    fq = L.Blob.from_bytes(raw)
    sq = fq.fastq_build()
    order, ndup = fq.names_sort(1, n)
    assert ndup == 0 and order.tolist() == _py_order([b"SRR8539271.%d" % (i + 1) for i in ids.tolist()])


def test_fasta_comp_letters(oracle, L):
    """k_fasta_comp: records long enough for the straight (whole-granule) path and for every phase of a wave's
    run, with soft-masked runs, N runs, CRLF, IUPAC codes, protein letters, '*', NUL and bytes >= 128 -- every bin of
    every record against the oracle (fasta.c:901-950)."""
    rng = np.random.default_rng(77)

    def body(n, alphabet, width, eol=b"\n", lower_runs=True, rare=None):
        a = np.frombuffer(alphabet, dtype=np.uint8)[rng.integers(0, len(alphabet), n)].copy()
        if lower_runs:                                       # soft-masked stretches
            for _ in range(max(1, n // 5000)):
                s = int(rng.integers(0, n)); e = min(n, s + int(rng.integers(1, 3000)))
                a[s:e] |= 0x20
        for _ in range(max(1, n // 40000)):                  # runs of N
            s = int(rng.integers(0, n)); e = min(n, s + int(rng.integers(1, 6000)))
            a[s:e] = ord("N") | (a[s:e] & 0x20)
        if rare is not None:
            idx = rng.integers(0, n, max(1, n // rare[1]))
            a[idx] = np.frombuffer(rare[0], dtype=np.uint8)[rng.integers(0, len(rare[0]), idx.size)]
        a[a == 10] = 65
        a[a == 62] = 65                                      # no '>' (it would start a record at a line start)
        rows = [a[i:i + width].tobytes() for i in range(0, n, width)]
        return eol.join(rows) + eol

    parts = [b">dna plain\n" + body(700_000, b"ACGT", 60),
             b">dna crlf\r\n" + body(300_000, b"ACGT", 70, eol=b"\r\n"),
             b">iupac\n" + body(400_000, b"ACGT", 80, rare=(b"RYKMSWBDHVUryn-*.", 50)),
             b">protein\n" + body(250_000, b"ACDEFGHIKLMNPQRSTVWY*", 60, lower_runs=False),
             b">noise\n" + body(300_000, b"ACGT", 4095, rare=(bytes(range(256)), 300)),
             b">one long line\n" + body(200_000, b"ACGTN", 10**9),
             b">short\nACGTNacgtn\n", b">empty\n", b">tail without newline\nACGTacgtNNnn*"]
    raw = b"".join(parts)
    recs, tot = oracle.fasta_index(raw)
    b, s, t = fasta_rows(L.Blob, raw)
    assert s.n_seq == len(recs) == len(parts)
    got = b.fasta_comp(s.n_seq)
    want = oracle.fasta_comp(raw, len(recs))
    np.testing.assert_array_equal(got, want)
    for _ in range(2):                                       # counters riding on the scan (k_scan_comp + k_comp_attribute), asked for twice
        assert b.fasta_build(comp=True).n_seq == s.n_seq
        np.testing.assert_array_equal(b.fasta_comp(s.n_seq), want)
        np.testing.assert_array_equal(b.fasta_comp(s.n_seq), want)
    assert b.fasta_build().n_seq == s.n_seq                  # and a plain build drops them
    np.testing.assert_array_equal(b.fasta_comp(s.n_seq), want)
    assert int(want[0][ord("a")]) > 0 and int(want[3][ord("W")]) > 0 and int(want[4][0]) > 0    # the cases are really in there
    # the same stream cut into shards: every shard counts its own bytes, rows add up (records that cross a cut
    # are completed by the owner of the header, see shard.py) -- here just the totals per letter
    tot_letters = got.sum(axis=0)
    assert int(tot_letters[ord("\n")]) == 0 and int(tot_letters[13]) == int(want[:, 13].sum())


def test_fastq_comp_letters(oracle, L):
    """k_fastq_comp: lower case, IUPAC codes, '*', bytes >= 128 (all N for the reference, fastq.c:715-753), reads
    shorter / longer than one 16-byte piece and than the 256-byte window, CRLF, quality bytes below '!' and above
    127 -- base counts and min / max quality against the oracle."""
    rng = np.random.default_rng(91)
    for eol in (b"\n", b"\r\n"):
        out = []
        alpha = np.frombuffer(b"ACGTNacgtnRYKM*-." + bytes([200, 255, 1]), dtype=np.uint8)
        for i in range(3000):
            n = int((1, 15, 16, 17, 150, 255, 256, 257, 700)[i % 9] if i % 4 else rng.integers(1, 400))
            seq = alpha[rng.integers(0, 5 if i % 3 else alpha.size, n)].tobytes().replace(b"\n", b"A")
            lo, hi = ((33, 75), (40, 41), (64, 105), (33, 127))[i % 4]
            q = rng.integers(lo, hi, n).astype(np.uint8)
            if i % 97 == 0:
                q[rng.integers(0, n)] = 14                    # below '!'
            if i % 101 == 0:
                q[rng.integers(0, n)] = 200                   # a negative char for the reference
            out += [b"@r%d some text" % i + eol, seq + eol, b"+" + eol, q.tobytes() + eol]
        raw = b"".join(out)
        c = oracle.fastq_composition(raw)
        b = L.Blob.from_bytes(raw)
        s = b.fastq_build()
        assert s.n_reads == 3000
        base, meta = _comp_both_ways(b)
        assert base.tolist() == [c["a"], c["c"], c["g"], c["t"], c["n"]], eol
        assert meta.tolist() == [c["maxlen"], c["minlen"], c["minqs"], c["maxqs"], c["phred"]], eol
        base2, meta2 = b.fastq_comp()                        # a second call starts from clean accumulators
        assert base2.tolist() == base.tolist() and meta2.tolist() == meta.tolist()


def test_fasta_comp_sparse(oracle, L):
    """fx_fasta_comp_sparse: the non-zero bins as (seqid, letter, count) triples in record order + column totals equal
    the dense composition (fasta.c:904-950); records without bytes, many records (several scan chunks), the fixture."""
    rng = np.random.default_rng(12)
    raws = [fixture_bytes("test.fa"), b">only header\n", b">a\nACGT\n>empty\n>b\nNNNN\nacgtRY\n"]
    parts = []
    for i in range(5000):                                    # > 4 chunks of 1024 records
        parts.append(b">r%d\n" % i)
        n = int(rng.integers(0, 200))
        if n:
            parts.append(np.frombuffer(b"ACGTNacgtn*", dtype=np.uint8)[rng.integers(0, 11, n)].tobytes() + b"\n")
    raws.append(b"".join(parts))
    for raw in raws:
        b, s, t = fasta_rows(L.Blob, raw)
        dense = b.fasta_comp(s.n_seq)
        for guess in (0, s.n_seq * 12):
            seqid, abc, num, total = b.fasta_comp_sparse(guess=guess)
            rec, letter = np.nonzero(dense)
            assert seqid.tolist() == (rec + 1).tolist() and abc.tolist() == letter.tolist()
            assert num.tolist() == dense[rec, letter].tolist()
            assert total.tolist() == (dense.sum(axis=0).tolist() if s.n_seq else [0] * 128)


def test_table_getters_refuse_a_wrong_row_count(L):
    """fx_fasta_table / fx_fasta_comp / fx_fastq_table fill as many rows as the index has; the binding allocates by
    the caller's n, so a stale or guessed n is refused instead of overrunning the arrays."""
    b, s, t = fasta_rows(L.Blob, b">a\nACGT\n>b\nGG\n")
    assert s.n_seq == 2
    for bad in (0, 1, 3):
        with pytest.raises(ValueError):
            b.fasta_table(bad)
        with pytest.raises(ValueError):
            b.fasta_comp(bad)
    b.fasta_build_begin()                                    # only enqueued: the getter waits for the count itself
    assert b.fasta_table(2)["slen"].tolist() == [4, 2]
    fq = L.Blob.from_bytes(b"@r\nAC\n+\nII\n")
    assert fq.fastq_build().n_reads == 1
    with pytest.raises(ValueError):
        fq.fastq_table(2)


def test_names_pack(oracle, L):
    """fx_names_pack: the names back to back + their offsets, from the record table in HBM (FASTA first token / whole
    header, FASTQ read names; empty names; more than one scan chunk of records)."""
    rng = np.random.default_rng(21)
    raw = fixture_bytes("test.fa")
    for full in (False, True):
        b, s, t = fasta_rows(L.Blob, raw, full_name=full)
        want = [raw[t["hoff"][i] + 1: t["hoff"][i] + 1 + t["name_len"][i]] for i in range(s.n_seq)]
        for guess in (0, 10**6):
            packed, offs = b.names_pack(0, s.n_seq, guess=guess)
            assert offs[0] == 0 and offs[-1] == packed.size == sum(len(x) for x in want)
            assert [packed[offs[i]:offs[i + 1]].tobytes() for i in range(s.n_seq)] == want
    b, s, t = fasta_rows(L.Blob, b">\nAC\n>x y\nGG\n>\n>zz\n")
    packed, offs = b.names_pack(0, s.n_seq)
    assert packed.tobytes() == b"xzz" and offs.tolist() == [0, 0, 1, 1, 3]
    raw = _rand_fastq(rng, 3000, 40, crlf=True, plus_name=True)
    recs, size, ln = oracle.fastq_index(raw)
    fq = L.Blob.from_bytes(raw)
    sq = fq.fastq_build()
    packed, offs = fq.names_pack(1, sq.n_reads)
    want = [raw[int(recs["name_off"][i]): int(recs["name_off"][i]) + int(recs["name_len"][i])] for i in range(sq.n_reads)]
    assert [packed[offs[i]:offs[i + 1]].tobytes() for i in range(sq.n_reads)] == want


def _mixed_fasta(rng, nbig=24, ntiny=30000):
    """Every FASTA shape in one stream (see test_fasta_mixed_shapes_60mb)."""
    parts = []

    def lines(seq, width, eol):
        return eol.join(seq[p:p + width] for p in range(0, len(seq), width)) + eol if seq else b""

    def bases(n, alpha=b"ACGTacgtNn"):
        return np.frombuffer(alpha, dtype=np.uint8)[rng.integers(0, len(alpha), n)].tobytes()

    for i in range(nbig):                                      # big records
        parts.append(b">big%d desc %d\n" % (i, i) + lines(bases(int(rng.integers(1, 4) * 1_000_000)), (60, 70, 80, 61)[i % 4], b"\n"))
    for i in range(ntiny):                                   # tiny records, every granule has several headers
        parts.append(b">t%d\n" % i + lines(bases(int(rng.integers(0, 400))), 60, b"\n"))
    parts.append(b">long header " + b"x" * 9000 + b"\n" + lines(bases(5000), 50, b"\n"))
    for i in range(200):                                     # CRLF
        parts.append(b">crlf%d\tq\r\n" % i + lines(bases(int(rng.integers(1, 3000))), 72, b"\r\n"))
    for i in range(200):                                     # ragged: line lengths vary -> norm = 0
        s, out, p = bases(int(rng.integers(1, 3000))), [], 0
        while p < len(s):
            w = int(rng.integers(1, 90)); out.append(s[p:p + w]); p += w
        parts.append(b">rag%d\n" % i + b"\n".join(out) + b"\n")
    parts.append(b">protein\n" + lines(bases(300000, b"ACDEFGHIKLMNPQRSTVWY*"), 60, b"\n"))
    noise = bytes(rng.integers(0, 256, 200000, dtype=np.uint8)).replace(b"\n", b"A").replace(b">", b"G")
    parts.append(b">noise\n" + lines(noise, 100, b"\n"))
    parts.append(b">blank lines\nACGT\n\n\nAC\n\n>last, unterminated\nACGTNNNNacgt")
    order = rng.permutation(len(parts) - 1).tolist() + [len(parts) - 1]      # shuffled, the unterminated one last
    raw = b"".join(parts[i] for i in order)
    return raw


def test_fasta_mixed_shapes_60mb(oracle, L):
    """One 60 MB stream of every shape at once -- multi-MB records, thousands of tiny ones, a header line longer
    than two granules, CRLF records, ragged records (norm = 0), protein / IUPAC / noise bytes, blank lines, an
    unterminated last line -- through the C ABI against the oracle: every index row, every composition bin, the sparse
    composition, the names, their sort order, and 3000 random fetches with all flag combinations."""
    rng = np.random.default_rng(2024)
    raw = _mixed_fasta(rng)
    assert 40_000_000 < len(raw) < 120_000_000
    b, recs, t = assert_fasta_equal(oracle, L, raw)          # rows + dense composition
    n = len(recs)
    dense = oracle.fasta_comp(raw, n)
    seqid, abc, num, total = b.fasta_comp_sparse(guess=n * 12)
    rr, ll = np.nonzero(dense)
    assert (seqid == rr + 1).all() and (abc == ll).all() and (num == dense[rr, ll]).all() and (total == dense.sum(axis=0)).all()
    names = [raw[int(recs["hoff"][i]) + 1: int(recs["hoff"][i]) + 1 + int(recs["name_len"][i])] for i in range(n)]
    packed, offs = b.names_pack(0, n)
    assert packed.tobytes() == b"".join(names) and offs[-1] == packed.size
    order_gpu, ndup = b.names_sort(0, n)
    assert order_gpu.tolist() == sorted(range(n), key=names.__getitem__) and ndup == 0
    ok = np.nonzero(recs["slen"] > 0)[0]
    nq = 3000
    ids = rng.choice(ok, nq)
    st = (rng.random(nq) * recs["slen"][ids]).astype(np.int64)
    sp = np.minimum(st + rng.integers(0, 400, nq), recs["slen"][ids])
    fl = rng.integers(0, 8, nq).astype(np.uint8)
    buf, offs, ol = b.fasta_fetch(ids, st, sp, flags_per_query=fl)
    for j in range(nq):
        r = recs[ids[j]]
        from test_host_logic import _reg_of
        if _reg_of(raw, r):                                   # line-regular: the arithmetic; else (the odd-line record included) the true slice
            off, bl = oracle.slice_range(int(r["boff"]), int(r["llen"]), int(r["elen"]), int(st[j]), int(sp[j]))
            want = oracle.fetch(raw, off, bl, int(sp[j] - st[j]), int(fl[j]))
        else:
            full = oracle.fetch(raw, int(r["boff"]), int(r["blen"]), 1 << 60, int(fl[j]) & 1)
            want = full[st[j]:sp[j]]
            if fl[j] & 4:
                want = oracle.revcomp(want, 2)
            if fl[j] & 2:
                want = want[::-1]
        assert buf[offs[j]:offs[j] + ol[j]].tobytes() == want, (j, int(ids[j]), int(fl[j]))


def test_fastq_mixed_shapes_40mb(oracle, L):
    """One 40 MB FASTQ stream with reads from 1 base to 200 kb, '+name' lines, lower case and IUPAC bases, qualities
    over the whole printable range -- index rows, composition, names, their order, name lookups and read fetches
    (seq, reverse complement, qual, phred-adjusted qualities) against the oracle."""
    rng = np.random.default_rng(77)
    alpha = np.frombuffer(b"ACGTNacgtnRYKM", dtype=np.uint8)
    out = []
    lens = np.concatenate([rng.integers(1, 400, 120000), rng.integers(1000, 20000, 300), [200000, 131072, 4096, 4095, 16, 15, 17]])
    rng.shuffle(lens)
    for i, n in enumerate(lens.tolist()):
        name = b"@m%d/%d extra %d" % (i, n, i % 7) if i % 3 else b"@m%d/%d" % (i, n)
        seq = alpha[rng.integers(0, 5 if i % 5 else alpha.size, n)].tobytes()
        qual = rng.integers(33, 127, n).astype(np.uint8).tobytes()
        out += [name + b"\n", seq + b"\n", (b"+" + name[1:] if i % 4 == 0 else b"+") + b"\n", qual + b"\n"]
    raw = b"".join(out)
    assert 30_000_000 < len(raw) < 80_000_000
    recs, size, ln = oracle.fastq_index(raw)
    b = L.Blob.from_bytes(raw)
    s = b.fastq_build()
    assert (s.n_reads, s.size, s.n_lines) == (len(recs), size, ln) and s.n_reads == len(lens)
    t = b.fastq_table(s.n_reads)
    for col in ("name_off", "name_len", "dlen", "rlen", "soff", "qoff"):
        np.testing.assert_array_equal(t[col], recs[col].astype(t[col].dtype), err_msg=col)
    base, meta = _comp_both_ways(b)
    c = oracle.fastq_composition(raw)
    assert base.tolist() == [c["a"], c["c"], c["g"], c["t"], c["n"]]
    assert meta.tolist() == [c["maxlen"], c["minlen"], c["minqs"], c["maxqs"], c["phred"]]
    n = s.n_reads
    names = [raw[int(recs["name_off"][i]): int(recs["name_off"][i]) + int(recs["name_len"][i])] for i in range(n)]
    packed, offs = b.names_pack(1, n)
    assert packed.tobytes() == b"".join(names)
    order, ndup = b.names_sort(1, n)
    assert ndup == 0 and order.tolist() == sorted(range(n), key=names.__getitem__)
    b.names_build(1)
    pick = rng.integers(0, n, 3000)
    assert b.names_lookup([names[i].decode() for i in pick] + ["absent"]).tolist() == pick.tolist() + [-1]
    ids = np.concatenate([rng.integers(0, n, 600), np.argsort(lens)[-8:]])       # the longest reads as well
    for flags in (0, L.FX_REVERSE | L.FX_COMPLEMENT):
        seq, qual, qi, offs = b.fastq_fetch(ids, t["rlen"][ids], phred=33, seq_flags=flags)
        for j, k in enumerate(ids):
            so, qo, m = int(recs["soff"][k]), int(recs["qoff"][k]), int(recs["rlen"][k])
            want = raw[so:so + m]
            if flags:
                want = oracle.revcomp(want, 3)
            assert seq[offs[j]:offs[j + 1]].tobytes() == want, (j, int(k))
            assert qual[offs[j]:offs[j + 1]].tobytes() == raw[qo:qo + m], (j, int(k))
            assert (qi[offs[j]:offs[j + 1]].astype(np.int16) + 33 == np.frombuffer(raw[qo:qo + m], dtype=np.uint8)).all(), (j, int(k))


def test_fetch_one_equals_batch(oracle, L):
    """fx_fetch_one (descriptor and result through pinned host memory, one launch) returns what fx_fetch_ranges
    returns for the same range: all flag combinations, skip, short and long ranges, and a result above its 1 MiB
    buffer (which takes the batch path)."""
    rng = np.random.default_rng(5)
    seq = np.frombuffer(b"ACGTNacgtnRY", dtype=np.uint8)[rng.integers(0, 12, 3_000_000)].tobytes()
    raw = b">r\n" + b"\n".join(seq[i:i + 60] for i in range(0, len(seq), 60)) + b"\n"
    b = L.Blob.from_bytes(raw)
    for _ in range(200):
        off = int(rng.integers(3, len(raw) - 10))
        blen = int(rng.integers(1, min(5000, len(raw) - off)))
        take = int(rng.integers(1, blen + 1))
        fl = int(rng.integers(0, 8))
        want, _, ol = b.fetch_ranges([off], [blen], [take], flags=fl)
        assert b.fetch_one(off, blen, take, flags=fl) == want[:int(ol[0])].tobytes()
    big = 2_500_000                                          # > 1 MiB of result
    want, _, ol = b.fetch_ranges([3], [big], [big], flags=1)
    assert b.fetch_one(3, big, big, flags=1) == want[:int(ol[0])].tobytes() and int(ol[0]) > (1 << 20)
    assert b.fetch_one(3, 100, 50, flags=0, skip=20) == oracle.fetch(raw, 3, 100, 1 << 30, 0)[20:70]


@pytest.mark.parametrize("crlf", [False, True])
def test_fastq_line_records(oracle, L, crlf):
    """The one-read FASTQ build (k_fastq_lines -> k_fastq_rows): a stream whose sampled windows hold long reads, so the
    count pass writes line records, with everything in it that the records have to get right -- a run of 4-byte reads
    that overflows the slots of its granules (those go back through k_fastq_emit), reads longer than a granule (granules
    without a newline), names that cross a granule boundary, header lines that do not start with '@' or start with a
    space, spaces in sequence / quality lines, empty names, CRLF -- against the oracle, row by row."""
    rng = np.random.default_rng(31 + crlf)
    eol = b"\r\n" if crlf else b"\n"
    alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)

    def rec(name, n, plus=b"+", seq=None, qual=None):
        s = alpha[rng.integers(0, 5, n)].tobytes() if seq is None else seq
        q = rng.integers(33, 127, n).astype(np.uint8).tobytes() if qual is None else qual
        return name + eol + s + eol + plus + eol + q + eol

    parts = []
    for i in range(700):                                      # long reads: what the three sampled windows see
        parts.append(rec(b"@long%d len=%d x" % (i, i), int(rng.integers(1500, 6000))))
    dense = b"".join(rec(b"@s%d" % i, 4) for i in range(900))                  # ~4-byte lines: > 128 lines per granule
    odd = [rec(b"no_at_sign%d some words" % 1, 50), rec(b"  two leading spaces", 30), rec(b"@", 10), rec(b"@ x", 10),
           rec(b"@q", 40, seq=b"ACGT ACGT" * 4 + b"ACGT", qual=b"II II" * 8),   # spaces where no name is
           rec(b"@" + b"n" * 5000 + b" tail", 20), rec(b"@w" + b"x" * 9000, 9000, plus=b"+" + b"w" * 9000)]
    mid = len(parts) // 3
    parts[mid:mid] = [dense] + odd
    # names across granule boundaries: pad so that a header line starts a few bytes before a multiple of 4096
    raw = b"".join(parts)
    for k in range(12):
        pad = (-len(raw) - 5 - k) % 4096
        raw += rec(b"@pad", max(pad - 20, 1) // 2 + 1)
        raw += rec(b"@cross_%d_boundary name rest" % k, 33)
    raw += b"".join(rec(b"@tail%d t" % i, int(rng.integers(1500, 6000))) for i in range(300))
    assert len(raw) > 3 * 262144 + 4 * len(dense)             # the dense run is not in a sampled window
    recs, size, ln = oracle.fastq_index(raw)
    b = L.Blob.from_bytes(raw)
    s = b.fastq_build()
    assert (s.n_reads, s.size, s.n_lines) == (len(recs), size, ln)
    t = b.fastq_table(s.n_reads)
    for col in ("name_off", "name_len", "dlen", "rlen", "soff", "qoff"):
        np.testing.assert_array_equal(t[col], recs[col].astype(t[col].dtype), err_msg=col)
    base, meta = _comp_both_ways(b)
    c = oracle.fastq_composition(raw)
    assert base.tolist() == [c["a"], c["c"], c["g"], c["t"], c["n"]]
    assert meta.tolist() == [c["maxlen"], c["minlen"], c["minqs"], c["maxqs"], c["phred"]]
    # the same stream unterminated
    raw2 = raw[:-len(eol)]
    recs2, size2, ln2 = oracle.fastq_index(raw2)
    b2 = L.Blob.from_bytes(raw2)
    s2 = b2.fastq_build()
    assert (s2.n_reads, s2.size, s2.n_lines) == (len(recs2), size2, ln2)
    t2 = b2.fastq_table(s2.n_reads)
    for col in ("name_off", "name_len", "dlen", "rlen", "soff", "qoff"):
        np.testing.assert_array_equal(t2[col], recs2[col].astype(t2[col].dtype), err_msg=col)


def _odd_line_fasta(rng, crlf):
    """Records with ONE odd line in the middle (norm = 1, not line-regular) among ordinary and ragged ones."""
    eol = b"\r\n" if crlf else b"\n"
    parts = []
    for i in range(40):
        parts.append(b">o%d some text" % i + eol)
        w = int(rng.integers(17, 90))
        m = int(rng.integers(3, 30))
        kind = i % 4                                         # 0: regular, 1: odd middle line, 2: long last line, 3: ragged
        for j in range(m):
            k = w
            if kind == 1 and j == m // 2:
                k = w - int(rng.integers(1, min(w - 1, 9)))
            if kind == 2 and j == m - 1:
                k = w + 3
            if kind == 3:
                k = int(rng.integers(1, w + 1))
            if kind == 0 and j == m - 1:
                k = int(rng.integers(1, w + 1))
            parts.append(bytes(rng.choice(list(b"ACGTNacgtn"), k).astype(np.uint8)) + eol)
    return b"".join(parts)


@pytest.mark.parametrize("crlf", [False, True])
def test_slices_of_records_with_one_odd_line(oracle, L, crlf):
    """A record with one odd line carries norm = 1 (index.c:342) and is NOT line-regular: every batched path -- by id
    (fx_fasta_fetch), by byte range with a slice after despacing (fx_fetch_slices) -- returns the true slice of the
    despaced record (sequence.c:100-110), like Sequence slices do; the device column agrees with the host rule."""
    from test_host_logic import _reg_of
    rng = np.random.default_rng(77 + crlf)
    raw = _odd_line_fasta(rng, crlf)
    b, recs, t = assert_fasta_equal(oracle, L, raw)
    reg = b.fasta_line_regular(len(recs))
    assert ((reg == 0) & (recs["norm"] == 1)).sum() >= 15 and (reg == 1).sum() >= 8
    nq = 4000
    ids = rng.integers(0, len(recs), nq)
    st = (rng.random(nq) * recs["slen"][ids]).astype(np.int64)
    sp = np.minimum(st + rng.integers(0, 300, nq), recs["slen"][ids])
    fl = rng.integers(0, 8, nq).astype(np.uint8)
    buf, offs, ol = b.fasta_fetch(ids, st, sp, flags_per_query=fl)
    full = [oracle.fetch(raw, r["boff"], r["blen"], r["slen"]) for r in recs]
    full5 = {f: [oracle.fetch(raw, r["boff"], r["blen"], r["slen"], f) for r in recs] for f in (0, 1, 4, 5)}
    for j in range(nq):
        w = full5[int(fl[j]) & 5][ids[j]][st[j]:sp[j]]
        w = w[::-1] if fl[j] & 2 else w
        assert buf[offs[j]:offs[j] + ol[j]].tobytes() == w, (j, int(ids[j]), int(st[j]), int(sp[j]), int(fl[j]), int(reg[ids[j]]))
    # the same through explicit ranges + skip (what ShardFetcher sends for records that are not line-regular)
    buf, offs, ol = b.fetch_ranges(recs["boff"][ids], recs["blen"][ids], sp - st, flags_per_query=fl, skip=st)
    for j in range(nq):
        w = full5[int(fl[j]) & 5][ids[j]][st[j]:sp[j]]
        w = w[::-1] if fl[j] & 2 else w
        assert buf[offs[j]:offs[j] + ol[j]].tobytes() == w, ("slices", j)
    # an installed table (rows of an existing .fxi) gets the same column
    b2 = L.Blob.from_bytes(raw)
    b2.fasta_set_table(*[recs[c] for c in ("boff", "blen", "slen", "llen", "elen", "norm")])
    np.testing.assert_array_equal(b2.fasta_line_regular(len(recs)), reg)
    buf2, offs2, ol2 = b2.fasta_fetch(ids, st, sp, flags_per_query=fl)
    assert buf2.tobytes() == b.fasta_fetch(ids, st, sp, flags_per_query=fl)[0].tobytes()


@pytest.mark.parametrize("n", [1, 2, 7, 200, 4097, 300_000])
def test_len_stats(L, n):
    """fx_fasta_len_stats (one device sort of the lengths) against numpy with the reference's definitions
    (fasta.c:573-849): count, first longest / shortest, the two middle lengths, N(p) / L(p) for several p."""
    rng = np.random.default_rng(n)
    for shape in range(3):
        if shape == 0:
            slen = rng.integers(0, 50, n)                      # many ties, zeros
        elif shape == 1:
            slen = np.exp(rng.uniform(0, np.log(3e8), n)).astype(np.int64)
        else:
            slen = np.full(n, 1234)
        slen = slen.astype(np.int64)
        b = L.Blob.from_bytes(b">x\nACGT\n")
        z = np.zeros(n, dtype=np.int64)
        b.fasta_set_table(z, z, slen, z + 5, (z + 1).astype(np.int32), (z + 1).astype(np.int32))
        total = int(slen.sum())
        srt = np.sort(slen)
        for p in (0, 37, 50, 90, 100):
            half = p / 100.0 * total
            for cmin in (0, 25, int(srt[n // 2])):
                st = b.fasta_len_stats(cmin, half)
                assert (st.n_seq, st.sum_len) == (n, total)
                assert st.count_ge == int((slen >= cmin).sum())
                assert (st.longest_id, st.longest_len) == (int(np.argmax(slen)), int(slen.max()))
                assert (st.shortest_id, st.shortest_len) == (int(np.argmin(slen)), int(slen.min()))
                assert st.med_lo == int(srt[(n - 1) // 2]) and st.med_hi == int(srt[(n - 1) // 2 + (1 if n % 2 == 0 else 0)])
                acc, want = 0, (0, 0)
                desc = srt[::-1]
                cs = np.cumsum(desc)
                hit = np.nonzero(cs.astype(np.float64) >= half)[0]
                if hit.size:
                    want = (int(desc[hit[0]]), int(hit[0]) + 1)
                assert (st.nx_len, st.nx_count) == want, (n, shape, p)


def test_fastq_one_read_build_on_a_file_that_misleads_its_guess(oracle, L):
    """fx_fastq_build_comp guesses the line-of-four of every run of granules from its 1-byte lines (the '+' lines of a usual
    file) and k_fastq_comp_reduce checks the guess against the newline prefixes.  Here the only 1-byte lines are SEQUENCE
    lines (the '+' lines repeat the name, the quality lines hold two bytes -- the reference checks none of that,
    fastq.c:89-171): every guess is wrong by one.  The composition must then come from the read table; the rows never
    depended on the guess.  Rows, base and meta equal the oracle's."""
    out = []
    for i in range(30_000):
        out += [b"@read%d some text here %d\n" % (i, i * 7), b"ACGTN"[i % 5:i % 5 + 1] + b"\n", b"+read%d\n" % i, b"I%c\n" % (40 + i % 50)]
    raw = b"".join(out)
    recs, size, ln = oracle.fastq_index(raw)
    c = oracle.fastq_composition(raw)
    b = L.Blob.from_bytes(raw)
    for comp in (True, False):
        s = b.fastq_build(comp=comp)
        assert (s.n_reads, s.size, s.n_lines) == (len(recs), size, ln) == (30_000, 30_000, 120_000)
        t = b.fastq_table(s.n_reads)
        for col in ("name_off", "name_len", "dlen", "rlen", "soff", "qoff"):
            np.testing.assert_array_equal(t[col], recs[col].astype(t[col].dtype), err_msg="%s comp=%s" % (col, comp))
        base, meta = b.fastq_comp()
        assert base.tolist() == [c["a"], c["c"], c["g"], c["t"], c["n"]]
        assert meta.tolist() == [c["maxlen"], c["minlen"], c["minqs"], c["maxqs"], c["phred"]]


def _fq_file(n, eol, rng, rlen=150, plus_name=False, long_every=0):
    alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)
    out = []
    for i in range(n):
        m = rlen if not long_every or i % long_every else 9000           # a read longer than two granules: runs without a '+' line in their first granule
        s = alpha[rng.integers(0, 5, m)].tobytes()
        q = rng.integers(35, 75, m).astype(np.uint8).tobytes()
        out.append(b"@r%d len=%d%s%s%s+%s%s%s%s" % (i, m, eol, s, eol, (b"r%d" % i) if plus_name else b"", eol, q, eol))
    return b"".join(out)


@pytest.mark.parametrize("shape", ["lf", "crlf", "crlf_fixture", "long_reads", "plus_name", "lone_cr", "cr_ends_granule"])
def test_fastq_one_read_composition_is_used_where_it_can_be(oracle, L, shape):
    """fx_fastq_build_comp on files of every shape (fastq.c:715-774 handles them all in its one loop): LF and CRLF files take the
    counts of the scan itself for (nearly) every run of granules -- fx_fastq_comp_info says how many runs were counted again
    from the prefixes --, a file whose '+' lines repeat the name is guessed from the first bytes of its lines, and a '\\r' that is not the end of its line
    sends everything through the table kernels; base / meta equal the oracle's in every case."""
    rng = np.random.default_rng(77)
    if shape == "crlf_fixture":
        raw = fixture_bytes("test.fq.gz")                    # the reference's own gzip fixture is a CRLF file
        assert raw.count(b"\r\n") == raw.count(b"\n") > 0
        raw = raw * 6                                        # (several runs of 64 KiB)
    elif shape == "lone_cr":
        raw = bytearray(_fq_file(3000, b"\n", rng))
        k = raw.index(b"\n+\n", len(raw) // 2) + 40          # a '\r' in the middle of a quality line
        raw[k] = 13
        raw = bytes(raw)
    elif shape == "cr_ends_granule":                         # the '\r' of a quality line as the LAST byte of a granule / of a run of 16, its '\n' behind it
        alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)
        parts, pos, i = [], 0, 0
        while pos < 24 * 65536:
            i += 1
            name = b"@r%d" % i
            rl = 150 + i % 5
            if i % 6 == 0:                                   # this record ends one byte behind a multiple of 4 KiB (every third time: of 64 KiB)
                unit = 65536 if i % 18 == 0 else 4096
                need = (1 - pos - len(name) - 9) % unit      # a record is len(name) + 2 rl + 9 bytes: name CRLF seq CRLF + CRLF qual CRLF
                if need % 2:
                    name += b"x"
                    need = (need - 1) % unit
                rl = need // 2
                if rl < 20:
                    rl += unit // 2
            rec = name + b"\r\n" + alpha[rng.integers(0, 5, rl)].tobytes() + b"\r\n+\r\n" + rng.integers(35, 75, rl).astype(np.uint8).tobytes() + b"\r\n"
            parts.append(rec)
            pos += len(rec)
            if i % 6 == 0:
                assert pos % 4096 == 1, (pos, len(rec))
        raw = b"".join(parts)
        ends = [k for k in range(4095, len(raw) - 1, 4096) if raw[k] == 13 and raw[k + 1] == 10]
        assert len(ends) > 20 and any(k % 65536 == 65535 for k in ends)
    else:
        # (plus_name: more than 64 runs of 64 KiB, all of them without a guess -- a handful would simply be counted again)
        raw = _fq_file(16000 if shape == "plus_name" else 4000, b"\r\n" if shape == "crlf" else b"\n", rng, plus_name=(shape == "plus_name"),
                       long_every=(25 if shape == "long_reads" else 0))
    recs, size, ln = oracle.fastq_index(raw)
    c = oracle.fastq_composition(raw)
    b = L.Blob.from_bytes(raw)
    s = b.fastq_build(comp=True)
    assert (s.n_reads, s.size, s.n_lines) == (len(recs), size, ln)
    runs, recounted, one_read = b.fastq_comp_info()
    base, meta = b.fastq_comp()
    assert base.tolist() == [c["a"], c["c"], c["g"], c["t"], c["n"]], shape
    assert meta.tolist() == [c["maxlen"], c["minlen"], c["minqs"], c["maxqs"], c["phred"]], shape
    assert runs == (len(raw) // 4096 + 15) // 16
    if shape in ("lf", "crlf", "crlf_fixture", "cr_ends_granule", "plus_name"):
        # (plus_name: no line of one byte anywhere -- the second guess: a line that begins with '@' whose next line but one begins with '+')
        assert one_read and 0 <= recounted <= max(1, runs // 100), (runs, recounted)
    elif shape == "long_reads":
        assert one_read and 0 < recounted <= max(64, runs // 8), (runs, recounted)
    else:
        assert not one_read                                  # the quirk of fastq.c:733-737 is k_fastq_qual_walk's
    t = b.fastq_table(s.n_reads)
    for col in ("name_off", "name_len", "dlen", "rlen", "soff", "qoff"):
        np.testing.assert_array_equal(t[col], recs[col].astype(t[col].dtype), err_msg=col)
