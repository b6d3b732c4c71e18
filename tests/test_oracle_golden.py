"""CPU: the C restatement (oracle/fx_oracle.c) against the golden vectors that
tests/golden/make_golden.py dumped from the REAL reference.  This is the
"oracle is pinned" check that also runs where /root/reference does not exist."""
import numpy as np
import pytest

from conftest import fixture_bytes, load_golden


def _fasta_rows(oracle, raw, full_name=False):
    recs, tot = oracle.fasta_index(raw, full_name=full_name)
    rows = []
    for i, r in enumerate(recs):
        name = raw[r["name_off"]:r["name_off"] + r["name_len"]].decode("latin-1")
        rows.append([i + 1, name] + [int(r[c]) for c in ("boff", "blen", "slen", "llen", "elen", "norm", "dlen")])
    return recs, tot, rows


def _comp_rows(oracle, raw, n):
    comp = oracle.fasta_comp(raw, n)
    rec, abc = np.nonzero(comp)
    rows = [[int(r) + 1, int(a), int(comp[r, a])] for r, a in zip(rec, abc)]
    tot = comp.sum(axis=0)
    return rows + [[0, b, int(tot[b])] for b in range(128)]


@pytest.mark.parametrize("fn", ["test.fa", "test.fa.gz"])
def test_fasta_fixture(oracle, fn):
    g = load_golden("fasta_fixture")[fn]
    raw = fixture_bytes(fn)
    recs, tot, rows = _fasta_rows(oracle, raw)
    assert rows == g["seq"] and [len(recs), tot] == g["stat"]
    assert _comp_rows(oracle, raw, len(recs)) == g["comp"]
    for rid, rec in g["records"].items():
        r = recs[int(rid) - 1]
        assert oracle.fetch(raw, r["boff"], r["blen"], r["slen"]).decode() == rec["seq"]
    for f in g["fetches"]:
        r = recs[f["id"] - 1]
        off, bl = oracle.slice_range(int(r["boff"]), int(r["llen"]), int(r["elen"]), f["start"], f["stop"])
        n = f["stop"] - f["start"]
        assert oracle.fetch(raw, off, bl, n, 0).decode() == f["seq"]
        assert oracle.fetch(raw, off, bl, n, 6).decode() == f["antisense"]
        assert oracle.fetch(raw, off, bl, n, 4).decode() == f["complement"]
        assert oracle.fetch(raw, off, bl, n, 2).decode() == f["reverse"]
    g2 = load_golden("fasta_fixture")[fn + ":full_name"]
    assert _fasta_rows(oracle, raw, full_name=True)[2][:5] == g2["seq"]


def test_fasta_edge(oracle):
    for name, case in load_golden("fasta_edge").items():
        raw = case["text"].encode()
        recs, tot, rows = _fasta_rows(oracle, raw)
        assert rows == case["seq"], name
        assert _comp_rows(oracle, raw, len(recs)) == case["comp"], name
        up = 1 if name.endswith(":upper") else 0
        for rid, rec in case["records"].items():
            r = recs[int(rid) - 1]
            assert oracle.fetch(raw, r["boff"], r["blen"], r["slen"], up).decode("latin-1") == rec["seq"], (name, rid)


@pytest.mark.parametrize("fn", ["test.fq", "test.fq.gz"])
def test_fastq_fixture(oracle, fn):
    g = load_golden("fastq_fixture")[fn]
    raw = fixture_bytes(fn)
    recs, size, ln = oracle.fastq_index(raw)
    rows = [[i + 1, raw[r["name_off"]:r["name_off"] + r["name_len"]].decode(), int(r["dlen"]), int(r["rlen"]),
             int(r["soff"]), int(r["qoff"])] for i, r in enumerate(recs)]
    assert rows == g["read"] and size == g["stat"][1] and len(recs) == g["stat"][0]
    c = oracle.fastq_composition(raw)
    assert [c["a"], c["c"], c["g"], c["t"], c["n"]] == g["base"]
    assert [c["maxlen"], c["minlen"], c["minqs"], c["maxqs"], c["phred"]] == g["meta"]
    for rd in g["reads"]:
        r = recs[rd["i"]]
        assert oracle.quali(raw, r["qoff"], r["rlen"], g["phred"]).tolist() == rd["quali"]


def test_fastq_edge(oracle):
    for name, case in load_golden("fastq_edge").items():
        raw = case["text"].encode()
        recs, size, ln = oracle.fastq_index(raw)
        rows = [[i + 1, raw[r["name_off"]:r["name_off"] + r["name_len"]].decode(), int(r["dlen"]), int(r["rlen"]),
                 int(r["soff"]), int(r["qoff"])] for i, r in enumerate(recs)]
        assert rows == case["read"], name
        c = oracle.fastq_composition(raw)
        assert [c["a"], c["c"], c["g"], c["t"], c["n"]] == case["base"], name
        assert [c["maxlen"], c["minlen"], c["minqs"], c["maxqs"], c["phred"]] == case["meta"], name


def test_fastx_golden(oracle):
    """fxo_kseq + fxoracle.fastx_tuples against what the reference's Fastx yielded (tests/golden/make_golden_fastx.py)."""
    cases = load_golden("fastx")
    assert len(cases) > 100
    for case in cases:
        raw = case["text"].encode("latin-1")
        for key, want in case["out"].items():
            fmt, up, com = key.split(":")
            got = [list(t) for t in oracle.fastx_tuples(raw, fmt, uppercase=bool(int(up)), comment=bool(int(com)))]
            assert got == want, (raw, key)


def test_revcomp(oracle):
    for s, want in load_golden("misc")["reverse_complement"]:
        assert oracle.revcomp(s.encode(), 3).decode() == want
