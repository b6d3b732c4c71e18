#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REAL reference (oracle/_ref).

Run in the build container only (needs /root/reference to have produced
oracle/_ref/pyfastx*.so via `make -C oracle ref`):

    python tests/golden/make_golden.py

It (1) runs reference pyfastx on the fixture files in tests/data/ and on a set
of generated edge-case inputs, (2) dumps every index row (`seq`, `stat`,
`comp`, `read`, `base`, `meta`) and a seeded sample of fetch results, and
(3) asserts that the C restatement in oracle/fx_oracle.c reproduces all of it
-- that assertion is what pins the oracle.  The JSON travels to the GPU box;
the reference does not.
"""
import gzip
import json
import os
import random
import shutil
import sqlite3
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyfastx            # noqa: E402  (the reference, built from /root/reference/src)
import fxoracle           # noqa: E402

DATA = os.path.join(ROOT, "tests", "data")

FASTA_EDGE = {
    "no_trailing_newline": ">a desc\nACGT\nAC",
    "crlf": ">a desc\r\nACGT\r\nACGT\r\nAC\r\n>b\r\nGG\r\n",
    "ragged": ">a\nACGT\nAC\nACGT\nA\nACGTAC\n>b\nAC\nACG\nA\n",
    "one_odd_middle_line": ">c\nAAAA\nCC\nGGGG\n",
    "leading_blank_lines": "\n\n>a\nACGTAC\nGT\n",
    "tab_in_header": ">chr1\tdesc here\nACGT\n>chr2 x\ty\nGGCC\nAA\n",
    "lowercase": ">s1\nacgtnNRYKM\nacgtnNRYKM\nac\n",
    "empty_records": ">a\n>b\nAC\n>c\n",
    "header_at_eof_no_newline": ">a\nAC\n>b",
    "blank_lines_inside": ">a\nAC\n\nGT\n\n>b\n\nAC\n",
    "single_long_line": ">one\n" + "ACGTTGCA" * 500 + "\n>two\n" + "GATTACA" * 100,
    "spaces_in_sequence": ">a\nAC GT\nAC GT\nA\n",
    "gt_inside_line": ">a\nAC>GT\nACCGT\n>b>c\nAA\n",
    "crlf_no_trailing_newline": ">a\r\nACGT\r\nAC",
    "crlf_header_lf_body": ">a x\r\nACGT\nACGT\nAC\n",
    "lf_header_crlf_body": ">a x\nACGT\r\nACGT\r\nAC\r\n",
    "bare_header": ">\nAC\n> lead space\nGG\n",
    "crlf_blank_lines": ">a\r\nAC\r\n\r\n\nGT\r\n",
    "iupac": ">iu\nACGTUMRWSYKVHDBN\nacgtumrwsykvhdbn\n*-.\n",
    "wide_then_narrow": ">w\n" + ("A" * 80 + "\n") * 5 + "A" * 33 + "\n>n\n" + ("C" * 7 + "\n") * 9,
}

FASTQ_EDGE = {
    "lf": "@r1 d1\nACGT\n+\nIIII\n@r2\nGGCCA\n+r2\n#!~AB\n",
    "crlf": "@r1 d1\r\nACGT\r\n+\r\nIIII\r\n@r2\r\nGGCCA\r\n+\r\n#!5AB\r\n",
    "no_trailing_newline": "@r1\nACGT\n+\nIIII\n@r2\nGG\n+\nAB",
    "incomplete_tail": "@r1\nACGT\n+\nIIII\n@r2\nGGTT\n+\n",
    "incomplete_tail2": "@r1\nACGT\n+\nIIII\n@r2\nGGTT\n",
    "names_with_spaces": "@a b c\nAC\n+\nII\n@x  y\nGT\n+\nJJ\n",
    "lowercase_and_n": "@r\nacgtNnACGT\n+\nhhhhhhhhhh\n",
    "phred64": "@r\nACGT\n+\nhgfe\n@s\nAAAA\n+\nefgh\n",
    "tab_in_name": "@r\t1 2\nAC\n+\nII\n",
    # a '\r' INSIDE a quality line: the reference's loop shrinks line.l as it goes (fastq.c:733-737), so the last bytes of the
    # line are never examined ('!' and '~' below stay out of minqs / maxqs) and meta.minlen / maxlen are the shrunken lengths
    "cr_in_quality_short": "@r\nACGT\n+\nI\rI!\n@s\nAC\n+\n\r~\n",
    "cr_in_quality_long": "@r1\nACGTACGTACGTACGTACGT\n+\nIIII\rIIIIIIIIIIIII#!\n@r2\nACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIII\n"
                          "@r3\nACGTACGTACGTACGTACGTAC\n+\nJJJJJJJJJJJJJJJJJJ\r\rJ~\n",
    "cr_in_quality_crlf": "@r1\r\nACGTACGTACGTACGTACGT\r\n+\r\nIIIIIIIIII\rIIIIIII5!\r\n@r2\r\nACGTA\rCGTACGTACGTACGT\r\n+\r\nHHHHHHHHHHHHHHHHHHHH\r\n",
}


def table(db, name):
    try:
        return [list(r) for r in db.execute("SELECT * FROM %s" % name)]
    except sqlite3.OperationalError:
        return None


def despaced_len(raw):
    return len(raw.replace(b"\n", b"").replace(b"\r", b"").replace(b" ", b""))


def ref_fasta(path, raw, full_name=False, uppercase=False, sample=0, seed=0):
    """Run the reference on `path`; return golden dict."""
    # NB: no index_file= kwarg -- the reference's copy of a user-supplied index path
    # writes one byte past its malloc (index.c:51-53); the default path+".fxi" is safe.
    fxi = path + ".fxi"
    if os.path.exists(fxi):
        os.unlink(fxi)
    fa = pyfastx.Fasta(path, full_index=True, full_name=full_name, uppercase=uppercase)
    g = {"size": fa.size, "count": len(fa)}
    db = sqlite3.connect(fxi)
    g["seq"] = table(db, "seq")
    g["stat"] = table(db, "stat")[0][:2]
    g["comp"] = [r[1:] for r in table(db, "comp")]
    db.close()
    # full sequences (only where slen equals the despaced byte count: otherwise the
    # reference copies past the despaced buffer -- undefined, not pinned)
    seqs = {}
    for row in g["seq"]:
        rid, name, boff, blen, slen = row[0], row[1], row[2], row[3], row[4]
        if slen >= 0 and despaced_len(raw[boff:boff + blen]) == slen:
            s = fa[rid - 1]
            if boff + blen <= len(raw):
                seqs[str(rid)] = {"seq": s.seq, "desc": s.description, "raw": s.raw}
            else:   # header is the last, unterminated line: boff = N+1 (index.c:231 quirk);
                    # raw/description fread past EOF into an uninitialised buffer -> not pinned
                seqs[str(rid)] = {"seq": s.seq}
    g["records"] = seqs
    fetches = []
    if sample:
        rng = random.Random(seed)
        # records whose first line is blank have llen-elen == 0: the reference divides by
        # zero on any slice/flank of them (sequence.c:500, fasta.c:300) -> excluded
        ok = [r for r in g["seq"] if str(r[0]) in seqs and r[4] > 0 and r[5] - r[6] > 0]
        for _ in range(sample if ok else 0):
            r = rng.choice(ok)
            slen = r[4]
            a = rng.randrange(0, slen)
            b = rng.randrange(a, min(slen, a + 400) + 1)
            if r[7]:
                # norm=1: byte range per sequence.c:498-510.  If the range holds fewer than
                # b-a bases (possible for the "one odd line is still normal" quirk) the
                # reference copies stale cache bytes -- undefined, so not recorded.
                bpl = r[5] - r[6]
                if bpl <= 0:
                    continue          # reference divides by zero (sequence.c:500)
                off = r[2] + a + r[6] * (a // bpl)
                bl = (b - a) + (b // bpl - a // bpl) * r[6]
                if despaced_len(raw[off:off + bl]) < b - a:
                    continue
            sub = fa[r[0] - 1][a:b]              # single-level slice only (nested is history dependent)
            fetches.append({"id": r[0], "start": a, "stop": b, "seq": sub.seq,
                            "antisense": sub.antisense, "complement": sub.complement,
                            "reverse": sub.reverse, "name": sub.name,
                            # raw of a full-length slice reads uninitialised desc_len in the
                            # reference (sequence.c:476-481 never copies it) -> not pinned
                            "raw": sub.raw if 0 < b - a < slen else None})
        # Fasta.fetch / flank (fasta.c:384-515, 322-382)
        g["fetch"] = []
        for _ in range(max(10, sample // 10) if ok else 0):
            r = rng.choice(ok)
            slen = r[4]
            iv = []
            budget = slen      # the reference mallocs strlen(seq)+1 for ALL intervals (fasta.c:482)
            for _k in range(rng.randrange(1, 4)):
                s = rng.randrange(1, slen + 1)
                e = rng.randrange(s, min(slen, s + 200, s + max(budget, 1) - 1) + 1)
                if e - s + 1 > budget:
                    break
                budget -= e - s + 1
                iv.append([s, e])
            if not iv:
                iv = [[1, 1]]
            strand = rng.choice("+-")
            arg = tuple(iv[0]) if len(iv) == 1 else [tuple(x) for x in iv]
            g["fetch"].append({"name": r[1], "intervals": iv, "strand": strand,
                               "seq": fa.fetch(r[1], arg, strand=strand)})
        g["flank"] = []
        for _ in range(max(10, sample // 10) if ok else 0):
            r = rng.choice(ok)
            slen = r[4]
            s = rng.randrange(1, slen + 1)
            e = rng.randrange(s, min(slen, s + 50) + 1)
            fl = rng.choice([0, 5, 50, 100])
            uc = rng.choice([0, 1])
            g["flank"].append({"name": r[1], "start": s, "end": e, "flank": fl, "use_cache": uc,
                               "out": list(fa.flank(r[1], s, e, flank_length=fl, use_cache=uc))})
    g["fetches"] = fetches
    try:
        g["gc_content"] = fa.gc_content
        g["composition"] = fa.composition
    except RuntimeError:
        pass
    del fa
    os.unlink(fxi)
    return g


def ref_fastq(path, raw, sample=0, seed=0):
    fxi = path + ".fxi"
    if os.path.exists(fxi):
        os.unlink(fxi)
    fq = pyfastx.Fastq(path, full_index=True)
    g = {"count": len(fq), "size": fq.size}
    db = sqlite3.connect(fxi)
    g["read"] = table(db, "read")
    g["stat"] = table(db, "stat")[0]
    g["base"] = table(db, "base")[0]
    g["meta"] = table(db, "meta")[0]
    db.close()
    g["phred"] = fq.phred
    reads = []
    if sample and len(fq):
        rng = random.Random(seed)
        for _ in range(sample):
            i = rng.randrange(len(fq))
            r = fq[i]
            reads.append({"i": i, "name": r.name, "seq": r.seq, "qual": r.qual, "quali": r.quali,
                          "antisense": r.antisense, "complement": r.complement,
                          "reverse": r.reverse, "raw": r.raw, "desc": r.description})
    g["reads"] = reads
    del fq
    os.unlink(fxi)
    return g


# ---- oracle-vs-reference assertions (this is what pins oracle/fx_oracle.c) ----

def check_fasta(raw, g, full_name=False, uppercase=False):
    recs, tot = fxoracle.fasta_index(raw, full_name=full_name)
    assert len(recs) == g["count"] == len(g["seq"]), (len(recs), g["count"])
    assert tot == g["stat"][1] and len(recs) == g["stat"][0]
    for r, row in zip(recs, g["seq"]):
        name = raw[r["name_off"]:r["name_off"] + r["name_len"]].decode("latin-1")
        got = [name, int(r["boff"]), int(r["blen"]), int(r["slen"]), int(r["llen"]),
               int(r["elen"]), int(r["norm"]), int(r["dlen"])]
        assert got == row[1:], (got, row)
    comp = fxoracle.fasta_comp(raw, len(recs))
    rows = []
    for i in range(len(recs)):
        for b in range(128):
            if comp[i, b] > 0:
                rows.append([i + 1, b, int(comp[i, b])])
    totals = comp.sum(axis=0)
    rows += [[0, b, int(totals[b])] for b in range(128)]
    assert rows == g["comp"], "comp mismatch"
    up = 1 if uppercase else 0
    for rid, rec in g["records"].items():
        r = recs[int(rid) - 1]
        s = fxoracle.fetch(raw, r["boff"], r["blen"], r["slen"], up).decode("latin-1")
        assert s == rec["seq"], (rid, s[:50], rec["seq"][:50])
    for f in g["fetches"]:
        r = recs[f["id"] - 1]
        if r["norm"]:
            off, bl = fxoracle.slice_range(int(r["boff"]), int(r["llen"]), int(r["elen"]),
                                           f["start"], f["stop"])
            sl = f["stop"] - f["start"]
            for key, fl in (("seq", 0), ("reverse", 2), ("complement", 4), ("antisense", 6)):
                got = fxoracle.fetch(raw, off, bl, sl, fl | up).decode("latin-1")
                assert got == f[key], (key, f, got)
            if f["raw"] is not None:
                assert raw[off:off + bl].decode("latin-1") == f["raw"]
        else:
            full = fxoracle.fetch(raw, r["boff"], r["blen"], r["slen"], up)
            assert full[f["start"]:f["stop"]].decode("latin-1") == f["seq"]


def check_fastq(raw, g):
    recs, size, ln = fxoracle.fastq_index(raw)
    assert len(recs) == g["count"] == len(g["read"]), (len(recs), g["count"])
    assert size == g["stat"][1], (size, g["stat"])
    for r, row in zip(recs, g["read"]):
        name = raw[r["name_off"]:r["name_off"] + r["name_len"]].decode("latin-1")
        got = [name, int(r["dlen"]), int(r["rlen"]), int(r["soff"]), int(r["qoff"])]
        assert got == row[1:], (got, row)
    c = fxoracle.fastq_composition(raw)
    assert [c["a"], c["c"], c["g"], c["t"], c["n"]] == g["base"], (c, g["base"])
    assert [c["maxlen"], c["minlen"], c["minqs"], c["maxqs"], c["phred"]] == g["meta"], (c, g["meta"])
    for rd in g["reads"]:
        r = recs[rd["i"]]
        assert raw[r["soff"]:r["soff"] + r["rlen"]].decode("latin-1") == rd["seq"]
        assert raw[r["qoff"]:r["qoff"] + r["rlen"]].decode("latin-1") == rd["qual"]
        assert fxoracle.quali(raw, r["qoff"], r["rlen"], g["phred"]).tolist() == rd["quali"]
        assert fxoracle.revcomp(rd["seq"].encode(), 3).decode() == rd["antisense"]


def main():
    out = {"fasta_fixture": {}, "fastq_fixture": {}, "fasta_edge": {}, "fastq_edge": {}, "misc": {}}
    tmp = tempfile.mkdtemp(prefix="fxgold")

    for fn, sample in (("test.fa", 300), ("test.fa.gz", 300)):
        path = os.path.join(tmp, fn)
        shutil.copy(os.path.join(DATA, fn), path)
        raw = gzip.open(path).read() if fn.endswith(".gz") else open(path, "rb").read()
        g = ref_fasta(path, raw, sample=sample, seed=20260612)
        check_fasta(raw, g)
        out["fasta_fixture"][fn] = g
        # uppercase / full_name variants: rows only + a few records
        g2 = ref_fasta(path, raw, full_name=True, sample=20, seed=7)
        check_fasta(raw, g2, full_name=True)
        out["fasta_fixture"][fn + ":full_name"] = {"seq": g2["seq"][:5], "count": g2["count"]}
        print(fn, "ok:", g["count"], "records", len(g["fetches"]), "fetches")

    for fn, sample in (("test.fq", 200), ("test.fq.gz", 200)):
        path = os.path.join(tmp, fn)
        shutil.copy(os.path.join(DATA, fn), path)
        raw = gzip.open(path).read() if fn.endswith(".gz") else open(path, "rb").read()
        g = ref_fastq(path, raw, sample=sample, seed=99)
        check_fastq(raw, g)
        out["fastq_fixture"][fn] = g
        print(fn, "ok:", g["count"], "reads")

    for name, text in FASTA_EDGE.items():
        raw = text.encode()
        p = os.path.join(tmp, name + ".fa")
        open(p, "wb").write(raw)
        for up in (False, True):
            g = ref_fasta(p, raw, uppercase=up, sample=20, seed=len(name))
            check_fasta(raw, g, uppercase=up)
            g["text"] = text
            out["fasta_edge"][name + (":upper" if up else "")] = g
        print("fasta edge", name, "ok", out["fasta_edge"][name]["seq"])

    for name, text in FASTQ_EDGE.items():
        raw = text.encode()
        p = os.path.join(tmp, name + ".fq")
        open(p, "wb").write(raw)
        g = ref_fastq(p, raw, sample=8, seed=len(name))
        check_fastq(raw, g)
        g["text"] = text
        out["fastq_edge"][name] = g
        print("fastq edge", name, "ok", g["read"], g["meta"])

    # module-level helpers (module.c:44-59; util.c:228-249)
    rc_in = ["ATGC", "ACGUacgu", "ACGTUMRWSYKVHDBNacgtumrwsykvhdbn*-", "", "N", "AC GT\n"]
    out["misc"]["reverse_complement"] = [[s, pyfastx.reverse_complement(s)] for s in rc_in]
    for s, want in out["misc"]["reverse_complement"]:
        assert fxoracle.revcomp(s.encode(), 3).decode() == want

    for k, v in out.items():
        with open(os.path.join(HERE, k + ".json"), "w") as f:
            json.dump(v, f, separators=(",", ":"), sort_keys=True)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
