#!/usr/bin/env python3
"""SURVEY 8f-3 in numbers: `for r in fq: r.seq, r.qual` over an indexed FASTQ file and `for s in fa: s.seq` over a
many-record FASTA, against the same objects taken one by one (`fq[i].seq`).  usage: python tools/iter_rate.py [n_reads]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyfastx_amd as fx  # noqa: E402
from pyfastx_amd import synth  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
    d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    dev = torch.device("cuda", 0)
    blob, cols = synth.fastq_generate(n, dev)
    pq = os.path.join(d, "iter_rate.fq")
    blob[:cols["n_bytes"]].cpu().numpy().tofile(pq)
    del blob
    for p in (pq + ".fxi",):
        if os.path.exists(p):
            os.remove(p)
    t0 = time.perf_counter()
    fq = fx.Fastq(pq)
    t1 = time.perf_counter()
    tot = 0
    for r in fq:
        tot += len(r.seq) + len(r.qual)
    t2 = time.perf_counter()
    assert tot == 2 * 150 * n
    # where the iteration's time goes: the three stages of a batch, each over the whole file on its own
    from pyfastx_amd import _fxobj
    bd = {}
    try:
        cur = _fxobj.RowCursor(fq._index_file, "SELECT ID, name, dlen, rlen, soff, qoff FROM read ORDER BY ID")
        b0 = time.perf_counter()
        batches = []
        while True:
            b = cur.fetch(16384)
            if b is None:
                break
            batches.append(b)
        b1 = time.perf_counter()
        fetched = []
        for kk, names, raw in batches:
            cols = np.frombuffer(raw, dtype=np.int64).reshape(5, kk)
            fetched.append(fq._st.blob.read_fetch(cols[3], cols[4], cols[2], want=("seq", "qual")))
        b2 = time.perf_counter()
        objs = []
        for (kk, names, raw), (sq, ql, _, of) in zip(batches, fetched):
            objs.append(_fxobj.read_batch_cols(fx.Read, fq, names, raw, sq, ql, of))
        b3 = time.perf_counter()
        tot = 0
        for ob in objs:
            for r in ob:
                tot += len(r.seq) + len(r.qual)
        b4 = time.perf_counter()
        bd = {"rows_from_sqlite_us_per_read": round((b1 - b0) / n * 1e6, 3), "gather_seq_qual_us_per_read": round((b2 - b1) / n * 1e6, 3),
              "objects_us_per_read": round((b3 - b2) / n * 1e6, 3), "python_loop_two_getters_us_per_read": round((b4 - b3) / n * 1e6, 3)}
        del batches, fetched, objs
    except Exception as e:  # noqa: BLE001
        bd = {"breakdown": str(e)[:100]}
    k = 20000
    ids = np.random.default_rng(1).integers(0, n, k).tolist()
    fq[0].seq
    t3 = time.perf_counter()
    for i in ids:
        fq[i].seq
    t4 = time.perf_counter()
    for i in ids:
        r = fq[i]
        r.seq, r.qual, r.quali
    t4b = time.perf_counter()
    rnames = [fq[i].name for i in ids]
    t4c = time.perf_counter()
    for nm in rnames:
        fq[nm].seq
    t4d = time.perf_counter()
    # an object that LOADS the index file the first one wrote: statements until fq[i] has been used often enough, then its own
    # copy of the integer columns (read from the file in one pass: that pass is inside the time)
    fq2 = fx.Fastq(pq)
    fq2[0].seq
    k2 = 200_000
    ids2 = np.random.default_rng(2).integers(0, n, k2).tolist()
    t5 = time.perf_counter()
    for i in ids2:
        fq2[i].seq
    t5b = time.perf_counter()
    for i in ids2:
        fq2[i].seq
    t5c = time.perf_counter()
    same2 = all(fq2[i].name == fq[i].name and fq2[i].seq == fq[i].seq for i in ids[:2000])
    # by name on the object that built the index: statements until it has been used often enough, then a hash of the packed names
    names2 = [fq[i].name for i in ids2]
    t6 = time.perf_counter()
    for nm in names2:
        fq[nm].seq
    t6b = time.perf_counter()
    for nm in names2:
        fq[nm].seq
    t6c = time.perf_counter()
    same3 = all(fq[nm].id == fq2[nm].id and fq[nm].seq == fq2[nm].seq for nm in names2[:2000])
    ours = {"one_by_one_reads_per_s": round(k / (t4 - t3)), "one_by_one_seq_qual_quali_reads_per_s": round(k / (t4b - t4)),
            "one_by_one_by_name_reads_per_s": round(k / (t4d - t4c)),
            "loaded_index_one_by_one_reads_per_s_first_200k": round(k2 / (t5b - t5)), "loaded_index_one_by_one_reads_per_s_next_200k": round(k2 / (t5c - t5b)),
            "loaded_index_table_rows": int(fq2._core_table_rows), "loaded_index_answers_equal": bool(same2),
            "by_name_reads_per_s_first_200k": round(k2 / (t6b - t6)), "by_name_reads_per_s_next_200k": round(k2 / (t6c - t6b)), "by_name_answers_equal": bool(same3),
            "single_getters_answered_by": "page cache (csrc/fxobj.c)" if fq._core_fd >= 0 else "resident kernel"}
    # the compiled reference on the same file, same box (its own index file)
    ref = {}
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref"))
        import pyfastx
        rix = pq + ".ref.fxi"
        if os.path.exists(rix):
            os.remove(rix)
        r0 = time.perf_counter()
        rq = pyfastx.Fastq(pq, index_file=rix)
        r1 = time.perf_counter()
        tot = 0
        for r in rq:
            tot += len(r.seq) + len(r.qual)
        r2 = time.perf_counter()
        assert tot == 2 * 150 * n
        for i in ids:
            rq[i].seq
        r3 = time.perf_counter()
        for i in ids:
            r = rq[i]
            r.seq, r.qual, r.quali
        r4 = time.perf_counter()
        for nm in rnames:
            rq[nm].seq
        r5 = time.perf_counter()
        same = all(rq[i].seq == fq[i].seq and rq[i].qual == fq[i].qual and rq[i].quali == fq[i].quali and rq[i].name == fq[i].name for i in ids[:2000])
        ref = {"reference_Fastq_ctor_s": round(r1 - r0, 2), "reference_iterate_seq_qual_M_reads_per_s": round(n / (r2 - r1) / 1e6, 3),
               "reference_one_by_one_reads_per_s": round(k / (r3 - r2)), "reference_one_by_one_seq_qual_quali_reads_per_s": round(k / (r4 - r3)),
               "reference_one_by_one_by_name_reads_per_s": round(k / (r5 - r4)), "answers_equal_reference_2000": bool(same)}
        del rq
        os.remove(rix)
    except Exception as e:  # noqa: BLE001
        ref = {"reference": str(e)[:120]}
    # FASTA: many short records
    rng = np.random.default_rng(2)
    m = n // 4
    seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (m, 300))]
    pa = os.path.join(d, "iter_rate.fa")
    with open(pa, "wb") as f:
        rows = np.full((m, 5, 61), 10, dtype=np.uint8)
        rows[:, :, :60] = seqs.reshape(m, 5, 60)
        body = rows.reshape(m, 305)
        for a in range(0, m, 100000):
            b = min(m, a + 100000)
            f.write(b"".join(b">r%d\n" % i + body[i - a + a].tobytes() for i in range(a, b)))
    if os.path.exists(pa + ".fxi"):
        os.remove(pa + ".fxi")
    t5 = time.perf_counter()
    fa = fx.Fasta(pa)
    t6 = time.perf_counter()
    tot = 0
    for s in fa:
        tot += len(s.seq)
    t7 = time.perf_counter()
    assert tot == 300 * m and len(fa) == m
    refa = {}
    try:
        import pyfastx
        rix = pa + ".ref.fxi"
        if os.path.exists(rix):
            os.remove(rix)
        r0 = time.perf_counter()
        ra = pyfastx.Fasta(pa, index_file=rix)
        r1 = time.perf_counter()
        tot = 0
        for s in ra:
            tot += len(s.seq)
        r2 = time.perf_counter()
        assert tot == 300 * m
        refa = {"reference_Fasta_ctor_s": round(r1 - r0, 2), "reference_iterate_seq_M_records_per_s": round(m / (r2 - r1) / 1e6, 3)}
        del ra
        os.remove(rix)
    except Exception as e:  # noqa: BLE001
        refa = {"reference_fasta": str(e)[:120]}
    out = {"fastq_reads": n, "Fastq_ctor_incl_fxi_s": round(t1 - t0, 2), "iterate_seq_qual_M_reads_per_s": round(n / (t2 - t1) / 1e6, 3)}
    out["iteration_breakdown"] = bd
    out.update(ours)
    out.update(ref)
    out.update({"fasta_records": m, "Fasta_ctor_incl_fxi_s": round(t6 - t5, 2), "iterate_seq_M_records_per_s": round(m / (t7 - t6) / 1e6, 3)})
    out.update(refa)
    print(json.dumps(out))
    for p in (pq, pq + ".fxi", pa, pa + ".fxi"):
        if os.path.exists(p):
            os.remove(p)


if __name__ == "__main__":
    main()
