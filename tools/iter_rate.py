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
    k = 20000
    ids = np.random.default_rng(1).integers(0, n, k).tolist()
    t3 = time.perf_counter()
    for i in ids:
        fq[i].seq
    t4 = time.perf_counter()
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
    print(json.dumps({"fastq_reads": n, "Fastq_ctor_incl_fxi_s": round(t1 - t0, 2), "iterate_seq_qual_M_reads_per_s": round(n / (t2 - t1) / 1e6, 3),
                      "one_by_one_reads_per_s": round(k / (t4 - t3)), "fasta_records": m, "Fasta_ctor_incl_fxi_s": round(t6 - t5, 2),
                      "iterate_seq_M_records_per_s": round(m / (t7 - t6) / 1e6, 3)}))
    for p in (pq, pq + ".fxi", pa, pa + ".fxi"):
        if os.path.exists(p):
            os.remove(p)


if __name__ == "__main__":
    main()
