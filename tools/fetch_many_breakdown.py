#!/usr/bin/env python3
"""Where a batched fetch spends its time, host arrays to host buffers: `Fasta.fetch_many` of 1 M random
100-base intervals on the C2 shape -- by id and by 1 M `str` names -- and `Fastq.fetch_many` of 1 M reads, each with the
library's own host-side phases (fx_fetch_phases), the kernel time (HIP events through the handle's profiler), the Python
share (call time minus the C call), a cold first call next to the warm ones, and the answers compared with the plain
caller-allocated entry points.
usage: python tools/fetch_many_breakdown.py [gbp] [fastq reads] > profiles/r04_fetch_many_breakdown.json"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyfastx_amd as fx  # noqa: E402
from pyfastx_amd import _lib, synth  # noqa: E402


def timed(f, reps=5):
    ts, ph = [], []
    out = None
    for _ in range(reps):
        out = None                                        # the previous answer goes back to the pinned pool first
        t0 = time.perf_counter()
        out = f()
        ts.append(time.perf_counter() - t0)
        ph.append(_lib.fetch_phases())
    k = int(np.argsort(ts)[len(ts) // 2])
    return out, ts, ph[k]


def kernel_ms(blob, f):
    blob.prof_enable(1)
    blob.prof_reset()
    f()
    blob.sync()
    d = blob.prof_read()
    blob.prof_enable(0)
    return {k: round(v[0], 4) for k, v in d.items() if v[1]}


def main():
    gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    n_reads = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
    nq = 1_000_000
    dev = torch.device("cuda", 0)
    res = {"what": "host arrays -> host buffers, %d queries; medians of 5 calls after one cold call" % nq}
    d = tempfile.mkdtemp(prefix="fxfm", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    # ------------------------------------------------------------------ FASTA
    plan = synth.fasta_plan(total_bp=int(gbp * 1e9))
    blob_t, _, _ = synth.fasta_generate(plan, dev, keep_flat=False)
    nb = int(plan["n_bytes"])
    path = os.path.join(d, "c2.fa")
    blob_t[:nb].cpu().numpy().tofile(path)
    del blob_t
    torch.cuda.empty_cache()
    fa = fx.Fasta(path)
    ids, st, sp, strand = synth.fasta_queries(plan, n=nq)
    names = [plan["names"][i] for i in ids]
    t0 = time.perf_counter()
    buf0, offs0 = fa.fetch_many(ids, st, sp, strand=strand)
    cold = time.perf_counter() - t0
    blob = fa._st.blob
    fl = np.where(strand != 0, 6, 0).astype(np.uint8)
    ref_buf, ref_offs, _ = blob.fasta_fetch(ids, st, sp, flags=0, flags_per_query=fl)      # caller-allocated pageable buffers
    same = bool(np.array_equal(ref_offs, np.asarray(offs0)) and np.array_equal(ref_buf, np.asarray(buf0)))
    del buf0, offs0
    (b1, o1), t_id, ph_id = timed(lambda: fa.fetch_many(ids, st, sp, strand=strand))
    del b1, o1
    (b2, o2), t_nm, ph_nm = timed(lambda: fa.fetch_many(names, st, sp, strand=strand))
    same = same and bool(np.array_equal(ref_buf, np.asarray(b2)))
    del b2, o2
    t0 = time.perf_counter()
    for _ in range(3):
        blob.fasta_fetch(ids, st, sp, flags=0, flags_per_query=fl)
    t_old = (time.perf_counter() - t0) / 3
    km = kernel_ms(blob, lambda: fa.fetch_many(ids, st, sp, strand=strand))
    med = lambda xs: float(np.median(xs))
    res["fasta"] = {
        "file_bytes": nb, "answers_bytes": int(ref_offs[-1]), "query_bytes_up": int(ids.nbytes + st.nbytes + sp.nbytes + fl.nbytes),
        "cold_first_call_ms": round(cold * 1e3, 2),
        "by_id_ms": round(med(t_id) * 1e3, 3), "by_id_all_ms": [round(x * 1e3, 2) for x in t_id], "by_id_phases": ph_id,
        "by_id_python_share_ms": round(med(t_id) * 1e3 - ph_id["call_ms"], 3),
        "by_1M_str_names_ms": round(med(t_nm) * 1e3, 3), "by_names_all_ms": [round(x * 1e3, 2) for x in t_nm], "by_names_phases": ph_nm,
        "by_names_python_share_ms": round(med(t_nm) * 1e3 - ph_nm["call_ms"], 3),
        "kernels_ms": km,
        "caller_allocated_pageable_buffers_ms": round(t_old * 1e3, 2),
        "answers_equal_the_plain_entry_point": same,
        "M_fetches_per_s_by_id": round(nq / med(t_id) / 1e6, 1), "M_fetches_per_s_by_names": round(nq / med(t_nm) / 1e6, 1),
    }
    del fa
    os.unlink(path)
    if os.path.exists(path + ".fxi"):
        os.unlink(path + ".fxi")
    # ------------------------------------------------------------------ FASTQ
    fqpath = os.path.join(d, "c3.fq")
    if n_reads > 0:
        blob_q, cols = synth.fastq_generate(n_reads, dev)
        blob_q[:int(cols["n_bytes"])].cpu().numpy().tofile(fqpath)
        del blob_q, cols
        torch.cuda.empty_cache()
        fq = fx.Fastq(fqpath)
        rid = np.random.default_rng(99).integers(0, n_reads, nq)
        t0 = time.perf_counter()
        g0 = fq.fetch_many(rid)
        coldq = time.perf_counter() - t0
        bq = fq._st.blob
        rl = fq._rlen_host[rid]
        rs, rq, ri, ro = bq.fastq_fetch(rid, rl, phred=fq._phred)
        sameq = bool(np.array_equal(ro, np.asarray(g0["offsets"])) and np.array_equal(rs, np.asarray(g0["seq"])) and
                     np.array_equal(rq, np.asarray(g0["qual"])) and np.array_equal(ri, np.asarray(g0["quali"])))
        tot = int(ro[-1])
        del g0
        g, t_q, ph_q = timed(lambda: fq.fetch_many(rid))
        del g
        g, t_s, ph_s = timed(lambda: fq.fetch_many(rid, want=("seq",)))
        del g
        qnames = [fq[int(i)].name for i in rid[:200_000]]
        g, t_qn, ph_qn = timed(lambda: fq.fetch_many(qnames), reps=3)
        sameq = sameq and bool(np.array_equal(np.asarray(g["seq"]), rs[:int(ro[200_000])]))
        del g
        res["fastq"] = {
            "reads_in_file": n_reads, "answers_bytes": 3 * tot, "cold_first_call_ms": round(coldq * 1e3, 2),
            "seq_qual_quali_ms": round(med(t_q) * 1e3, 3), "all_ms": [round(x * 1e3, 2) for x in t_q], "phases": ph_q,
            "python_share_ms": round(med(t_q) * 1e3 - ph_q["call_ms"], 3),
            "seq_only_ms": round(med(t_s) * 1e3, 3), "seq_only_phases": ph_s,
            "by_200k_str_names_ms": round(med(t_qn) * 1e3, 3),
            "answers_equal_the_plain_entry_point": sameq,
            "M_reads_per_s": round(nq / med(t_q) / 1e6, 1),
        }
        del fq
        os.unlink(fqpath)
        if os.path.exists(fqpath + ".fxi"):
            os.unlink(fqpath + ".fxi")
    os.rmdir(d)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
