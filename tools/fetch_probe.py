#!/usr/bin/env python3
"""The two gather kernels on their BASELINE workloads, device-resident, for profilers and tuning:
k_fetch (C2: 3 Gbp FASTA, random 100 bp intervals by record id) and k_fastq_fetch (C3 shape: 150 bp reads, seq + qual + quali),
each with 1 M and 16 M queries; optionally the same queries sorted by offset.   usage: python tools/fetch_probe.py [gbp] [reads]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfastx_amd import _lib, synth  # noqa: E402


def main():
    gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    nreads = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20_000_000
    dev = torch.device("cuda", 0)
    out = {}
    plan = synth.fasta_plan(total_bp=int(gbp * 1e9))
    blob_t, _, _ = synth.fasta_generate(plan, dev, keep_flat=False)
    b = _lib.Blob.from_device(blob_t.data_ptr(), int(plan["n_bytes"]), device=0, keepalive=blob_t)
    b.fasta_build()
    L = _lib.lib()
    for nq in (1_000_000, 16_000_000):
        ids, st, sp, strand = synth.fasta_queries(plan, n=nq)
        for tag, order in (("random", None), ("sorted_by_offset", np.argsort(plan["boff"][ids] + st, kind="stable"))):
            if order is not None:
                ids, st, sp, strand = ids[order], st[order], sp[order], strand[order]
            d = [torch.from_numpy(x).to(dev) for x in (ids, st, sp)]
            fl = torch.from_numpy((strand * 6).astype(np.uint8)).to(dev)
            off = torch.arange(nq, device=dev, dtype=torch.int64) * 100
            dst = torch.zeros(nq * 100, dtype=torch.uint8, device=dev)
            ol = torch.zeros(nq, dtype=torch.int64, device=dev)
            b.prof_enable(1); b.prof_reset()
            for _ in range(5):
                b.fasta_fetch_dev(nq, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), dst.data_ptr(), off.data_ptr(),
                                  flags_per_query=fl.data_ptr(), out_len=ol.data_ptr())
            b.sync()
            ms = b.prof_read()["k_fetch"]
            out["k_fetch_%dM_%s_ms" % (nq // 1_000_000, tag)] = round(ms[0] / ms[1], 4)
            b.prof_enable(0)
            del d, fl, off, dst, ol
    b.close()
    del blob_t
    torch.cuda.empty_cache()
    if os.environ.get("FX_PROBE_FASTA_ONLY"):
        print(json.dumps(out))
        return
    bq, cols = synth.fastq_generate(nreads, dev)
    q = _lib.Blob.from_device(bq.data_ptr(), cols["n_bytes"], device=0, keepalive=bq)
    q.fastq_build()
    for nq in (1_000_000, 16_000_000):
        rng = np.random.default_rng(99)
        ids = rng.integers(0, nreads, nq)
        for tag in ("random", "sorted_by_offset"):
            if tag != "random":
                ids = np.sort(ids)
            d_ids = torch.from_numpy(ids).to(dev)
            off = torch.arange(nq, device=dev, dtype=torch.int64) * 150
            o = [torch.zeros(nq * 150, dtype=torch.uint8, device=dev) for _ in range(2)] + [torch.zeros(nq * 150, dtype=torch.int8, device=dev)]
            q.prof_enable(1); q.prof_reset()
            for _ in range(5):
                _lib.check(L.fx_fastq_fetch(q._h, _lib.FX_DEVICE, nq, d_ids.data_ptr(), 33, 0, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), off.data_ptr()))
            q.sync()
            ms = q.prof_read()["k_fastq_fetch"]
            out["k_fastq_fetch_%dM_%s_ms" % (nq // 1_000_000, tag)] = round(ms[0] / ms[1], 4)
            q.prof_enable(0)
            del d_ids, off, o
    print(json.dumps(out))


if __name__ == "__main__":
    main()
