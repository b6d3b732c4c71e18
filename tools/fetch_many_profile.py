#!/usr/bin/env python3
"""Where the time of Fasta.fetch_many (1 M queries by name, host arrays to host buffer) goes: cProfile of the call on the
C2 shape.  usage: python tools/fetch_many_profile.py [gbp]"""
import cProfile
import os
import pstats
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyfastx_amd as fx  # noqa: E402
from pyfastx_amd import synth  # noqa: E402


def main():
    gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    dev = torch.device("cuda", 0)
    plan = synth.fasta_plan(total_bp=int(gbp * 1e9))
    blob_t, _, _ = synth.fasta_generate(plan, dev, keep_flat=False)
    nb = int(plan["n_bytes"])
    d = tempfile.mkdtemp(prefix="fxfm")
    path = os.path.join(d, "c2.fa")
    blob_t[:nb].cpu().numpy().tofile(path)
    del blob_t
    torch.cuda.empty_cache()
    fa = fx.Fasta(path)
    ids, st, sp, strand = synth.fasta_queries(plan, n=1_000_000)
    names = [plan["names"][i] for i in ids]
    fa.fetch_many(names, st, sp, strand=strand)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        fa.fetch_many(names, st, sp, strand=strand)
        ts.append(time.perf_counter() - t0)
    print("fetch_many 1M by name: %.1f ms (median of 3)" % (sorted(ts)[1] * 1e3))
    t0 = time.perf_counter()
    fa.fetch_many(ids, st, sp, strand=strand)
    print("fetch_many 1M by id:   %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    fa.fetch_many(names, st, sp, strand=strand)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)


if __name__ == "__main__":
    main()
