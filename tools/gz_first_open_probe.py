#!/usr/bin/env python3
"""Where the first open of a single-stream gzip FASTA spends its time: Fasta(path) under cProfile, with the phases of the
parallel inflate (FX_TRACE_PGZ=1) on stderr.  usage: python tools/gz_first_open_probe.py [gbp]   (default 1.0)"""
import cProfile
import os
import pstats
import sys
import tempfile
import time
from multiprocessing import get_context

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FX_TRACE_PGZ", "1")
from pyfastx_amd import synth  # noqa: E402
import pyfastx_amd as fx  # noqa: E402


def main():
    gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    dev = torch.device("cuda", 0)
    plan = synth.fasta_plan(total_bp=int(gbp * 1e9), seed=20260612)
    blob, _, _ = synth.fasta_generate(plan, dev, keep_flat=False)
    host = blob[:int(plan["n_bytes"])].cpu().numpy()
    del blob
    gz = synth.gzip_single_stream_parallel(host, procs=32)
    d = tempfile.mkdtemp(prefix="fxgz", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    p = os.path.join(d, "s.fa.gz")
    with open(p, "wb") as f:
        f.write(gz)
    print("file: %.2f GB compressed, %.2f GB inflated" % (len(gz) / 1e9, host.size / 1e9), flush=True)
    del gz
    settings = [None] + [int(x) for x in os.environ.get("FX_PROBE_THREADS", "").split(",") if x]
    for rep in range(2 * len(settings)):
        th = settings[rep // 2]
        if th is None:
            os.environ.pop("FX_PGZ_THREADS", None)
        else:
            os.environ["FX_PGZ_THREADS"] = str(th)
        print("threads:", th or "default", flush=True)
        if os.path.exists(p + ".fxi"):
            os.remove(p + ".fxi")
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        pr.enable()
        fa = fx.Fasta(p)
        pr.disable()
        t1 = time.perf_counter()
        print("Fasta(path) %.3f s" % (t1 - t0), flush=True)
        del fa
        if rep == 2 * len(settings) - 1 and not os.environ.get('FX_PROBE_THREADS'):
            pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
    os.remove(p); os.remove(p + ".fxi"); os.rmdir(d)


if __name__ == "__main__":
    main()
