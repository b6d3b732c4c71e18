#!/usr/bin/env python3
"""FASTQ index build only, C3 shape: wall time per build, per-kernel averages, rows against the generator's truth.
FX_FQ_LINES=0 selects the two-read build (count-only pass + k_fastq_emit over every granule).   usage: python tools/fq_build_bench.py [n_reads]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfastx_amd import _lib, synth  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
    dev = torch.device("cuda", 0)
    blob_t, cols = synth.fastq_generate(n, dev)
    nb = cols["n_bytes"]
    b = _lib.Blob.from_device(blob_t.data_ptr(), nb, device=0, keepalive=blob_t)
    b.fastq_build()
    b.prof_enable(True); b.prof_reset()
    R = 10
    t0 = time.perf_counter()
    for _ in range(R):
        s = b.fastq_build()
    t1 = time.perf_counter()
    prof = {k: round(v[0] / v[1], 4) for k, v in b.prof_read().items()}
    ok = (s.n_reads, s.size) == (n, n * 150)
    t = b.fastq_table(n)
    for k in ("name_off", "name_len", "dlen", "rlen", "soff", "qoff"):
        ok &= bool((t[k] == cols[k]).all())
    print(json.dumps({"reads": n, "GB": round(nb / 1e9, 2), "FX_FQ_LINES": os.environ.get("FX_FQ_LINES", "auto"),
                      "build_ms": round((t1 - t0) / R * 1e3, 3), "GBps": round(nb / ((t1 - t0) / R) / 1e9, 1), "kernels_ms_avg": prof,
                      "rows_equal_truth": ok}))
    if not ok:
        raise SystemExit("PARITY FAILURE")


if __name__ == "__main__":
    main()
