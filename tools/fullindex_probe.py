#!/usr/bin/env python3
"""FASTA index + composition on the C2 shape resident in HBM: the two-read form (k_span_scan, then k_fasta_comp) and the
one-read form (k_scan_comp + k_comp_attribute), per-kernel averages and wall time, rows compared.
usage: python tools/fullindex_probe.py [gbp]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfastx_amd import _lib, synth  # noqa: E402


def main():
    gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    dev = torch.device("cuda", 0)
    plan = synth.fasta_plan(total_bp=int(gbp * 1e9))
    blob_t, _, _ = synth.fasta_generate(plan, dev, keep_flat=False)
    nb = int(plan["n_bytes"])
    b = _lib.Blob.from_device(blob_t.data_ptr(), nb, device=0, keepalive=blob_t)
    nrec = len(plan["slen"])
    c1 = torch.zeros((nrec, 128), dtype=torch.int64, device=dev)
    c2 = torch.zeros_like(c1)
    out = {"GB": round(nb / 1e9, 3)}
    for name, comp, dst in (("two_reads", False, c1), ("one_read", True, c2)):
        b.fasta_build(False, comp=comp); b.fasta_comp_dev(dst.data_ptr()); b.sync()
        b.prof_enable(True); b.prof_reset()
        for _ in range(5):
            b.fasta_build(False, comp=comp); b.fasta_comp_dev(dst.data_ptr())
        b.sync()
        prof = {k: round(v[0] / v[1], 4) for k, v in b.prof_read().items()}
        b.prof_enable(False)
        t0 = time.perf_counter()
        for _ in range(5):
            b.fasta_build(False, comp=comp); b.fasta_comp_dev(dst.data_ptr())
        b.sync()
        out[name] = {"wall_ms": round((time.perf_counter() - t0) / 5 * 1e3, 3), "kernels_ms_avg": prof}
    out["rows_equal"] = bool((c1 == c2).all())
    print(json.dumps(out))
    if not out["rows_equal"]:
        raise SystemExit("PARITY FAILURE")


if __name__ == "__main__":
    main()
