#!/usr/bin/env python3
"""End-to-end (file on disk -> index usable, PCIe and page cache included) timing of the object API on
one MI355X, next to the real reference on the same host if oracle/_ref loads.
usage: python tools/e2e_file.py [gbp]"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyfastx_amd as fx  # noqa: E402
from pyfastx_amd import _lib, synth  # noqa: E402


def main():
    gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    dev = torch.device("cuda", 0)
    plan = synth.fasta_plan(total_bp=int(gbp * 1e9))
    blob_t, _, _ = synth.fasta_generate(plan, dev, keep_flat=False)
    nb = int(plan["n_bytes"])
    d = tempfile.mkdtemp(prefix="fxe2e")
    path = os.path.join(d, "c2.fa")
    blob_t[:nb].cpu().numpy().tofile(path)
    del blob_t
    torch.cuda.empty_cache()
    out = {"file_GB": round(nb / 1e9, 3)}
    _lib.Blob.from_file(path).close()                       # warm the page cache, first-touch the driver
    t0 = time.perf_counter()
    b = _lib.Blob.from_file(path)
    t1 = time.perf_counter()
    s = b.fasta_build()
    t = b.fasta_table(s.n_seq)
    t2 = time.perf_counter()
    out["stage_file_to_hbm_s"] = round(t1 - t0, 4)
    out["stage_GBps"] = round(nb / (t1 - t0) / 1e9, 2)
    out["build_resident_s"] = round(t2 - t1, 4)
    b.close()
    t0 = time.perf_counter()
    fa = fx.Fasta(path)                                     # stage + build + names + .fxi on disk
    t1 = time.perf_counter()
    out["Fasta_ctor_incl_fxi_s"] = round(t1 - t0, 4)
    ids, st, sp, strand = synth.fasta_queries(plan, n=1_000_000)
    names = plan["names"]
    t0 = time.perf_counter()
    buf, offs = fa.fetch_many([names[i] for i in ids], st, sp, strand=strand)
    t1 = time.perf_counter()
    out["fetch_many_1M_host_arrays_s"] = round(t1 - t0, 4)
    t0 = time.perf_counter()
    for j in range(20000):
        s_ = fa[names[ids[j]]][int(st[j]):int(sp[j])]
        _ = s_.antisense if strand[j] else s_.seq
    t1 = time.perf_counter()
    out["per_object_fetch_per_s"] = round(20000 / (t1 - t0))
    os.unlink(path + ".fxi")
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
        import pyfastx
        t0 = time.perf_counter()
        ref = pyfastx.Fasta(path)
        t1 = time.perf_counter()
        out["reference_Fasta_ctor_s"] = round(t1 - t0, 4)
        # spot parity: 2000 of the batched answers against the reference object API
        ok = True
        for j in range(0, 1_000_000, 500):
            s_ = ref[names[ids[j]]][int(st[j]):int(sp[j])]
            want = s_.antisense if strand[j] else s_.seq
            ok &= buf[offs[j]:offs[j + 1]].tobytes().decode() == want
        out["batched_answers_equal_reference_sample"] = bool(ok)
        del ref
        os.unlink(path + ".fxi")
    except Exception as e:
        out["reference"] = "unavailable: %s" % str(e)[:80]
    os.unlink(path)
    os.rmdir(d)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
