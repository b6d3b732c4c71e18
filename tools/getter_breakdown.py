#!/usr/bin/env python3
"""Where the time of one per-object getter goes (fa[name][s:e].seq, the reference's benchmark idiom):
usage: python tools/getter_breakdown.py [gbp]"""
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


def rate(f, n):
    t0 = time.perf_counter()
    for j in range(n):
        f(j)
    return (time.perf_counter() - t0) / n * 1e6


def main():
    gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
    dev = torch.device("cuda", 0)
    plan = synth.fasta_plan(total_bp=int(gbp * 1e9))
    blob_t, _, _ = synth.fasta_generate(plan, dev, keep_flat=False)
    nb = int(plan["n_bytes"])
    d = tempfile.mkdtemp(prefix="fxget")
    path = os.path.join(d, "c2.fa")
    blob_t[:nb].cpu().numpy().tofile(path)
    del blob_t
    fa = fx.Fasta(path)
    ids, st, sp, strand = synth.fasta_queries(plan, n=20000)
    names = plan["names"]
    ii, ss, ee = ids.tolist(), st.tolist(), sp.tolist()
    N = 20000
    out = {}
    fa[names[0]][0:10].seq
    out["full_getter_us"] = rate(lambda j: fa[names[ii[j]]][ss[j]:ee[j]].seq, N)
    out["subscript_by_name_us"] = rate(lambda j: fa[names[ii[j]]], N)
    s0 = fa[names[0]]
    out["slice_object_us"] = rate(lambda j: s0[ss[j] % 1000:ss[j] % 1000 + 100], N)
    sl = [fa[names[ii[j]]][ss[j]:ee[j]] for j in range(N)]
    out["seq_of_a_slice_us"] = rate(lambda j: sl[j].seq, N)
    b = fa._st.blob
    rows = [(s._range(s.start - 1, s.end)) for s in sl]
    out["blob_fetch_one_us"] = rate(lambda j: b.fetch_one(rows[j][0], rows[j][1], 100), N)
    L = _lib.lib()
    out["ctypes_trivial_call_us"] = rate(lambda j: L.fx_size(b._h), N)
    # the floor under ANY single getter: fx_fetch_one called from C in a loop (request line -> resident kernel -> answer
    # through pinned memory: the PCIe round trips and the gather, no interpreter, no object)
    from pyfastx_amd import _fxobj
    out["fetch_one_from_C_floor_us"] = _fxobj.bench_fetch_one(int(b._h.value), rows[0][0], rows[0][1], 100, N)
    out["getters_per_s"] = round(1e6 / out["full_getter_us"])
    out["answered_by"] = "page cache (plain file: pread + byte work in csrc/fxobj.c)" if fa._core_fd >= 0 else "resident kernel (fx_fetch_one)"
    out["antisense_getter_us"] = rate(lambda j: fa[names[ii[j]]][ss[j]:ee[j]].antisense, N)
    # the same getters with the host path switched off: everything through the resident kernel (what a gzip input gets)
    fa._core_stage(int(b._h.value), None)
    out["full_getter_through_the_resident_kernel_us"] = rate(lambda j: fa[names[ii[j]]][ss[j]:ee[j]].seq, N)
    fa._core_stage(int(b._h.value), path)
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
        import pyfastx
        os.unlink(path + ".fxi")
        rf = pyfastx.Fasta(path)
        out["reference_full_getter_us"] = rate(lambda j: rf[names[ii[j]]][ss[j]:ee[j]].seq, N)
        out["reference_antisense_getter_us"] = rate(lambda j: rf[names[ii[j]]][ss[j]:ee[j]].antisense, N)
        out["reference_subscript_by_name_us"] = rate(lambda j: rf[names[ii[j]]], N)
        same = all(rf[names[ii[j]]][ss[j]:ee[j]].seq == fa[names[ii[j]]][ss[j]:ee[j]].seq and
                   rf[names[ii[j]]][ss[j]:ee[j]].antisense == fa[names[ii[j]]][ss[j]:ee[j]].antisense for j in range(2000))
        out["answers_equal_reference_2000"] = bool(same)
        assert same
    except Exception as e:  # noqa: BLE001
        out["reference"] = str(e)[:100]
    print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in out.items()}))
    for f in (path, path + ".fxi"):
        if os.path.exists(f):
            os.unlink(f)
    os.rmdir(d)


if __name__ == "__main__":
    main()
