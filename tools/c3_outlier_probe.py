#!/usr/bin/env python3
"""`Fastq(path, full_index=True)` on the 0.7 GB sample file took 5.9 s in one bench run and 0.17 s in the
others.  This replays what bench.py's C3 leg does in front of that constructor -- the 34.8 GB configuration resident in HBM,
built, closed, its tensors freed -- and times the constructor's parts (FX_TRACE_ALLOC=1 prints every pool miss).
usage: python tools/c3_outlier_probe.py [reads] [sample]"""
import json
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyfastx_amd as fx  # noqa: E402
from pyfastx_amd import _lib, synth  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    m = int(float(sys.argv[2])) if len(sys.argv) > 2 else 2_000_000
    full = len(sys.argv) > 3 and sys.argv[3] == "full"
    dev = torch.device("cuda", 0)
    blob_t, cols = synth.fastq_generate(n, dev)
    nb = int(cols["n_bytes"])
    rec = int(cols["rec"])
    b = _lib.Blob.from_device(blob_t.data_ptr(), nb, device=0, keepalive=blob_t)
    b.fastq_build(); b.fastq_comp(); b.fastq_build(comp=True); b.fastq_comp()
    d = tempfile.mkdtemp(prefix="fxout", dir="/dev/shm")
    path = os.path.join(d, "c3.fq")
    blob_t[:m * rec].cpu().numpy().tofile(path)
    if full:
        with open(os.path.join(d, "c3_full.fq"), "wb") as f:
            for x in range(0, nb, 1 << 30):
                f.write(memoryview(blob_t[x:min(x + (1 << 30), nb)].cpu().numpy()))
    out = {"reads": n, "sample": m, "wrote_full_file": full}
    t = time.perf_counter()
    b.close()
    del b, blob_t
    torch.cuda.empty_cache()
    out["close_and_free_s"] = round(time.perf_counter() - t, 3)
    t = time.perf_counter()
    _lib.Blob.from_file(path).close()
    out["warm_open_close_s"] = round(time.perf_counter() - t, 3)
    runs = []
    for rep in range(3):
        if os.path.exists(path + ".fxi"):
            os.remove(path + ".fxi")
        t0 = time.perf_counter()
        fq = fx.Fastq(path, full_index=True)
        t1 = time.perf_counter()
        runs.append({"ctor_s": round(t1 - t0, 3), "build_phases": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (fq.build_phases or {}).items()},
                     "index_phases": None if fq.index_phases is None else {k: round(v, 4) for k, v in fq.index_phases.items()}})
        del fq
    out["runs"] = runs
    os.remove(path + ".fxi"); os.remove(path)
    if full:                                                # the full file: first and second constructor of the process (FX_TRACE=1: blob / staging laps on stderr)
        fp = os.path.join(d, "c3_full.fq")
        out["full_runs"] = []
        for rep in range(3):
            if os.path.exists(fp + ".fxi"):
                os.remove(fp + ".fxi")
            t0 = time.perf_counter()
            fq = fx.Fastq(fp)
            t1 = time.perf_counter()
            out["full_runs"].append({"ctor_s": round(t1 - t0, 3), "build_phases": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in (fq.build_phases or {}).items()}})
            del fq
        os.remove(fp + ".fxi")
        os.remove(os.path.join(d, "c3_full.fq"))
    os.rmdir(d)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
