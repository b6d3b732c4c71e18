#!/usr/bin/env python3
"""Where the time of Fastq(path) on a large FASTQ file goes, phase by phase (C ABI calls timed one by one), and the whole
constructor beside it.  usage: python tools/c3_phases.py [reads] [dir]   (default 5e7 reads, /dev/shm)"""
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
from pyfastx_amd import _lib, fxi, synth  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
    where = sys.argv[2] if len(sys.argv) > 2 else "/dev/shm"
    dev = torch.device("cuda", 0)
    blob_t, cols = synth.fastq_generate(n, dev)
    nb = int(cols["n_bytes"])
    d = tempfile.mkdtemp(prefix="fxc3p", dir=where)
    path = os.path.join(d, "c3.fq")
    with open(path, "wb") as f:
        for x in range(0, nb, 1 << 30):
            f.write(memoryview(blob_t[x:min(x + (1 << 30), nb)].cpu().numpy()))
    del blob_t
    torch.cuda.empty_cache()
    out = {"reads": n, "file_GB": round(nb / 1e9, 2), "cpus": len(os.sched_getaffinity(0))}
    _lib.Blob.from_file_range(path, 0, 1 << 24, 0).close()
    lap = {}
    t = time.perf_counter()

    def mark(k):
        nonlocal t
        t2 = time.perf_counter()
        lap[k] = round(t2 - t, 4)
        t = t2
    b = _lib.Blob.from_file(path); mark("stage")
    s = b.fastq_build(); mark("build")
    tab = b.fastq_table(s.n_reads); mark("table_to_host")
    packed, offs = b.names_pack(1, s.n_reads, guess=int(np.maximum(tab["name_len"], 0).sum())); mark("names_pack")
    order, ndup = b.names_sort(1, s.n_reads); mark("names_sort")
    p = path + ".fxi"
    db = fxi.connect(p)
    db.executescript(fxi.FASTQ_DDL)
    db.execute("CREATE UNIQUE INDEX readidx ON read (name)")
    root = dict(db.execute("SELECT name, rootpage FROM sqlite_master").fetchall())
    db.close(); mark("sqlite_schema")
    _lib.fxi_bulk_rows(p, root["read"], packed, offs, [tab["dlen"], tab["rlen"], tab["soff"], tab["qoff"]]); mark("bulk_rows")
    _lib.fxi_bulk_index(p, root["readidx"], packed, offs, order); mark("bulk_index")
    out["fxi_GB"] = round(os.path.getsize(p) / 1e9, 2)
    out["phases_s"] = lap
    out["sum_s"] = round(sum(lap.values()), 3)
    b.close()
    os.unlink(p)
    del tab, packed, offs, order
    out["ctor_runs"] = []
    for rep in range(int(os.environ.get("C3_REPS", 2))):
        # pipelined: the ranges of the input are indexed and their table leaves shipped while the rest still arrives (round 6);
        # pipelined_wait_room: the same, nothing written before the whole index file is allocated; presize: one blob, one build
        # (round 5); no_presize: and the index file grown only when its size is known
        for mode in os.environ.get("C3_MODES", "pipelined,pipelined_wait_room,presize,no_presize").split(","):
            if os.path.exists(p):
                os.unlink(p)
            for k in ("FX_FXI_NO_PRESIZE", "FX_FQ_NO_PIPELINE", "FX_FQ_PIPELINE_WAIT_ROOM", "FX_FQ_PIPELINE"):
                os.environ.pop(k, None)
            if mode.startswith("pipelined"):
                os.environ["FX_FQ_PIPELINE"] = "1"
            if mode == "no_presize":
                os.environ["FX_FXI_NO_PRESIZE"] = "1"
            elif mode == "presize":
                os.environ["FX_FQ_NO_PIPELINE"] = "1"
            elif mode == "pipelined_wait_room":
                os.environ["FX_FQ_PIPELINE_WAIT_ROOM"] = "1"
            t0 = time.perf_counter()
            fq = fx.Fastq(path)
            t1 = time.perf_counter()
            out["ctor_runs"].append({"mode": mode, "Fastq_ctor_s": round(t1 - t0, 3),
                                     "build_phases": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in (fq.build_phases or {}).items()},
                                     "index_phases": None if fq.index_phases is None else {k: round(v, 4) for k, v in fq.index_phases.items()}})
            del fq
    out["fxi_GB_dev"] = round(os.path.getsize(p) / 1e9, 2)
    import sqlite3
    db = sqlite3.connect(p)
    i = n // 3
    out["probe_ok"] = db.execute("SELECT ID FROM read WHERE name=(SELECT name FROM read WHERE ID=?)", (i,)).fetchone()[0] == i
    db.close()
    os.unlink(p)
    os.unlink(path)
    os.rmdir(d)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
