#!/usr/bin/env python3
"""What the FIRST open of a large file costs a FRESH process (no torch, nothing allocated before): `make` writes a C3-shaped FASTQ
of n reads to /dev/shm (a process of its own), `open` -- another process -- times Blob.from_file twice and Fastq(path) twice.
`two` (round 6): two files of DIFFERENT 35 GB-class sizes opened and closed in turn in one process -- the allocation
wait of every open (the library keeps the blobs of closed streams under FX_SCRATCH_KEEP_BIG_MB: nothing is freed, nothing waits).
usage: python tools/first_open_probe.py make <reads> [path] | open [pretouch] | two"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PATH = "/dev/shm/fx_first_open.fq"


def main():
    if sys.argv[1] == "make":
        import torch
        from pyfastx_amd import synth
        n = int(float(sys.argv[2]))
        path = sys.argv[3] if len(sys.argv) > 3 else PATH
        blob, cols = synth.fastq_generate(n, torch.device("cuda", 0))
        nb = int(cols["n_bytes"])
        with open(path, "wb") as f:
            for x in range(0, nb, 1 << 30):
                f.write(memoryview(blob[x:min(x + (1 << 30), nb)].cpu().numpy()))
        print(json.dumps({"made": path, "GB": round(nb / 1e9, 2)}))
        return
    if sys.argv[1] == "two":
        os.environ["FX_NO_TORCH"] = "1"
        from pyfastx_amd import _lib
        a, b = PATH, PATH + ".2"
        out = {"files_GB": [round(os.path.getsize(a) / 1e9, 2), round(os.path.getsize(b) / 1e9, 2)], "opens": []}
        for p in (a, b, a, b, b, a):
            t0 = time.perf_counter()
            h = _lib.Blob.from_file(p)
            t1 = time.perf_counter()
            al, st = _lib.open_laps()
            free, total = _lib.device_memory(0)
            h.close()
            out["opens"].append({"GB": round(os.path.getsize(p) / 1e9, 1), "open_s": round(t1 - t0, 3), "allocation_wait_s": round(al, 4), "page_cache_to_hbm_s": round(st, 3),
                                 "device_free_GB_while_open": round(free / 1e9, 1)})
        out["max_allocation_wait_s"] = max(o["allocation_wait_s"] for o in out["opens"])
        print(json.dumps(out))
        return
    os.environ["FX_NO_TORCH"] = "1"
    import pyfastx_amd as fx
    from pyfastx_amd import _lib
    out = {"file_GB": round(os.path.getsize(PATH) / 1e9, 2), "env": {k: v for k, v in os.environ.items() if k.startswith("FX_")}}
    runs = []
    for rep in range(3):
        t0 = time.perf_counter()
        b = _lib.Blob.from_file(PATH)
        t1 = time.perf_counter()
        al, st = _lib.open_laps()
        b.close()
        runs.append({"open_s": round(t1 - t0, 3), "device_alloc_s": round(al, 3), "page_cache_to_hbm_s": round(st, 3)})
    out["Blob_from_file"] = runs
    ctor = []
    for rep in range(2):
        if os.path.exists(PATH + ".fxi"):
            os.remove(PATH + ".fxi")
        t0 = time.perf_counter()
        fq = fx.Fastq(PATH)
        t1 = time.perf_counter()
        ctor.append({"ctor_s": round(t1 - t0, 3), **{k: round(v, 3) for k, v in (fq.build_phases or {}).items() if isinstance(v, float)}})
        del fq
    out["Fastq_ctor"] = ctor
    os.remove(PATH + ".fxi")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
