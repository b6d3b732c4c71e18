#!/usr/bin/env python3
"""What the FIRST open of a large file costs a FRESH process (no torch, nothing allocated before): `make` writes a C3-shaped FASTQ
of n reads to /dev/shm (a process of its own), `open` -- another process -- times Blob.from_file twice and Fastq(path) twice.
usage: python tools/first_open_probe.py make <reads> | open [pretouch]"""
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
        blob, cols = synth.fastq_generate(n, torch.device("cuda", 0))
        nb = int(cols["n_bytes"])
        with open(PATH, "wb") as f:
            for x in range(0, nb, 1 << 30):
                f.write(memoryview(blob[x:min(x + (1 << 30), nb)].cpu().numpy()))
        print(json.dumps({"made": PATH, "GB": round(nb / 1e9, 2)}))
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
