#!/usr/bin/env python3
"""One Fastq(path) of N synthetic reads with the index file written from the device (k_fxi_*): the process the PMC passes of
tools/gpu.sh pmc_fxi run.  usage: python tools/fxi_pmc_probe.py [reads] [dir]"""
import json
import os
import shutil
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyfastx_amd as fx  # noqa: E402
from pyfastx_amd import synth  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
    where = sys.argv[2] if len(sys.argv) > 2 else "/dev/shm"
    dev = torch.device("cuda", 0)
    blob_t, cols = synth.fastq_generate(n, dev)
    nb = int(cols["n_bytes"])
    d = tempfile.mkdtemp(prefix="fxpmc", dir=where)
    try:
        path = os.path.join(d, "r.fq")
        with open(path, "wb") as f:
            for x in range(0, nb, 1 << 30):
                f.write(memoryview(blob_t[x:min(x + (1 << 30), nb)].cpu().numpy()))
        del blob_t
        torch.cuda.empty_cache()
        t = time.perf_counter()
        fq = fx.Fastq(path)
        el = time.perf_counter() - t
        print(json.dumps({"reads": n, "file_bytes": nb, "fxi_bytes": os.path.getsize(path + ".fxi"), "Fastq_ctor_s": round(el, 4),
                          "len": len(fq), "phases": getattr(fq, "build_phases", None),
                          "index_phases": {k: round(v, 4) for k, v in (getattr(fq, "index_phases", None) or {}).items()}}))
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
