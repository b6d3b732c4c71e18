#!/usr/bin/env python3
"""The name sort alone, C3 shape (fx_fxi_dev_sort over N synthetic reads resident in HBM): wall time per sort and the order
against the generator's truth (names are `tile:x:i` with tile and i rising: the sorted order is by (tile, x, i)).
usage: python tools/sort_probe.py [n_reads]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfastx_amd import _lib, synth  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
    dev = torch.device("cuda", 0)
    blob_t, cols = synth.fastq_generate(n, dev)
    b = _lib.Blob.from_device(blob_t.data_ptr(), cols["n_bytes"], device=0, keepalive=blob_t)
    b.fastq_build()
    b.fxi_dev_sort(1)
    times = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ndup = b.fxi_dev_sort(1)
        times.append((time.perf_counter() - t0) * 1e3)
    out = {"reads": n, "sort_ms": [round(x, 2) for x in times], "sort_ms_min": round(min(times), 2), "n_dup": ndup}
    if n <= 30_000_000:                                    # the order against numpy's (lexsort over the three numbers of a name)
        order, nd = b.names_sort(1, n)
        i = np.arange(n, dtype=np.int64)
        want = np.lexsort((i, (i * 7919) % 100_000, (i // 50_000) % 10_000))
        out["order_equal_truth"] = bool((order == want).all()) and nd == 0
    print(json.dumps(out))
    if out.get("order_equal_truth") is False:
        raise SystemExit("PARITY FAILURE")


if __name__ == "__main__":
    main()
