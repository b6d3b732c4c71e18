#!/usr/bin/env python3
"""File -> HBM staging rate of fx_open_file on a plain file in the page cache, for the staging parameters in the
environment (FX_STAGE_THREADS, FX_STAGE_PIECE_MB).  usage: python tools/stage_probe.py [GB]"""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfastx_amd import _lib  # noqa: E402


def main():
    gb = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    n = int(gb * 1e9)
    d = tempfile.mkdtemp(prefix="fxstage")
    path = os.path.join(d, "x.fa")
    a = np.full(n, ord("A"), dtype=np.uint8)
    a[::61] = 10
    a[0] = ord(">")
    a.tofile(path)
    del a
    _lib.Blob.from_file(path).close()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        b = _lib.Blob.from_file(path)
        ts.append(time.perf_counter() - t0)
        b.close()
    os.unlink(path)
    print(json.dumps({"GB": gb, "threads": os.environ.get("FX_STAGE_THREADS", "default"), "piece_MB": os.environ.get("FX_STAGE_PIECE_MB", "8"),
                      "open_s_median": round(sorted(ts)[2], 4), "GBps": round(n / sorted(ts)[2] / 1e9, 1), "best_GBps": round(n / min(ts) / 1e9, 1)}))


if __name__ == "__main__":
    main()
