#!/usr/bin/env python3
"""A FASTQ file over two byte-range shards (shard.ShardedFastq, two logical ranks on one GPU): open + scan + rows per rank,
then ONE .fxi from raw arrays passed through files -- rows and seconds.   usage: python tools/fastq_shard_merge.py [reads]"""
import json
import os
import shutil
import sqlite3
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfastx_amd import _lib, shard, synth  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
    dev = torch.device("cuda", 0)
    blob_t, cols = synth.fastq_generate(n, dev)
    nb = int(cols["n_bytes"])
    d = tempfile.mkdtemp(prefix="fxq", dir="/dev/shm" if shutil.disk_usage("/dev/shm").free > 3 * nb else None)
    path = os.path.join(d, "reads.fq")
    blob_t[:nb].cpu().numpy().tofile(path)
    del blob_t
    torch.cuda.empty_cache()
    world = 2
    size, _ = _lib.stream_size(path)
    t0 = time.perf_counter()
    cores = []
    for r in range(world):
        lo, hi = size * r // world, size * (r + 1) // world
        b = _lib.Blob.from_file_range(path, lo, hi - lo, 0)
        cores.append(b.fastq_scan())
        b.close()
    table = np.array(cores, dtype=np.int64)
    t1 = time.perf_counter()
    ranks = [shard.ShardedFastq(path, r, world, gather=lambda mine, t=table: t) for r in range(world)]
    t2 = time.perf_counter()
    scratch = os.path.join(d, "scratch")
    os.mkdir(scratch)
    t_parts = []
    for r in ranks[1:] + ranks[:1]:
        ts = time.perf_counter()
        rows = r.write_index(path + ".fxi", scratch, barrier=lambda: None)
        t_parts.append(time.perf_counter() - ts)
    t3 = time.perf_counter()
    db = sqlite3.connect(path + ".fxi")
    ok = db.execute("PRAGMA integrity_check").fetchone()[0] == "ok"
    cnt = db.execute("SELECT counts, size FROM stat").fetchone()
    first = db.execute("SELECT name, dlen, rlen, soff, qoff FROM read WHERE ID=1").fetchone()
    last = db.execute("SELECT dlen, rlen, soff, qoff FROM read WHERE ID=?", (n,)).fetchone()
    db.close()
    good = ok and cnt == (n, n * 150) and last == (int(cols["dlen"][-1]), 150, int(cols["soff"][-1]), int(cols["qoff"][-1]))
    print(json.dumps({"reads": n, "file_GB": round(nb / 1e9, 2), "shards": world, "open_scan_build_per_rank_s": round((t2 - t1) / world, 3),
                      "rank_parts_to_files_s": round(sum(t_parts[:-1]), 3), "rank0_merge_sort_write_s": round(t_parts[-1], 3),
                      "merged_fxi_s": round(t3 - t2, 3), "rows_per_s_M": round(n / (t3 - t2) / 1e6, 2), "fxi_GB": round(os.path.getsize(path + ".fxi") / 1e9, 2),
                      "integrity_and_rows_ok": bool(good), "first_row": first[:1] + first[1:]}))
    shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
