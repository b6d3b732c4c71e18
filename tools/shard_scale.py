#!/usr/bin/env python3
"""C5 shape on ONE MI355X: G logical shards of a (G x gbp) Gbp FASTA, built one after the other with exactly the
calls a rank of `bench.py --gpus G` makes (ShardedFasta.build_begin / build_end, the device-resident summary and
stitch kernels); the all-gather is a copy.  Every shard's rows are checked against the generator's analytic
ground truth, the stitched record that crosses each cut included.   usage: python tools/shard_scale.py [G] [gbp]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfastx_amd import shard, synth  # noqa: E402


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    gbp = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
    dev = torch.device("cuda", 0)
    plans = [synth.fasta_plan(total_bp=int(gbp * 1e9), seed=20260612 + r, tag="p%d_" % r) for r in range(G)]
    sizes = [int(p["n_bytes"]) for p in plans]
    jobs = []
    nxt_piece = None
    # build the shards from the last to the first so that only two pieces are alive at a time
    heads = [None] * G
    pieces = [None] * G
    for r in reversed(range(G)):
        piece, _, _ = synth.fasta_generate(plans[r], dev, keep_flat=False)
        heads[r] = piece[:shard.ShardedFasta.DELTA].clone()
        job = shard.ShardedFasta(piece, sizes[r], dev, r, G, logical={"sizes": sizes, "next_head": heads[r + 1] if r + 1 < G else None})
        del piece
        torch.cuda.empty_cache()
        jobs.append(job)
    jobs.reverse()
    prof = []
    for j in jobs[:2]:
        j.blob.prof_enable(1); j.blob.prof_reset()
        for _ in range(3):
            j.build_begin(); j.sync()
        prof.append({k: round(v[0] / v[1], 4) for k, v in j.blob.prof_read().items()})
        j.blob.prof_enable(0)
    split = []
    for j in jobs[:2]:
        for _ in range(3):
            t0 = time.perf_counter(); j.blob.fasta_build(); t1 = time.perf_counter()
            j.blob.shard_summary_dev(j._mine.data_ptr()); j.sync(); t2 = time.perf_counter()
        split.append([round((t1 - t0) * 1e3, 3), round((t2 - t1) * 1e3, 3)])
    times = []
    for rep in range(3):
        t = []
        for j in jobs:
            t0 = time.perf_counter()
            j.build_begin()
            j.sync()
            t.append(time.perf_counter() - t0)
        allS = torch.cat([j._mine for j in jobs])
        for j in jobs:
            j._all.copy_(allS)                             # stands in for all_gather_into_tensor
        torch.cuda.synchronize()
        t2 = []
        for j in jobs:
            t0 = time.perf_counter()
            j.build_end()
            j.sync()
            t2.append(time.perf_counter() - t0)
        times.append((t, t2))
    ok = True
    for r, j in enumerate(jobs):
        j.finish()                                         # the count of this shard's records (local_rows allocates by it)
        rows = j.local_rows()
        ok &= bool(j.check_against_plan(plans[r], rows, plans[r + 1] if r + 1 < G else None))
    bt, st = times[-1]
    print(json.dumps({"workload": "C5 shape: %d logical shards x %.1f Gbp (%.1f GB) on one MI355X" % (G, gbp, sum(sizes) / 1e9),
                      "rows_equal_plan_all_shards": ok, "build_begin_ms_per_shard": [round(x * 1e3, 3) for x in bt],
                      "stitch_ms_per_shard": [round(x * 1e3, 3) for x in st],
                      "sum_build_ms": round(sum(bt) * 1e3, 3), "kernels_ms_rank0_rank1": prof, "build_vs_summary_ms_rank0_rank1": split}))
    if not ok:
        raise SystemExit("PARITY FAILURE")


if __name__ == "__main__":
    main()
