#!/usr/bin/env python3
"""SURVEY 8e "Fetch" at scale on ONE MI355X: G logical byte-range shards of a (G x gbp) Gbp FASTA (built as
tools/shard_scale.py builds them), 1 M random 100 bp queries over the WHOLE stream (contig ~ length, 50 % '-' strand)
plus windows laid across every cut, answered through shard.ShardFetcher -- routed on the host to the shard that holds
the bytes, fetched by the ordinary kernel shard by shard, cross-cut queries put together from two pieces -- and compared
base for base with the generator's un-wrapped sequences.   usage: python tools/shard_fetch_scale.py [G] [gbp] [nq]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfastx_amd import _lib, shard, synth  # noqa: E402


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    gbp = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    nq = int(float(sys.argv[3])) if len(sys.argv) > 3 else 1_000_000
    dev = torch.device("cuda", 0)
    plans = [synth.fasta_plan(total_bp=int(gbp * 1e9), seed=20260612 + r, tag="p%d_" % r) for r in range(G)]
    sizes = [int(p["n_bytes"]) for p in plans]
    jobs, flats, heads = [], [None] * G, [None] * G
    for r in reversed(range(G)):
        piece, flat, fstart = synth.fasta_generate(plans[r], dev, keep_flat=True)
        flats[r] = (flat, fstart)
        heads[r] = piece[:shard.ShardedFasta.DELTA].clone()
        jobs.append(shard.ShardedFasta(piece, sizes[r], dev, r, G, logical={"sizes": sizes, "next_head": heads[r + 1] if r + 1 < G else None}))
        del piece
        torch.cuda.empty_cache()
    jobs.reverse()
    for j in jobs:
        j.build_begin(); j.sync()
    allS = torch.cat([j._mine for j in jobs])
    for j in jobs:
        j._all.copy_(allS)
    torch.cuda.synchronize()
    for j in jobs:
        j.build_end(); j.sync(); j.finish()
    cols = ("boff", "blen", "slen", "llen", "elen", "norm")
    rows = [j.local_rows() for j in jobs]
    table = {c: np.concatenate([np.asarray(t[c]) for t in rows]) for c in cols}
    nrec = [len(p["slen"]) for p in plans]
    assert int(sum(nrec)) == table["boff"].size
    first = np.concatenate([[0], np.cumsum(nrec)])
    bases, ends = [j.base for j in jobs], [j.base + j.n_bytes for j in jobs]
    assert bases[1:] == ends[:-1]
    # ---- the batch: per piece contig ~ length, then the windows across the cuts
    per = nq // G
    parts = [synth.fasta_queries(plans[r], per, 100, seed=12345 + r) for r in range(G)]
    ids = np.concatenate([p[0] + first[r] for r, p in enumerate(parts)])
    st = np.concatenate([p[1] for p in parts])
    sp = np.concatenate([p[2] for p in parts])
    neg = np.concatenate([p[3] for p in parts])
    piece_of = np.repeat(np.arange(G), per)
    local = np.concatenate([p[0] for p in parts])
    cross = []
    for r in range(1, G):
        i = int(first[r])                                     # first contig of piece r: the cut lies inside it
        at = bases[r] - int(table["boff"][i])
        ll = int(table["llen"][i])
        mid = (at // ll) * (ll - 1) + min(at % ll, ll - 2)
        for d in (50, 1, 99):
            cross.append((i, mid - d, mid - d + 100, (r + d) & 1, r, 0))
    c = np.array(cross, dtype=np.int64)
    ids, st, sp = np.concatenate([ids, c[:, 0]]), np.concatenate([st, c[:, 1]]), np.concatenate([sp, c[:, 2]])
    neg = np.concatenate([neg, c[:, 3].astype(np.uint8)])
    piece_of, local = np.concatenate([piece_of, c[:, 4]]), np.concatenate([local, c[:, 5]])
    fl = np.where(neg != 0, shard.F_REV | shard.F_COMP, 0).astype(np.uint8)
    f = shard.ShardFetcher({r: j.blob for r, j in enumerate(jobs)}, bases, ends, table)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        P = _lib.shard_route(ids, st, sp, f._cols, bases, ends, 0, fl)        # what fetch() begins with (fx_shard_route)
        t1 = time.perf_counter()
        qidx, buf, offs = f.fetch(ids, st, sp, flags_per_query=fl)
        t2 = time.perf_counter()
        times.append((t1 - t0, t2 - t1))
    n_cross = int((P["cnt"] > 1).sum())
    # ---- truth from the un-wrapped bases
    ok = qidx.size == ids.size and bool((np.diff(offs) == 100).all())
    inv = np.empty(ids.size, dtype=np.int64)
    inv[qidx] = np.arange(ids.size)
    got = torch.from_numpy(np.ascontiguousarray(buf).reshape(-1, 100)).to(dev)[torch.from_numpy(inv).to(dev)]
    for r in range(G):
        sel = np.nonzero(piece_of == r)[0]
        flat, fstart = flats[r]
        want = synth.expected_fetch(flat, fstart, local[sel], st[sel], 100, neg[sel], dev)
        ok &= bool((got[torch.from_numpy(sel).to(dev)] == want).all())
    route_s, fetch_s = min(t[0] for t in times), min(t[1] for t in times)
    print(json.dumps({"workload": "%d logical shards x %.2f Gbp (%.1f GB) on one MI355X, %d queries of 100 bp over the whole stream"
                                  % (G, gbp, sum(sizes) / 1e9, ids.size),
                      "answers_equal_generator": ok, "queries_crossing_a_cut": n_cross,
                      "routing_only_ms": round(route_s * 1e3, 2), "fetch_all_shards_ms": round(fetch_s * 1e3, 2),
                      "M_queries_per_s_host_to_host": round(ids.size / fetch_s / 1e6, 2)}))
    if not ok:
        raise SystemExit("PARITY FAILURE")


if __name__ == "__main__":
    main()
