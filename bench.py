#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: FASTA index build + 1 M random
100-bp sub-sequence fetches on a synthetic 3 Gbp hg38-shaped plain FASTA per
GPU (configs[1]); at N>1 the N pieces form ONE stream sharded by byte range
across the ranks (configs[4] shape) and stitched with one RCCL all-gather.

A "step" = fx_fasta_build (one-read granule scan -> prefixes -> record table) over
the shard resident in HBM  +  fx_fasta_fetch of 1 M (id,start,stop,strand)
queries into a device buffer, enqueued back to back (fx_fasta_build_begin ... fetch ...
fx_fasta_build_end: one host synchronisation per step).  Inputs are resident in HBM before the timed
region.  One JSON line on rank 0 (contract in the task statement), plus
`roofline` for the dominant kernel (k_span_scan, HIP events on the library's own
stream) and `cpu_baseline` (the real reference built from /root/reference ->
oracle/_ref when loadable, else the C port) at N=1.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gbp", type=float, default=3.0, help="Gbp of FASTA per GPU (3.0 = BASELINE config)")
    ap.add_argument("--queries", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    return ap.parse_args()


def cpu_baseline(blob_t, nbytes, plan, q, verify_rows):
    """Time the CPU path on this host: reference pyfastx (oracle/_ref) if it
    loads, else the C port.  Sample = the FULL single-GPU workload."""
    import tempfile
    ids, st, sp, strand = q
    tmpdir = tempfile.mkdtemp(prefix="fxbench")
    path = os.path.join(tmpdir, "c2.fa")
    host = blob_t[:nbytes].cpu().numpy()
    out = {"cores": 1}
    try:
        host.tofile(path)
        sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
        import pyfastx                                    # the reference itself
        t0 = time.perf_counter()
        fa = pyfastx.Fasta(path)                          # benchmark/pyfastx_fasta_build_index.py idiom
        t1 = time.perf_counter()
        names = plan["names"]
        nq = len(ids)
        for j in range(nq):                               # benchmark/pyfastx_fasta_extract_subsequences.py idiom
            s = fa[names[ids[j]]][int(st[j]):int(sp[j])]
            _ = s.antisense if strand[j] else s.seq
        t2 = time.perf_counter()
        # full-size parity of the index rows against the real reference
        import sqlite3
        db = sqlite3.connect(path + ".fxi")
        rows = db.execute("SELECT chrom,boff,blen,slen,llen,elen,norm,dlen FROM seq ORDER BY ID").fetchall()
        db.close()
        ok = (len(rows) == len(verify_rows)) and all(tuple(a) == tuple(b) for a, b in zip(rows, verify_rows))
        out.update(kind="reference", index_s=t1 - t0, fetch_s=t2 - t1, rows_equal_gpu=bool(ok),
                   sample="full workload: pyfastx.Fasta() on the %.2f GB file + %d fa[name][s:e].seq/.antisense"
                          % (nbytes / 1e9, nq))
        del fa
    except Exception as e:                               # reference .so not loadable here: use the C port
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import fxoracle
        t0 = time.perf_counter()
        recs, tot = fxoracle.fasta_index(host)
        t1 = time.perf_counter()
        nq = min(len(ids), 200_000)
        r = recs[ids[:nq]]
        bpl = r["llen"] - r["elen"]
        off = r["boff"] + st[:nq] + r["elen"] * (st[:nq] // bpl)
        bl = (sp[:nq] - st[:nq]) + (sp[:nq] // bpl - st[:nq] // bpl) * r["elen"]
        fxoracle.fetch_batch(host, off, bl, sp[:nq] - st[:nq], np.where(strand[:nq] > 0, 6, 0))
        t2 = time.perf_counter()
        scale = len(ids) / nq
        out.update(kind="port", index_s=t1 - t0, fetch_s=(t2 - t1) * scale,
                   sample="C port of the scan on the full %.2f GB + %d fetches (scaled to %d); reference unavailable: %s"
                          % (nbytes / 1e9, nq, len(ids), str(e)[:80]))
    finally:
        for f in (path, path + ".fxi"):
            try:
                os.unlink(f)
            except OSError:
                pass
        try:
            os.rmdir(tmpdir)
        except OSError:
            pass
    return out


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from pyfastx_amd import _lib, synth, shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % a.gpus)
    backend = os.environ.get("FX_BENCH_BACKEND", "nccl")    # "nccl" IS RCCL on ROCm; "gloo" only for the 1-GPU plumbing test
    local = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # ---------------- workload: piece `rank` of the concatenated stream, resident in HBM
    total_bp = int(a.gbp * 1e9)
    plan = synth.fasta_plan(total_bp=total_bp, seed=20260612 + rank, tag=("p%d_" % rank) if world > 1 else "")
    blob, flat, flat_start = synth.fasta_generate(plan, dev, keep_flat=not a.no_verify)
    # world > 1: the first contig of every piece crosses a shard cut, so queries (answered from
    # the local shard only -- no collective on the fetch path) use the other contigs
    q = synth.fasta_queries(plan, n=a.queries, seed=12345 + rank, skip_first=world > 1)
    ids, st, sp, strand = q
    qlen = int(sp[0] - st[0])
    job = shard.ShardedFasta(blob, int(plan["n_bytes"]), dev, rank, world)     # moves the shard cut off the piece boundary
    id_shift = 1 if (world > 1 and rank > 0) else 0       # local row index of piece contig i is i - id_shift
    d_ids = torch.from_numpy(ids - id_shift).to(dev); d_st = torch.from_numpy(st).to(dev); d_sp = torch.from_numpy(sp).to(dev)
    d_fl = torch.from_numpy((strand * 6).astype(np.uint8)).to(dev)               # '-' = reverse|complement
    d_off = torch.arange(a.queries, device=dev, dtype=torch.int64) * qlen
    d_out = torch.zeros(a.queries * qlen, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(a.queries, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    def step():
        job.build_async()                                 # scan + tables (+ all-gather & stitch when world > 1), enqueued
        job.fetch_local(a.queries, d_ids, d_st, d_sp, d_fl, d_out, d_off, d_len)      # reads the record count on the device
        job.finish()                                      # the step's one host synchronisation: totals of the build
        job.sync()

    def build_only():
        ts = time.perf_counter()
        job.build()
        return time.perf_counter() - ts

    for _ in range(a.warmup):
        step()
    job.blob.prof_enable(2)                               # events around the dominant kernel only (k_span_scan)
    job.blob.prof_reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    # the index build on its own (its own synchronisation), outside the timed region: the first half of the metric
    t_index = sum(build_only() for _ in range(a.steps))
    elapsed = torch.tensor([t1 - t0, t_index], dtype=torch.float64, device=job.comm_dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    el, ti = float(elapsed[0]), float(elapsed[1])
    prof = job.blob.prof_read()
    # per-kernel table from a separate, untimed pass (timing every kernel costs ~20 events per step)
    job.blob.prof_enable(1)
    job.blob.prof_reset()
    for _ in range(5):
        step()
    prof_all = job.blob.prof_read()
    job.blob.prof_enable(0)

    # ---------------- parity at full size (size-independent properties + analytic truth)
    verified = None
    if not a.no_verify:
        rows = job.local_rows()                           # records that START in this shard, as host arrays
        nxt = synth.fasta_plan(total_bp=total_bp, seed=20260612 + rank + 1, tag="p%d_" % (rank + 1)) if rank < world - 1 else None
        verified = bool(job.check_against_plan(plan, rows, nxt))
        exp = synth.expected_fetch(flat, flat_start, ids, st, qlen, strand, dev)
        verified = verified and bool((d_out.view(a.queries, qlen) == exp).all()) and bool((d_len == qlen).all())
        if not verified:
            raise SystemExit("PARITY FAILURE at full size: refusing to report a speed-up")
    comp_ms = None
    if world == 1 and not a.no_verify:
        # per-record composition (fasta.c:901-950) at full size vs torch.bincount of the un-wrapped bases
        # (extra information, outside the timed region: full_index is lazy in the reference too)
        nrec = len(plan["slen"])
        d_comp = torch.zeros((nrec, 128), dtype=torch.int64, device=dev)
        job.blob.fasta_comp_dev(d_comp.data_ptr())
        job.sync()
        tc = time.perf_counter()
        for _ in range(3):
            job.blob.fasta_comp_dev(d_comp.data_ptr())
        job.sync()
        comp_ms = (time.perf_counter() - tc) / 3 * 1e3
        okc = True
        for i in range(nrec):
            L = int(plan["slen"][i])
            if L:
                seg = flat[int(flat_start[i]):int(flat_start[i]) + L]
                okc &= bool((torch.bincount(seg.long(), minlength=128)[:128] == d_comp[i]).all())
        if not okc:
            raise SystemExit("PARITY FAILURE (composition) at full size")

    if world > 1 and not a.no_verify:
        # composition across the cuts (shard.ShardedFasta.composition: local counting + two small all-gathers):
        # the contigs that lie wholly in this rank's piece row by row, and -- for the contig that crosses each cut,
        # whose bases sit on two ranks -- the sum of all rows of all ranks against the sum of all bincounts
        tc = time.perf_counter()
        comp = job.composition()
        comp_ms = (time.perf_counter() - tc) * 1e3
        first = 1 if rank > 0 else 0
        okc = True
        for i in range(first, len(plan["slen"])):
            L = int(plan["slen"][i])
            seg = flat[int(flat_start[i]):int(flat_start[i]) + L]
            okc &= bool((torch.bincount(seg.long(), minlength=128)[:128].cpu() == torch.from_numpy(comp[i - first])).all())
        tot = torch.from_numpy(comp.sum(axis=0)).to(dev)
        want = torch.zeros(128, dtype=torch.int64, device=dev)
        for i in range(len(plan["slen"])):
            L = int(plan["slen"][i])
            want += torch.bincount(flat[int(flat_start[i]):int(flat_start[i]) + L].long(), minlength=128)[:128]
        both = torch.stack([tot, want]).to(job.comm_dev)
        dist.all_reduce(both, op=dist.ReduceOp.SUM)
        okc &= bool((both[0] == both[1]).all())
        if not okc:
            raise SystemExit("PARITY FAILURE (composition across shards) at full size")

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    ms = el / a.steps * 1e3
    shard_bytes = job.n_bytes
    fetch_ms = prof_all.get("k_fetch", (0.0, 1))[0] / max(prof_all.get("k_fetch", (0.0, 1))[1], 1)   # kernel time of one batch
    scan_ms, scan_n = prof.get("k_span_scan", (0.0, 0))
    scan_avg = scan_ms / max(scan_n, 1)
    achieved = shard_bytes / (scan_avg * 1e-3) / 1e9 if scan_avg > 0 else 0.0
    line = {
        "metric": "FASTA index build + 1M random 100bp subseq fetches, 3 Gbp plain FASTA per GPU (throughput of the whole step)",
        "value": round(world * a.gbp / (el / a.steps), 3), "unit": "Gbp/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "configs[1]: synthetic %.1f Gbp hg38-shaped plain FASTA per GPU (200 contigs, 60-col LF, soft-masked, N runs), "
                               "index build + %d random %d bp intervals (50%% '-' strand)" % (a.gbp, a.queries, qlen),
                   "file_bytes_per_gpu": int(plan["n_bytes"]), "parallelism": "byte-range shards x%d, 1 all-gather" % world},
        "index_build_s": round(ti / a.steps, 6),
        "fetch_M_per_s": round(world * a.queries / max(fetch_ms * 1e-3, 1e-9) / 1e6, 2),
        "parity_verified_full_size": verified,
        "composition_pass_ms": None if comp_ms is None else round(comp_ms, 3),
        "kernels_ms_avg": {k: round(v[0] / v[1], 4) for k, v in prof_all.items()},
        "roofline": {"kernel": "fx::k_span_scan<0>", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": shard.pmc_traffic(ROOT, shard_bytes),
                     "algorithmic_bytes_per_launch": int(shard_bytes), "avg_launch_ms": round(scan_avg, 4)},
    }
    if world == 1 and not a.no_cpu_baseline:
        rows = job.local_rows()
        names = [n for n in plan["names"]]
        vrows = [(names[i], int(rows["boff"][i]), int(rows["blen"][i]), int(rows["slen"][i]), int(rows["llen"][i]),
                  int(rows["elen"][i]), int(rows["norm"][i]), int(rows["dlen"][i])) for i in range(len(names))]
        cb = cpu_baseline(blob, int(plan["n_bytes"]), plan, q, vrows)
        cpu_s = cb["index_s"] + cb["fetch_s"]
        cb["value"] = round(a.gbp / cpu_s, 4)
        cb["unit"] = "Gbp/s"
        cb["index_s"] = round(cb["index_s"], 3); cb["fetch_s"] = round(cb["fetch_s"], 3)
        cb["cpu"] = "%d logical cores on the box, 1 used (reference is single-threaded)" % (os.cpu_count() or 0)
        line["cpu_baseline"] = cb
        line["speedup_vs_cpu"] = round(line["value"] / cb["value"], 1)
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
