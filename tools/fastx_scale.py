#!/usr/bin/env python3
"""Fastx (the kseq walk, fx_kseq.hpp) at scale on one MI355X: a C3-shaped FASTQ stream (150-base reads) and an
hg38-shaped FASTA stream (60-column lines) generated on the device -- line table, walk and gather timed per kernel,
the records checked against the generator, the Python iteration rate of pyfastx_amd.Fastx over a file of the same
shape beside the compiled reference's (oracle/_ref) when it is there.
usage: python tools/fastx_scale.py [n_reads] [fasta_bytes]   (defaults 10 M reads = 3.5 GB, 3.1 GB)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyfastx_amd import _lib, synth  # noqa: E402
import pyfastx_amd  # noqa: E402


def leg(blob, label, expect_records):
    blob.kseq_scan()                                          # warm-up (allocations)
    blob.prof_enable(True); blob.prof_reset()
    t0 = time.perf_counter()
    n_rec, n_lines, seq_bytes, code = blob.kseq_scan()
    t1 = time.perf_counter()
    assert n_rec == expect_records and code == -1, (n_rec, code)
    prof = blob.prof_read()
    blob.prof_reset()
    # gather: batches of 64 k records / 64 MiB as Fastx.__iter__ takes them -- here the first 4 M records
    done, tg = 0, 0.0
    first = blob.kseq_records(0, min(n_rec, 65536))
    while done < min(n_rec, 1 << 22):
        k = min(65536, n_rec - done)
        recs = blob.kseq_records(done, k)
        ends = recs["seq_cum"] + recs["seq_len"] - recs["seq_cum"][0]
        k = max(1, int(np.searchsorted(ends, 64 << 20, side="right")))
        t2 = time.perf_counter()
        blob.kseq_fetch(done, k, int(ends[k - 1]), want_qual=bool(recs["flags"][0] & 1))
        tg += time.perf_counter() - t2
        done += k
    g = blob.prof_read()
    out = {"label": label, "bytes": blob.size, "lines": n_lines, "records": n_rec, "seq_bytes": seq_bytes,
           "scan_wall_ms": round((t1 - t0) * 1e3, 2),
           "kernels_ms": {k: round(v[0], 3) for k, v in prof.items()},
           "gather_records": done, "gather_wall_ms": round(tg * 1e3, 2), "gather_kernel_ms": round(g.get("k_kq_gather", (0, 0))[0], 3)}
    return out, first


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    fa_bytes = int(float(sys.argv[2])) if len(sys.argv) > 2 else 3_100_000_000
    dev = torch.device("cuda", 0)
    res = []
    blob_t, cols = synth.fastq_generate(n, dev)
    b = _lib.Blob.from_device(blob_t.data_ptr(), cols["n_bytes"], device=0, keepalive=blob_t)
    r, first = leg(b, "fastq 150-base reads", n)
    assert (first["hdr_off"] == cols["name_off"][:first.size]).all() and (first["seq_len"] == 150).all() and (first["flags"] == 1).all()
    # the END of the stream as well (offsets beyond 4 GiB when n is large): record table and gathered strings against the blob
    k = min(n, 65536)
    lastr = b.kseq_records(n - k, k)
    assert (lastr["hdr_off"] == cols["name_off"][n - k:]).all() and (lastr["seq_cum"] == 150 * np.arange(n - k, n)).all()
    sq, ql = b.kseq_fetch(n - k, k, 150 * k)
    so, qo, rec = int(cols["soff"][0]), int(cols["qoff"][0]), cols["rec"]
    v = blob_t[:cols["n_bytes"]].view(n, rec)[n - k:]
    assert bytes(sq) == v[:, so:so + 150].contiguous().cpu().numpy().tobytes() and bytes(ql) == v[:, qo:qo + 150].contiguous().cpu().numpy().tobytes()
    r["tail_records_and_strings_equal_the_blob"] = True
    res.append(r)
    b.close(); del blob_t
    torch.cuda.empty_cache()
    # FASTA: 24 records of 60-column lines
    per = fa_bytes // 24
    lines = per // 61
    row = torch.from_numpy(np.frombuffer(b"ACGT" * 15 + b"\n", dtype=np.uint8).copy()).to(dev)
    body = row.repeat(lines)
    parts = []
    for i in range(24):
        parts += [torch.from_numpy(np.frombuffer(b">chr%d  test\n" % i, dtype=np.uint8).copy()).to(dev), body]
    fa = torch.cat(parts)
    b = _lib.Blob.from_device(fa.data_ptr(), fa.numel(), device=0, keepalive=fa)
    r, first = leg(b, "fasta 60-column lines", 24)
    assert (first["seq_len"] == lines * 60).all() and (first["s_n"][:-1] == lines).all()
    res.append(r)
    b.close(); del fa, body
    torch.cuda.empty_cache()
    # the Python iteration over a FILE, beside the reference
    m = min(n, 2_000_000)
    path = "/tmp/fastx_scale.fq"
    rng = np.random.default_rng(1)
    with open(path, "wb") as f:
        for a in range(0, m, 100000):
            k = min(100000, m - a)
            s = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), (k, 150))
            q = rng.integers(33, 74, (k, 150), dtype=np.uint8)
            f.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (a + i, s[i].tobytes(), q[i].tobytes()) for i in range(k)))
    t0 = time.perf_counter()
    cnt = sum(1 for _ in pyfastx_amd.Fastx(path))
    t1 = time.perf_counter()
    it = {"file_reads": m, "ours_s": round(t1 - t0, 3)}
    assert cnt == m
    # where the time goes: the same loop as Fastx.__iter__, section by section
    from pyfastx_amd import _fxobj
    tm = {"open": 0.0, "scan": 0.0, "records": 0.0, "gather": 0.0, "headers": 0.0, "tuples": 0.0}
    c = time.perf_counter
    t = c(); blob = _lib.Blob.from_file(path); tm["open"] = c() - t
    t = c(); n_rec = blob.kseq_scan()[0]; tm["scan"] = c() - t
    state = [False, None]
    for a in range(0, n_rec, 65536):
        t = c(); recs = blob.kseq_records(a, min(65536, n_rec - a)); tm["records"] += c() - t
        nb = int(recs["seq_cum"][-1] + recs["seq_len"][-1] - recs["seq_cum"][0])
        t = c(); seq, qual = blob.kseq_fetch(a, recs.size, nb); tm["gather"] += c() - t
        hl = recs["hdr_len"].astype(np.int64)
        t = c(); hdr, ho, _ = blob.fetch_ranges(recs["hdr_off"], hl, hl, flags=_lib.FX_RAW); tm["headers"] += c() - t
        t = c(); out = _fxobj.fastx_batch(hdr, ho, seq, qual, recs, True, False, state); tm["tuples"] += c() - t
        del out
    blob.close()
    it["sections_s"] = {k: round(v, 3) for k, v in tm.items()}
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    try:
        import pyfastx
        t0 = time.perf_counter()
        cnt = sum(1 for _ in pyfastx.Fastx(path))
        it["reference_s"] = round(time.perf_counter() - t0, 3)
    except ImportError:
        pass
    os.remove(path)
    res.append(it)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
