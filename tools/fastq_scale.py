#!/usr/bin/env python3
"""C3-shaped FASTQ at scale on one MI355X: index build, composition, 1 M read fetches
(seq + qual + int8 quali), verified against the generator's analytic truth / torch.
usage: python tools/fastq_scale.py [n_reads]   (default 10 M reads = 3.5 GB)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfastx_amd import _lib, synth  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    dev = torch.device("cuda", 0)
    blob_t, cols = synth.fastq_generate(n, dev)
    nb = cols["n_bytes"]
    b = _lib.Blob.from_device(blob_t.data_ptr(), nb, device=0, keepalive=blob_t)
    b.fastq_build(); b.fastq_comp()                       # warm-up (allocations)
    b.prof_enable(True); b.prof_reset()
    R = 5
    t0 = time.perf_counter()
    for _ in range(R):
        s = b.fastq_build()
    t1 = time.perf_counter()
    for _ in range(R):
        base, meta = b.fastq_comp()
    t2 = time.perf_counter()
    # index AND composition in one read of the stream (fx_fastq_build_comp), against the two passes above
    b.fastq_build(comp=True); b.fastq_comp()
    t2b = time.perf_counter()
    for _ in range(R):
        b.fastq_build(comp=True)
        base1, meta1 = b.fastq_comp()
    t2c = time.perf_counter()
    one_read = {"index_and_composition_one_read_ms": round((t2c - t2b) / R * 1e3, 3),
                "equal_two_pass": bool((base1 == base).all() and (meta1 == meta).all())}
    assert one_read["equal_two_pass"], (base1, base, meta1, meta)
    s = b.fastq_build()
    assert (s.n_reads, s.size) == (n, n * 150)
    t = b.fastq_table(n)
    for k in ("name_off", "name_len", "dlen", "rlen", "soff", "qoff"):
        assert (t[k] == cols[k]).all(), k
    rec = cols["rec"]
    v = blob_t[:nb].view(n, rec)
    so, qo = int(cols["soff"][0]), int(cols["qoff"][0])
    seqs = v[:, so:so + 150]
    want = [int((seqs == c).sum()) for c in b"ACGT"]
    want.append(n * 150 - sum(want))
    assert base.tolist() == want, (base.tolist(), want)
    q = v[:, qo:qo + 150]
    assert meta.tolist() == [150, 150, int(q.min()), int(q.max()), 33], meta.tolist()
    # 1 M random read fetches on device
    nq = 1_000_000
    rng = np.random.default_rng(99)
    ids = torch.from_numpy(rng.integers(0, n, nq)).to(dev)
    off = torch.arange(nq, device=dev, dtype=torch.int64) * 150
    o_seq = torch.zeros(nq * 150, dtype=torch.uint8, device=dev); o_q = torch.zeros_like(o_seq)
    o_qi = torch.zeros(nq * 150, dtype=torch.int8, device=dev)
    L = _lib.lib()
    def fetch():
        _lib.check(L.fx_fastq_fetch(b._h, _lib.FX_DEVICE, nq, ids.data_ptr(), 33, 0, o_seq.data_ptr(), o_q.data_ptr(),
                                    o_qi.data_ptr(), off.data_ptr()))
        b.sync()
    fetch()
    t3 = time.perf_counter()
    for _ in range(R):
        fetch()
    t4 = time.perf_counter()
    assert bool((o_seq.view(nq, 150) == seqs[ids]).all()) and bool((o_q.view(nq, 150) == q[ids]).all())
    assert bool((o_qi.view(nq, 150) == (q[ids].to(torch.int16) - 33).to(torch.int8)).all())
    # name -> id table in HBM: build, then 1 M names resolved in one call (host arrays, H2D / D2H included)
    t5 = time.perf_counter()
    b.names_build(1)
    t6 = time.perf_counter()
    hid = ids.cpu().numpy()
    tile_ = (hid // 50_000) % 10_000
    x_ = (hid * 7919) % 100_000
    qn = [b"SYN:1:FC:1:%04d:%05d:%09d" % (int(a), int(c), int(i)) for a, c, i in zip(tile_[:200_000], x_[:200_000], hid[:200_000])]
    offs_n = np.zeros(len(qn) + 1, dtype=np.int64)
    np.cumsum([len(x) for x in qn], out=offs_n[1:])
    packed = np.frombuffer(b"".join(qn) + b"\0" * 16, dtype=np.uint8)
    out_ids = np.empty(len(qn), dtype=np.int64)
    t7 = time.perf_counter()
    _lib.check(L.fx_names_lookup(b._h, _lib.FX_HOST, len(qn), packed.ctypes.data, offs_n.ctypes.data, out_ids.ctypes.data))
    t8 = time.perf_counter()
    assert (out_ids == hid[:len(qn)]).all()
    prof = {k: round(v[0] / v[1], 4) for k, v in b.prof_read().items()}
    # the .fxi of these reads: names off the GPU (one gather), their sorted order (GPU radix sort), both b-trees as pages
    from pyfastx_amd import fxi
    import sqlite3
    fx_t = {}
    tA = time.perf_counter()
    ln = t["name_len"].astype(np.int64)
    packed_n, offs_all = b.names_pack(1, n, guess=int(ln.sum()))
    tB = time.perf_counter()
    order, ndup = b.names_sort(1, n)
    tC = time.perf_counter()
    assert ndup == 0
    w = int(ln[0])
    assert (ln == w).all()
    fixed = packed_n[:n * w].view("S%d" % w)
    srt = fixed[order]
    assert (srt[:-1] < srt[1:]).all()                    # strictly increasing in memcmp order
    out_dir = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    path = os.path.join(out_dir, "fastq_scale.fxi")
    if os.path.exists(path):
        os.remove(path)
    tD = time.perf_counter()
    db = fxi.write_fastq_bulk(path, packed_n[:int(offs_all[-1])], offs_all, t, s.size, order=order)
    db.close()
    tE = time.perf_counter()
    db = sqlite3.connect(path)
    chk = db.execute("PRAGMA integrity_check").fetchall() if n <= 30_000_000 else db.execute("PRAGMA quick_check").fetchall()
    tF = time.perf_counter()
    assert chk == [("ok",)], chk[:3]
    for i in rng.integers(0, n, 200).tolist():
        nm = fixed[i].decode()
        assert db.execute("SELECT ID, soff FROM read WHERE name=?", (nm,)).fetchone() == (i + 1, int(t["soff"][i]))
    assert db.execute("SELECT count(*) FROM read").fetchone()[0] == n
    db.close()
    fx_t = {"names_gather_ms": round((tB - tA) * 1e3, 1), "names_sort_ms": round((tC - tB) * 1e3, 1),
            "write_pages_s": round(tE - tD, 2), "fxi_MB": round(os.path.getsize(path) / 1e6, 1),
            "rows_per_s_M": round(n / (tE - tA) / 1e6, 2), "sqlite_check_s": round(tF - tE, 1), "dir": out_dir}
    os.remove(path)
    print(json.dumps({"workload": "synthetic FASTQ %d x 150 bp (%.2f GB)" % (n, nb / 1e9), "index_build_ms": round((t1 - t0) / R * 1e3, 3),
                      "index_build_GBps": round(nb / ((t1 - t0) / R) / 1e9, 1), "composition_ms": round((t2 - t1) / R * 1e3, 3), "full_index": one_read,
                      "fetch_1M_reads_ms": round((t4 - t3) / R * 1e3, 3), "M_reads_per_s": round(nq / ((t4 - t3) / R) / 1e6, 1),
                      "names_table_build_ms": round((t6 - t5) * 1e3, 3), "names_lookup_200k_host_arrays_ms": round((t8 - t7) * 1e3, 3),
                      "fxi_bulk": fx_t, "kernels_ms_avg": prof, "verified": True}))


if __name__ == "__main__":
    main()
