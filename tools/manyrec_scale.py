#!/usr/bin/env python3
"""A FASTA of MANY small records (transcriptome / metagenome shape) at scale on one MI355X: every 4 KiB granule holds
several header lines, so the build runs through k_hdr_rec / k_gran_exact everywhere and the composition through the
segment path of k_fasta_comp.  Index rows, composition and by-name fetches are verified against the generator's
analytic truth; the .fxi (seq table + chromidx) is written as b-tree pages and checked by SQLite.
usage: python tools/manyrec_scale.py [n_records] [bases_per_record]   (default 5 M x 300)"""
import json
import os
import sqlite3
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfastx_amd import _lib, fxi  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5_000_000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    W = 60
    dev = torch.device("cuda", 0)
    head = b">tx0000000 gene=G len=%d\n" % L
    hl = len(head)
    nlines = (L + W - 1) // W
    rec = hl + L + nlines
    t = torch.empty((n, rec), dtype=torch.uint8, device=dev)
    t[:, :hl] = torch.frombuffer(bytearray(head), dtype=torch.uint8).to(dev)
    idx = torch.arange(n, device=dev, dtype=torch.int64)
    for k in range(7):
        t[:, 9 - k] = (48 + (idx // (10 ** k)) % 10).to(torch.uint8)
    g = torch.Generator(device=dev); g.manual_seed(3)
    protein = len(sys.argv) > 3 and sys.argv[3] == "protein"
    letters = b"ACDEFGHIKLMNPQRSTVWY" if protein else b"ACGTacgtNn"
    lut = torch.tensor(list((letters * 2)[:16]) if protein else list(letters), dtype=torch.uint8, device=dev)
    body = t[:, hl:].view(n, nlines, -1) if (L % W == 0) else None
    seq_cols = []                                            # column indices of the bases inside a record
    for j in range(L):
        seq_cols.append(hl + j + j // W)
    seq_cols = torch.tensor(seq_cols, device=dev)
    step = 1 << 18
    for a in range(0, n, step):
        b = min(n, a + step)
        r = torch.randint(0, 1000, (b - a, L), device=dev, generator=g)
        bases = lut[(r & 15).long()] if protein else lut[(r & 7).long()]
        if not protein:
            bases[r >= 995] = lut[8 + (r[r >= 995] & 1)]
        t[a:b, seq_cols] = bases
    nl_cols = torch.tensor([hl + min((k + 1) * W, L) + k for k in range(nlines)], device=dev)
    t[:, nl_cols] = 10
    nb = n * rec
    blob_t = torch.zeros(nb + 131072, dtype=torch.uint8, device=dev)
    blob_t[:nb] = t.view(-1)
    b = _lib.Blob.from_device(blob_t.data_ptr(), nb, device=0, keepalive=blob_t)
    b.fasta_build()                                          # warm-up (allocations)
    b.prof_enable(True); b.prof_reset()
    t0 = time.perf_counter()
    for _ in range(3):
        s = b.fasta_build()
    t1 = time.perf_counter()
    assert (s.n_seq, s.seq_len) == (n, n * L), (s.n_seq, s.seq_len)
    rows = b.fasta_table(n)
    i = np.arange(n, dtype=np.int64)
    want = {"hoff": i * rec, "boff": i * rec + hl, "blen": np.full(n, L + nlines), "slen": np.full(n, L),
            "llen": np.full(n, min(W, L) + 1), "elen": np.full(n, 1), "norm": np.full(n, 1), "dlen": np.full(n, hl - 2),
            "name_len": np.full(n, 9)}
    for k, v in want.items():
        assert (rows[k] == v).all(), k
    t2 = time.perf_counter()
    comp = b.fasta_comp(n)
    t3 = time.perf_counter()
    seqs = t[:, seq_cols]
    for c in set(letters):
        assert (torch.from_numpy(comp[:, c]).to(dev) == (seqs == c).sum(dim=1)).all(), chr(c)
    assert int(comp.sum()) == n * L
    # names: hash table + sort + the .fxi as pages
    t4 = time.perf_counter()
    order, ndup = b.names_sort(0, n)
    t5 = time.perf_counter()
    assert ndup == 0 and (order == i).all()                  # zero-padded numbers: already in order
    ln = rows["name_len"].astype(np.int64)
    packed, offs = b.names_pack(0, n, guess=int(ln.sum()))
    path = "/dev/shm/manyrec.fxi" if os.path.isdir("/dev/shm") else "/tmp/manyrec.fxi"
    if os.path.exists(path):
        os.remove(path)
    t6 = time.perf_counter()
    db = fxi.write_fasta_bulk(path, packed[:int(offs[-1])], offs, rows, s.seq_len, order=order)
    db.close()
    t7 = time.perf_counter()
    # the comp table of the same file: sparse triples off the GPU, table + seqidx as pages
    t8 = time.perf_counter()
    seqid, abc, num, total = b.fasta_comp_sparse(guess=n * 12)
    t9 = time.perf_counter()
    rec_nz, letter_nz = np.nonzero(comp)
    assert (seqid == rec_nz + 1).all() and (abc == letter_nz).all() and (num == comp[rec_nz, letter_nz]).all()
    assert (total == comp.sum(axis=0)).all()
    t10 = time.perf_counter()
    db = fxi.write_fasta_comp_bulk(path, np.concatenate([seqid, np.zeros(128, dtype=np.int64)]),
                                   np.concatenate([abc, np.arange(128, dtype=np.int64)]), np.concatenate([num, total]))
    db.close()
    t11 = time.perf_counter()
    db = sqlite3.connect(path)
    assert db.execute("PRAGMA integrity_check").fetchall() == [("ok",)]
    assert db.execute("SELECT count(*) FROM comp").fetchone()[0] == len(seqid) + 128
    j = n // 2
    assert db.execute("SELECT abc, num FROM comp WHERE seqid=? ORDER BY ID", (j + 1,)).fetchall() == \
        [(int(a), int(c)) for a, c in zip(np.nonzero(comp[j])[0], comp[j][np.nonzero(comp[j])[0]])]
    rng = np.random.default_rng(1)
    for j in rng.integers(0, n, 100).tolist():
        assert db.execute("SELECT ID, boff, slen FROM seq WHERE chrom=?", ("tx%07d" % j,)).fetchone() == (j + 1, j * rec + hl, L)
    db.close()
    size = os.path.getsize(path)
    os.remove(path)
    prof = {k: round(v[0] / v[1], 4) for k, v in b.prof_read().items()}
    print(json.dumps({"workload": "synthetic %s FASTA %d records x %d residues (%.2f GB), %d-column lines" % ("protein" if protein else "DNA", n, L, nb / 1e9, W),
                      "index_build_ms": round((t1 - t0) / 3 * 1e3, 3), "index_build_GBps": round(nb / ((t1 - t0) / 3) / 1e9, 1),
                      "composition_ms": round((t3 - t2) * 1e3, 2), "names_sort_ms": round((t5 - t4) * 1e3, 1),
                      "comp_sparse_ms": round((t9 - t8) * 1e3, 1), "comp_rows_M": round(len(seqid) / 1e6, 2), "comp_table_write_s": round(t11 - t10, 2),
                      "fxi_write_s": round(t7 - t6, 2), "fxi_MB": round(size / 1e6, 1), "fxi_rows_per_s_M": round(n / (t7 - t6) / 1e6, 2),
                      "kernels_ms_avg": prof, "verified": True}))


if __name__ == "__main__":
    main()
