#!/usr/bin/env python3
"""FASTQ build + composition, per-kernel averages (HIP events) and parity with the generator's truth.
usage: python tools/fq_comp_probe.py [n_reads] [rlen]     env FX_FQ_LPR: lanes per record in k_fastq_comp"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfastx_amd import _lib, synth  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
    rlen = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    dev = torch.device("cuda", 0)
    blob_t, cols = synth.fastq_generate(n, dev, rlen=rlen)
    nb = cols["n_bytes"]
    b = _lib.Blob.from_device(blob_t.data_ptr(), nb, device=0, keepalive=blob_t)
    b.fastq_build(); b.fastq_comp()
    b.prof_enable(True); b.prof_reset()
    for _ in range(5):
        s = b.fastq_build()
        base, meta = b.fastq_comp()
    prof = {k: round(v[0] / v[1], 4) for k, v in b.prof_read().items()}
    v = blob_t[:nb].view(n, cols["rec"])
    so, qo = int(cols["soff"][0]), int(cols["qoff"][0])
    seqs, q = v[:, so:so + rlen], v[:, qo:qo + rlen]
    want = [int((seqs == c).sum()) for c in b"ACGT"]
    want.append(n * rlen - sum(want))
    ok = base.tolist() == want and meta.tolist() == [rlen, rlen, int(q.min()), int(q.max()), 33] and s.n_reads == n
    print(json.dumps({"reads": n, "rlen": rlen, "GB": round(nb / 1e9, 2), "FX_FQ_LPR": os.environ.get("FX_FQ_LPR", "auto"),
                      "kernels_ms_avg": prof, "comp_equal_truth": ok}))
    if not ok:
        raise SystemExit("PARITY FAILURE")


if __name__ == "__main__":
    main()
