#!/usr/bin/env python3
"""C4's BGZF file opened all at once (FX_BGZF_GROUP=0) and in groups of several sizes behind its staging: wall time of
Blob.from_file (min / median of 5) and the kernels' time per open.  usage: python tools/bgzf_group_probe.py [gbp]"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from pyfastx_amd import _lib, synth
    gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    dev = torch.device("cuda", 0)
    plan = synth.fasta_plan(total_bp=int(gbp * 1e9))
    blob, _, _ = synth.fasta_generate(plan, dev, keep_flat=False)
    host = blob[:int(plan["n_bytes"])].cpu().numpy()
    del blob
    torch.cuda.empty_cache()
    d = tempfile.mkdtemp(prefix="fxprobe")
    path = os.path.join(d, "c4.fa.gz")
    with open(path, "wb") as f:
        f.write(synth.bgzf_compress_parallel(host))
    _lib.lib().fx_prof_default(1)
    _lib.Blob.from_file(path).close()
    for group in [0] + [int(x) for x in (sys.argv[2:] or ["64", "128", "256", "384", "512"])]:
        os.environ["FX_BGZF_GROUP"] = str(group << 20)
        ts, kern = [], {}
        for _ in range(5):
            t0 = time.perf_counter()
            b = _lib.Blob.from_file(path)
            ts.append(time.perf_counter() - t0)
            kern = {k: round(v[0], 2) for k, v in b.prof_read().items()}
            b.close()
        ts.sort()
        print(json.dumps({"group_MiB": group, "open_ms_min": round(ts[0] * 1e3, 1), "open_ms_median": round(ts[2] * 1e3, 1),
                          "kernels_ms_per_open": kern, "kernels_sum_ms": round(sum(kern.values()), 2)}), flush=True)
    os.unlink(path)
    os.rmdir(d)


if __name__ == "__main__":
    main()
