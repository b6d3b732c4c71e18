#!/usr/bin/env python3
"""Timing probes of k_bgzf_decode_par (FX_BGZF_DBG): writes a BGZF file once, then opens it in child processes, one per probe."""
import os, subprocess, sys, tempfile
from multiprocessing import Pool
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def _comp(chunk):
    from pyfastx_amd import synth
    return synth.bgzf_compress(chunk)[:-28]

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        from pyfastx_amd import _lib
        try:
            _lib.Blob.from_file(sys.argv[2])
            _lib.Blob.from_file(sys.argv[2])
        except Exception as e:
            pass
        sys.exit(0)
    import torch
    from pyfastx_amd import synth
    gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    dev = torch.device("cuda", 0)
    plan = synth.fasta_plan(total_bp=int(gbp * 1e9))
    blob_t, _, _ = synth.fasta_generate(plan, dev, keep_flat=False)
    nb = int(plan["n_bytes"])
    host = blob_t[:nb].cpu().numpy().tobytes()
    step = 65280 * 64
    with Pool(min(64, os.cpu_count() or 8)) as pool:
        parts = pool.map(_comp, [host[a:a + step] for a in range(0, nb, step)])
    d = tempfile.mkdtemp(prefix="fxbgzf")
    path = os.path.join(d, "c4.fa.gz")
    open(path, "wb").write(b"".join(parts) + synth.bgzf_compress(b""))
    for dbg in sys.argv[2:] or ["8", "1", "2", "4", "16"]:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "child", path], env=dict(os.environ, FX_BGZF_DBG=dbg), capture_output=True, text=True)
        print("\n".join(l for l in out.stderr.splitlines() if "dbg=" in l))
    os.unlink(path)
