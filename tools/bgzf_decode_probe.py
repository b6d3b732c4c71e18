#!/usr/bin/env python3
"""Kernel-time probe for the BGZF decoder: the C4 file once, then k_bgzf_* times for every library named on the command
line (experiment builds under build/, chosen through FX_LIBFXGPU).  usage: python tools/bgzf_decode_probe.py [gbp] lib.so ..."""
import json
import os
import subprocess
import sys
import tempfile
from multiprocessing import get_context

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _part(chunk):
    from pyfastx_amd import synth
    return synth.bgzf_compress(chunk)[:-28]


CHILD = r'''
import sys, json
sys.path.insert(0, %r)
from pyfastx_amd import _lib
_lib.lib().fx_prof_default(1)
_lib.Blob.from_file(%r).close()
out = []
for _ in range(3):
    b = _lib.Blob.from_file(%r)
    out.append({k: round(v[0] / v[1], 3) for k, v in b.prof_read().items()})
    b.close()
print(json.dumps([o.get('k_bgzf_decode') for o in out]), json.dumps(out[-1]))
'''


def main():
    gbp = float(sys.argv[1])
    libs = sys.argv[2:]
    keep = os.environ.get("FX_PROBE_FILE")                 # generate once, reuse across profiler passes
    if keep and os.path.exists(keep):
        path, d = keep, None
    else:
        import torch
        from pyfastx_amd import synth
        dev = torch.device("cuda", 0)
        plan = synth.fasta_plan(total_bp=int(gbp * 1e9))
        blob, _, _ = synth.fasta_generate(plan, dev, keep_flat=False)
        host = blob[:int(plan["n_bytes"])].cpu().numpy()
        del blob
        torch.cuda.empty_cache()
        d = None if keep else tempfile.mkdtemp(prefix="fxprobe")
        path = keep or os.path.join(d, "c4.fa.gz")
        with open(path, "wb") as f:
            f.write(synth.bgzf_compress_parallel(host))
        del host
    if not libs:                                            # in this process (what a profiler attached to it sees)
        from pyfastx_amd import _lib
        _lib.lib().fx_prof_default(1)
        for _ in range(3):
            b = _lib.Blob.from_file(path)
            print("in-process", json.dumps({k: round(v[0] / v[1], 3) for k, v in b.prof_read().items()}), flush=True)
            b.close()
    for lib in libs:
        env = dict(os.environ)
        name = lib
        if "@" in lib:                                      # lib.so@KEY=VAL,KEY=VAL: the same library under other settings
            lib, kv = lib.split("@", 1)
            env.update(dict(x.split("=", 1) for x in kv.split(",") if x))
        env["FX_LIBFXGPU"] = os.path.join(ROOT, lib)
        r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, path, path)], env=env, capture_output=True, text=True)
        print(name, r.stdout.strip() or r.stderr[-500:], flush=True)
    if d:
        os.unlink(path)
        os.rmdir(d)


if __name__ == "__main__":
    main()
