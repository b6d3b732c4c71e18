import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
from pyfastx_amd import _lib, synth
dev = torch.device("cuda", 0)
n = int(float(sys.argv[1]))
blob_t, cols = synth.fastq_generate(n, dev)
nb = int(cols["n_bytes"])
b = _lib.Blob.from_device(blob_t.data_ptr(), nb, device=0, keepalive=blob_t)
b.fastq_build(); base, meta = b.fastq_comp()
b.fastq_build(comp=True); b1, m1 = b.fastq_comp()
assert (b1 == base).all() and (m1 == meta).all()
b.prof_enable(1); b.prof_reset()
t0 = time.perf_counter()
for _ in range(4):
    b.fastq_build(comp=True); b.fastq_comp()
t1 = time.perf_counter()
print("one-read ms", (t1 - t0) / 4 * 1e3, {k: round(v[0] / v[1], 3) for k, v in b.prof_read().items()})
