#!/usr/bin/env python3
"""Where the time of making the Read objects of a batch goes (host only; run on the GPU box: its host differs from the dev container)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyfastx_amd as fx
from pyfastx_amd import _fxobj

k = 16384
names = ["SYN:1:FC:1:%04d:%05d:%09d" % (i // 50000, i * 7919 % 100000, i) for i in range(k)]
cols = np.arange(5 * k, dtype=np.int64).tobytes()
rng = np.random.default_rng(0)
seq = rng.integers(65, 85, 150 * k).astype(np.uint8)
qual = rng.integers(35, 71, 150 * k).astype(np.uint8)
offs = np.arange(k + 1, dtype=np.int64) * 150


class FQ:
    pass


fq = FQ()
res = {}
for label, s, q in (("numpy", seq, qual), ("bytes", seq.tobytes(), qual.tobytes())):
    t0 = time.perf_counter()
    for _ in range(60):
        objs = _fxobj.read_batch_cols(fx.Read, fq, names, cols, s, q, offs)
    res["read_batch_cols_%s_us" % label] = (time.perf_counter() - t0) / 60 / k * 1e6
t0 = time.perf_counter()
for _ in range(60):
    objs = _fxobj.read_batch_cols(fx.Read, fq, names, cols, seq, qual, offs)
    for r in objs:
        pass
res["batch_plus_bare_loop_us"] = (time.perf_counter() - t0) / 60 / k * 1e6
t0 = time.perf_counter()
for _ in range(60):
    objs = _fxobj.read_batch_cols(fx.Read, fq, names, cols, seq, qual, offs)
    for r in objs:
        r.seq, r.qual
res["batch_plus_two_getters_us"] = (time.perf_counter() - t0) / 60 / k * 1e6
b = seq.tobytes()
t0 = time.perf_counter()
for _ in range(60):
    l = [b[i * 150:(i + 1) * 150].decode("latin-1") for i in range(k)]
res["python_slice_decode_us"] = (time.perf_counter() - t0) / 60 / k * 1e6
t0 = time.perf_counter()
for _ in range(60):
    l = [fx.Read(fq, i, names[i], 40, 150, 0, 0) for i in range(k)]
res["python_ctor_us"] = (time.perf_counter() - t0) / 60 / k * 1e6
import gc
res["gc_tracked_read"] = gc.is_tracked(objs[0])
res["flags_have_gc"] = bool(fx.Read.__flags__ & (1 << 14))
print(res)
