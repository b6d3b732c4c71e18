"""The staging lanes of fx_open_file (csrc/fxgpu.hip: stage_plain_file, StageAsync): pieces of the file go to the device in
order, taken by whichever lane is free; extra lanes join in when the file turns out to be cold (one that nobody has read since
it was written comes out of the page cache at a fraction of the usual rate).  In a process of its own the piece size is
1 MiB and every file counts as cold (FX_STAGE_COLD_FORCE=1), so that a file of a few dozen MiB goes through all of it: what
lands in HBM is the file, for a plain file (one call, and the asynchronous open the pipelined constructor uses) and for a
BGZF file large enough to be inflated in groups behind its staging."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, zlib
import numpy as np
sys.path.insert(0, %(root)r)
from pyfastx_amd import _lib, synth
d = %(dir)r
rng = np.random.default_rng(3)
raw = b"".join(b">r%%d\n%%s\n" %% (i, bytes(rng.integers(65, 85, 1000 + (i %% 97), dtype=np.uint8))) for i in range(40000))      # ~42 MB
p = os.path.join(d, "plain.fa")
open(p, "wb").write(raw)
want = zlib.crc32(raw)
for T in ("2", "8"):
    os.environ["FX_STAGE_THREADS"] = T                          # (read once per process: the first value stands; the loop is for the record)
    b = _lib.Blob.from_file(p)
    got = b.read_bytes(0, len(raw))
    assert len(got) == len(raw) and zlib.crc32(bytes(got)) == want
    b.close()
b = _lib.Blob.from_file_async(p)
b.stage_wait(5 << 20)
assert bytes(b.read_bytes(0, 5 << 20)) == raw[:5 << 20]
b.stage_wait(-1)
assert zlib.crc32(bytes(b.read_bytes(0, len(raw)))) == want
b.close()
z = os.path.join(d, "plain.fa.gz")
open(z, "wb").write(synth.bgzf_compress(raw))
os.environ["FX_BGZF_GROUP"] = str(4 << 20)                      # groups of 4 MiB: the inflate runs behind the staging
b = _lib.Blob.from_file(z)
assert b.size == len(raw) and zlib.crc32(bytes(b.read_bytes(0, len(raw)))) == want
b.close()
print("ok")
'''


def test_pieces_in_order_and_the_lanes_of_a_cold_file(tmp_path):
    env = dict(os.environ, FX_STAGE_PIECE_MB="1", FX_STAGE_COLD_FORCE="1", FX_STAGE_THREADS="3", FX_STAGE_EXTRA_THREADS="5", FX_TRACE_STAGE="1")
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "dir": str(tmp_path)}], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-2000:], out.stderr[-3000:])
    assert "the extra lanes join in" in out.stderr               # the decision was taken, and for a cold file
