"""-m gpu: bench.py's N>1 path end to end on the single GPU of the test box:
two ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one device),
everything else -- the pieces written into ONE file, every rank opening only its byte
range of it (fx_open_file_range), shard build on the HIP path, all-gather of summaries,
stitch, local fetch, the merged .fxi, fetches over the whole stream through ShardFetcher,
full-size parity checks -- is the code the 8-GPU run executes.  The RCCL flavour of the
collective path runs in test_nccl_collective_path_on_one_gpu (world size 1)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _last_json(out):
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("world", [2, 3, 8])
def test_bench_two_ranks_one_gpu(world):
    """`python bench.py --gpus N` with NO launcher around it (the way the driver runs --gpus 1): bench.py starts its N ranks
    itself; fewer devices than ranks -> gloo, the ranks share the device.  The weak line (configs[4] shape) carries the
    strong leg (ONE file split over the ranks) beside it.  world = 8: the driver's multi-GPU command shape with small inputs (the
    full default sizes on one device: profiles/r05_bench_8ranks_gloo_one_gpu.json, 25 s of wall)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(FX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--gbp", "0.05", "--queries", "20000", "--fastq-reads", "3e5"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _last_json(out)
    assert line["n_gpus"] == world and line["parity_verified_full_size"] is True
    assert line["value"] > 0 and line["scaling"] == "weak"
    fqs = line["fastq_strong"]                                # round 4: ONE FASTQ file over the same ranks -- build, base / meta, ONE .fxi, routed fetch
    assert fqs["rows_base_meta_fetch_equal_generator"] is True and fqs["reads_indexed"] == 300000 and fqs["build_s"] > 0 and fqs["fetch_routed_s"] > 0
    sf = line["sharded_file"]
    assert sf["merged_fxi_rows_equal_plan"] is True and sf["every_query_answered_once_and_sample_equals_file"] is True
    assert sf["open_range_s"] > 0 and sf["queries_crossing_a_cut"] >= 0
    st = line["strong"]
    assert all(v is True for v in st["parity"].values()), st["parity"]
    assert st["Gbp_per_s"] > 0 and st["e2e"]["total_s"] > 0 and st["file_bytes_per_gpu"] * world <= st["file_bytes"] + world


@pytest.mark.parametrize("world", [1, 2])
def test_bench_strong_scaling(world):
    """--scaling strong: ONE file over N ranks (N = 1: the same code without a process group); started without a launcher
    and, for N = 2 on this one-GPU box, WITHOUT naming a backend: bench.py picks gloo when there are fewer devices than ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FX_BENCH_BACKEND")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    if world > 1 and torch.cuda.device_count() >= world:
        env["FX_BENCH_BACKEND"] = "gloo"                      # (the RCCL flavour of the same run: test_bench_over_rccl_two_gpus)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--scaling", "strong", "--steps", "3", "--warmup", "1",
           "--gbp", "0.05", "--queries", "30000"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _last_json(out)
    assert line["n_gpus"] == world and line["scaling"] == "strong" and line["value"] > 0 and line["parity_verified_full_size"] is True
    st = line["strong"]
    assert all(v is True for v in st["parity"].values()), st["parity"]
    assert st["e2e"]["open_range_s"] > 0 and st["e2e"]["fetch_1M_host_to_host_s"] > 0 and line["roofline"]["achieved"] > 0
    if world > 1:
        assert "gloo" in line["config"]["parallelism"] and st["queries_per_gpu_max"] < 30000


def test_bench_over_rccl_two_gpus():
    """N = min(2, devices) ranks over RCCL -- the real thing, wherever the box has two GPUs (the driver's 8-GPU node; the
    one-GPU test box skips).  Weak line + strong leg, every parity flag."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU here: RCCL refuses two ranks on one device (its world-size-1 run is test_nccl_collective_path_on_one_gpu)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FX_BENCH_BACKEND")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--gbp", "0.2", "--queries", "100000"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _last_json(out)
    assert line["n_gpus"] == 2 and "(nccl)" in line["config"]["parallelism"] and line["parity_verified_full_size"] is True
    assert all(v is True for v in line["strong"]["parity"].values())


@pytest.mark.parametrize("flavour", ["fx", "torch"])
def test_nccl_collective_path_on_one_gpu(tmp_path, flavour):
    """flavour fx: the all-gather is the library's own (fx_comm_init + fx_fasta_build_sharded_begin: ncclAllGather on the
    handle's stream, the entry a C caller uses); torch: all_gather_into_tensor of the process group.
    VERDICT r1 #5: the `nccl` (= RCCL) branch of the sharded build -- fx_shard_summary_dev into the send buffer,
    all_gather_into_tensor on torch's stream ordered against the library's stream with ExternalStream events,
    fx_fasta_stitch_dev (k_stitch_tail) -- executed on an MI355X with a process group of ONE rank (force_collective):
    the rows must equal the plain single-handle build of the same file, fetches enqueued behind it included."""
    script = tmp_path / "nccl_one.py"
    script.write_text('''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="%d", RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
from pyfastx_amd import _lib, shard, synth
plan = synth.fasta_plan(total_bp=30_000_000, seed=5)
blob, flat, fs = synth.fasta_generate(plan, dev, keep_flat=True)
nb = int(plan["n_bytes"])
path = %r
blob[:nb].cpu().numpy().tofile(path)
job = shard.ShardedFasta.from_file(path, dev, 0, 1, force_collective=True)
assert job.comm_dev.type == "cuda" and dist.get_backend() == "nccl"
ids, st, sp, strand = synth.fasta_queries(plan, n=50000, seed=3)
d = lambda x: torch.from_numpy(x).to(dev)
d_out = torch.zeros(50000 * 100, dtype=torch.uint8, device=dev); d_len = torch.zeros(50000, dtype=torch.int64, device=dev)
d_off = torch.arange(50000, device=dev, dtype=torch.int64) * 100
for _ in range(3):
    job.build_async()                                     # summary kernel -> RCCL all-gather -> stitch kernel, stream-ordered
    job.fetch_local(50000, d(ids), d(st), d(sp), d((strand * 6).astype(np.uint8)), d_out, d_off, d_len)
    s = job.finish(); job.sync()
rows = job.local_rows()
ref = _lib.Blob.from_file(path)
rs = ref.fasta_build()
want = ref.fasta_table(rs.n_seq)
assert s.n_seq == rs.n_seq == len(plan["slen"])
for k in want:
    assert (rows[k] == want[k]).all(), k
assert (job.blob.fasta_line_regular(s.n_seq) == ref.fasta_line_regular(rs.n_seq)).all()
if %r == "fx":
    assert job.fxcomm is not None and job._all is None     # the library's communicator did the all-gather
    gathered = job.fxcomm.summaries(job.blob)[0].to_array()
else:
    assert job.fxcomm is None
    gathered = job._all.cpu().numpy()
mine = job.blob.shard_summary().to_array()
assert (gathered == mine).all()                            # what the all-gather delivered is this shard's summary
exp = synth.expected_fetch(flat, fs, ids, st, 100, strand, dev)
assert bool((d_out.view(50000, 100) == exp).all()) and bool((d_len == 100).all())
dist.destroy_process_group()
print("NCCL_PATH_OK")
''' % (ROOT, _free_port(), str(tmp_path / "one.fa"), flavour))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FX_COMM=flavour))
    assert out.returncode == 0 and "NCCL_PATH_OK" in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])


def test_fx_comm_without_torch(tmp_path):
    """The collective under the C ABI with NOTHING but the library in the process (no torch, no process group): RCCL is
    loaded by fx_comm_unique_id, a communicator of one rank builds a FASTA and a FASTQ shard through
    fx_fasta_build_sharded / fx_fastq_build_sharded, fx_comm_allgather moves a host array -- rows equal the plain build."""
    script = tmp_path / "comm_one.py"
    script.write_text('''
import os, sys
import numpy as np
sys.path.insert(0, %r)
from pyfastx_amd import _lib
assert "torch" not in sys.modules
raw = open(os.path.join(%r, "tests", "data", "test.fa"), "rb").read()
rawq = open(os.path.join(%r, "tests", "data", "test.fq"), "rb").read()
c = _lib.Comm(0, 1, _lib.Comm.unique_id(), 0)
assert "torch" not in sys.modules
b = _lib.Blob.from_bytes(raw)
b.set_shard(0, 10, True)
b.fasta_build_sharded_begin(c)
s = b.fasta_build_end()
ref = _lib.Blob.from_bytes(raw); rs = ref.fasta_build()
assert s.n_seq == rs.n_seq == 211
t, w = b.fasta_table(211), ref.fasta_table(211)
assert all((t[k] == w[k]).all() for k in w)
S = c.summaries(b)
assert len(S) == 1 and S[0].n_hdr == 211 and S[0].to_array().tolist() == b.shard_summary().to_array().tolist()
q = _lib.Blob.from_bytes(rawq)
q.set_shard(0, 10, True)
sq = q.fastq_build_sharded(c)
rq = _lib.Blob.from_bytes(rawq); rsq = rq.fastq_build()
assert (sq.n_reads, sq.size) == (rsq.n_reads, rsq.size) == (800, 120000)
tq, wq = q.fastq_table(800), rq.fastq_table(800)
assert all((tq[k] == wq[k]).all() for k in wq)
got = c.allgather(np.arange(7, dtype=np.int64) * 3)
assert got.shape == (1, 7) and got[0].tolist() == [0, 3, 6, 9, 12, 15, 18]
c.close()
print("FX_COMM_OK")
''' % (ROOT, ROOT, ROOT))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FX_NO_TORCH="1"))
    assert out.returncode == 0 and "FX_COMM_OK" in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])


def test_bench_single_small():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--gbp", "0.05",
           "--queries", "20000", "--c3-reads", "2e5", "--c3-sample", "5e4"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["parity_verified_full_size"] is True and line["roofline"]["achieved"] > 0
    assert line["cpu_baseline"]["kind"] in ("reference", "port")
    e = line["e2e"]                                       # file -> .fxi and host -> host answers, in the same run
    assert e["open_file_s"] > 0 and e["fxi_durable_s"] > 0 and e["fetch_many_1M_host_to_host_s"] > 0     # (medians of 3 at 50 MB: no order between them)
    if line["cpu_baseline"]["kind"] == "reference":
        assert line["cpu_baseline"]["rows_equal_gpu"] is True and line["cpu_baseline"]["fetch_bytes_equal_gpu"] is True
        assert e["fetch_bytes_equal_reference"] is True and line["speedup_vs_cpu"] > 0
        assert line["c3"]["file_sample"]["rows_equal_reference"] is True
        assert line["c3"]["file_sample"]["fetch_bytes_equal_reference"] is True
        assert line["c3"]["file_sample"]["fastx"]["tuples_equal_reference"] is True
        assert line["c4"]["rows_equal_reference"] is True and line["c4"]["fetch_sample_equal_reference"] is True
    assert line["c3"]["full"]["rows_base_meta_fetch_equal_generator"] is True
    assert line["c4"]["inflated_size_ok"] is True and line["c4"]["roofline"]["achieved"] > 0
    # the dominant kernel's HBM traffic comes from the counters of THIS run (two rocprofv3 --pmc passes over a child)
    r = line["roofline"]
    assert r["traffic_source"].startswith("measured in this run"), r["traffic_source"]
    assert 0.5 * r["algorithmic_bytes_per_launch"] < r["traffic"] < 2.0 * r["algorithmic_bytes_per_launch"]


def test_bench_sharded_path_over_nccl_with_one_rank():
    """bench.py's N > 1 code path -- the pieces written into one file, fx_open_file_range, the step with the all-gather and
    the stitch, composition across cuts, the merged .fxi, ShardFetcher with its exchange -- with the `nccl` backend and a
    process group of ONE rank (FX_BENCH_FORCE_SHARDED): every collective call of that path runs over RCCL on an MI355X
    before an 8-GPU node ever sees it."""
    env = dict(os.environ, FX_BENCH_FORCE_SHARDED="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--gbp", "0.05",
           "--queries", "20000"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["parity_verified_full_size"] is True and "1 all-gather (nccl)" in line["config"]["parallelism"]
    sf = line["sharded_file"]
    assert sf["merged_fxi_rows_equal_plan"] is True and sf["every_query_answered_once_and_sample_equals_file"] is True
