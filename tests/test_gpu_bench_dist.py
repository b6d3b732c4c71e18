"""-m gpu: bench.py's N>1 path end to end on the single GPU of the test box:
two ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one device),
everything else -- piece generation, cut shifting, shard build on the HIP path,
all-gather of summaries, stitch, local fetch, full-size parity checks -- is the
code the 8-GPU run executes."""
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


@pytest.mark.parametrize("world", [2, 3])
def test_bench_two_ranks_one_gpu(world):
    env = dict(os.environ, FX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--gbp", "0.05", "--queries", "20000"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == world and line["parity_verified_full_size"] is True
    assert line["value"] > 0 and line["scaling"] == "weak"


def test_bench_single_small():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--gbp", "0.05",
           "--queries", "20000", "--c3-reads", "2e5", "--c3-sample", "5e4"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["parity_verified_full_size"] is True and line["roofline"]["achieved"] > 0
    assert line["cpu_baseline"]["kind"] in ("reference", "port")
    e = line["e2e"]                                       # file -> .fxi and host -> host answers, in the same run
    assert e["open_file_s"] > 0 and e["fxi_durable_s"] >= e["open_file_s"] * 0.5 and e["fetch_many_1M_host_to_host_s"] > 0
    if line["cpu_baseline"]["kind"] == "reference":
        assert line["cpu_baseline"]["rows_equal_gpu"] is True and line["cpu_baseline"]["fetch_bytes_equal_gpu"] is True
        assert e["fetch_bytes_equal_reference"] is True and line["speedup_vs_cpu"] > 0
        assert line["c3"]["file_sample"]["rows_equal_reference"] is True
        assert line["c3"]["file_sample"]["fetch_bytes_equal_reference"] is True
        assert line["c4"]["rows_equal_reference"] is True and line["c4"]["fetch_sample_equal_reference"] is True
    assert line["c3"]["full"]["rows_base_meta_fetch_equal_generator"] is True
    assert line["c4"]["inflated_size_ok"] is True and line["c4"]["roofline"]["achieved"] > 0
