#!/usr/bin/env python3
"""Why is the FIRST staging of a freshly written file slow (1.4-1.7 s against 0.65 s for 34.8 GB)?  Hypothesis: the file's page
cache pages lie on the NUMA node of the thread that wrote them; the staging lanes run on the CPUs next to the device; the first
pass reads remote memory and the kernel's NUMA balancing moves the pages, so the second pass is local.  Test: write the same file
(a) from an unbound thread, (b) from a thread bound to the device's local CPUs (/sys/class/drm/card*/device/local_cpulist), and
stage each twice.  usage: python tools/numa_stage_probe.py [GB]"""
import glob
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfastx_amd import _lib  # noqa: E402


def cpulist(text):
    out = []
    for part in text.strip().split(","):
        if "-" in part:
            a, b = part.split("-")
            out += list(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    return out


def main():
    gb = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
    n = int(gb * (1 << 30))
    lists = {}
    for f in glob.glob("/sys/class/drm/card*/device/local_cpulist"):
        try:
            lists[f] = open(f).read().strip()
        except OSError:
            pass
    out = {"GB": gb, "local_cpulists": lists, "affinity_at_start": len(os.sched_getaffinity(0))}
    try:
        out["numa_balancing"] = open("/proc/sys/kernel/numa_balancing").read().strip()
    except OSError:
        out["numa_balancing"] = None
    _lib.Blob.from_bytes(b">a\nACGT\n").close()
    chunk = np.full(1 << 28, 65, dtype=np.uint8)
    chunk[60::61] = 10
    all_cpus = sorted(os.sched_getaffinity(0))
    near = cpulist(next(iter(lists.values()))) if lists else all_cpus
    near = [c for c in near if c in all_cpus] or all_cpus
    for tag, cpus in (("unbound_writer", all_cpus), ("writer_on_the_devices_cpus", near), ("writer_on_the_other_cpus", [c for c in all_cpus if c not in near] or all_cpus)):
        os.sched_setaffinity(0, cpus)
        path = "/dev/shm/fx_numa_probe.bin"
        t0 = time.perf_counter()
        with open(path, "wb") as f:
            f.write(b">r\n")
            for _ in range(n // chunk.size):
                f.write(memoryview(chunk))
        tw = time.perf_counter() - t0
        os.sched_setaffinity(0, all_cpus)
        runs = []
        for _ in range(3):
            t0 = time.perf_counter()
            b = _lib.Blob.from_file(path)
            runs.append(round(time.perf_counter() - t0, 3))
            al, st = _lib.open_laps()
            runs[-1] = {"open_s": runs[-1], "alloc_s": round(al, 3), "stage_s": round(st, 3)}
            b.close()
        os.unlink(path)
        out[tag] = {"write_s": round(tw, 2), "cpus": len(cpus), "opens": runs}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
