#!/usr/bin/env python3
"""FETCH_SIZE calibration: known bytes of tools/gathercal.hip's kernels against what rocprofv3 reported for them.
usage: gathercal_summary.py gpurun_out/<tag>   (reads gathercal_known.txt, gc_fetch.txt, gc_req.txt, gc_hit.txt written by tools/gpu.sh gathercal)"""
import ast
import json
import os
import sys


def rows(path):
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        line = line.strip()
        if not line.startswith("("):
            continue
        key, vals = line.split(") {", 1)
        name = ast.literal_eval(key + ")")[1]
        vals = ast.literal_eval("{" + vals)
        k = next((c for c in ("cal_stream", "cal_line128", "cal_half64", "cal_unal100", "cal_unal300") if c in name), None)
        if k:
            acc = out.setdefault(k, {})
            for c, v in vals.items():
                acc.setdefault(c, []).append(v)
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in out.items()}


def main():
    d = sys.argv[1]
    known = {}
    for line in open(os.path.join(d, "gathercal_known.txt")):
        if line.startswith("{"):
            r = json.loads(line)
            known[r["kernel"]] = r
    fetch, req, hit = rows(os.path.join(d, "gc_fetch.txt")), rows(os.path.join(d, "gc_req.txt")), rows(os.path.join(d, "gc_hit.txt"))
    print("# FETCH_SIZE (KB, per launch) against the bytes each kernel is known to read; factor = known / (FETCH_SIZE x 1024)")
    for k, r in known.items():
        f = fetch.get(k, {}).get("FETCH_SIZE")
        line = {"kernel": k, "ms_avg": r["ms_avg"], "known_64B": r["known_bytes_64B_blocks"], "known_128B": r["known_bytes_128B_blocks"],
                "offsets_bytes": 0 if k == "cal_stream" else r["offsets_bytes"]}
        if f:
            raw = f * 1024
            line["FETCH_SIZE_bytes"] = int(raw)
            off = line["offsets_bytes"]
            # the offsets are a coalesced 8 B/lane stream; their own factor is taken as the stream's (2.0) -- 8 MB against >= 64 MB
            line["factor_vs_64B_blocks"] = round((r["known_bytes_64B_blocks"] + off) / raw, 3)
            line["factor_vs_128B_blocks"] = round((r["known_bytes_128B_blocks"] + off) / raw, 3)
        line.update({c: int(v) for c, v in req.get(k, {}).items()})
        line.update({c: int(v) for c, v in hit.get(k, {}).items()})
        print(json.dumps(line))


if __name__ == "__main__":
    main()
