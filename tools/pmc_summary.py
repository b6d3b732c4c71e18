#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected
separately, as MI355X_MICROARCH.md prescribes) -> the JSON committed under profiles/.
usage: pmc_summary.py <dir_fetch> <dir_write> <kernel substring> <file_bytes> [out.json]
Counter CSVs come from `rocprofv3 --pmc X --kernel-trace --output-format csv`."""
import csv
import glob
import json
import os
import sys


def load(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                k = row["Kernel_Name"]
                a = acc.setdefault(k, [0.0, set()])
                a[0] += float(row["Counter_Value"])
                a[1].add(row.get("Dispatch_Id", len(a[1])))
    return {k: (v[0], max(len(v[1]), 1)) for k, v in acc.items()}


def main():
    dfetch, dwrite, pat, nbytes = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    fe, wr = load(dfetch, "FETCH_SIZE"), load(dwrite, "WRITE_SIZE")
    allk = {}
    for k in sorted(set(fe) | set(wr)):
        if "fx::" not in k:
            continue
        short = k.split("(")[0].replace("void ", "")
        e = allk.setdefault(short, {})
        if k in fe:
            e["FETCH_SIZE_KB"] = round(fe[k][0] / fe[k][1], 1); e["launches"] = fe[k][1]
        if k in wr:
            e["WRITE_SIZE_KB"] = round(wr[k][0] / wr[k][1], 1)
    key = next((k for k in allk if pat in k), None)
    out = {"source": "rocprofv3 --pmc FETCH_SIZE --kernel-trace / rocprofv3 --pmc WRITE_SIZE --kernel-trace (separate passes) "
                     "-- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-verify, 1x MI355X",
           "kernel": key, "file_bytes": nbytes, "all_kernels_raw_KB": allk}
    if key:
        f_kb, w_kb = allk[key].get("FETCH_SIZE_KB", 0.0), allk[key].get("WRITE_SIZE_KB", 0.0)
        rd, wrb = int(f_kb * 1024 * 2), int(w_kb * 1024)
        out.update({"FETCH_SIZE_KB_avg": f_kb, "WRITE_SIZE_KB_avg": w_kb,
                    "correction": "FETCH_SIZE x 1024 x 2 (gfx950 reports exactly half of a wide coalesced 16 B/lane stream, "
                                  "MI355X_MICROARCH.md HBM section; check: raw value ~ file_bytes / 2); WRITE_SIZE x 1024 uncorrected",
                    "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wrb, "hbm_bytes_per_launch": rd + wrb,
                    "ratio_to_algorithmic": round((rd + wrb) / nbytes, 4)})
    text = json.dumps(out, indent=1)
    if len(sys.argv) > 5:
        open(sys.argv[5], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
