"""Golden vectors for Fastx (fastx.c + kseq.c): what the REFERENCE's Fastx yields for the hand-written cases of
tests/kseq_cases.py and for seeded random files, for both builders and the comment / uppercase options.

Run here (the container that has oracle/_ref built from /root/reference): `python tests/golden/make_golden_fastx.py`
writes tests/golden/fastx.json and checks the oracle (oracle/fx_oracle.c: fxo_kseq + fxoracle.fastx_tuples) against every
vector on the way.  Inputs are stored as latin-1 text.  Cases in which the reference reads a quality buffer it never wrote
(a record without bases whose '+' line ends the stream, before any quality string: kseq.c:169-177 -- its value is whatever
the heap held) are left out; the oracle reports them as `undefined`."""
import json
import os
import random
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "oracle", "_ref"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import pyfastx  # noqa: E402  (the compiled reference)
import fxoracle  # noqa: E402
from kseq_cases import FIXED, gen  # noqa: E402

OPTIONS = [("fasta", False, False), ("fasta", True, True), ("fasta", False, True), ("fastq", False, False), ("fastq", False, True)]


def main():
    tmp = tempfile.mkdtemp(prefix="fxgoldx")
    rng = random.Random(20260926)
    inputs = list(FIXED) + [gen(rng) for _ in range(80)]
    cases, skipped = [], 0
    for k, data in enumerate(inputs):
        if fxoracle.kseq_undefined(data):
            skipped += 1
            continue
        p = os.path.join(tmp, "c%d.fx" % k)
        with open(p, "wb") as f:
            f.write(data)
        case = {"text": data.decode("latin-1"), "out": {}}
        for fmt, up, com in OPTIONS:
            theirs = [list(t) for t in pyfastx.Fastx(p, format=fmt, uppercase=up, comment=com)]
            ours = [list(t) for t in fxoracle.fastx_tuples(data, fmt, uppercase=up, comment=com)]
            assert theirs == ours, (k, fmt, up, com, data, theirs, ours)
            case["out"]["%s:%d:%d" % (fmt, up, com)] = theirs
        cases.append(case)
    with open(os.path.join(HERE, "fastx.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    print("fastx.json:", len(cases), "cases,", skipped, "undefined ones left out")


if __name__ == "__main__":
    main()
