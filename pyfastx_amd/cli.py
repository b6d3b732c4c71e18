"""Batched forms of the reference's `pyfastx subseq`, `pyfastx sample` and `pyfastx extract` (pyfastxcli.py:240-387,
SURVEY 8f-2): the same arguments, the same bytes on the output -- but a region file, a BED file, a name list or a sample
becomes ONE batch for the GPU (Fasta.fetch_many / raw_many, Fastq.raw_many) instead of one `fa.fetch()` / `fx[name].raw`
per line.  The rest of the reference's command line (build, info, split, fq2fa, ...) is out of scope.

    python -m pyfastx_amd subseq  [-r REGION_FILE | -b BED_FILE] [-o OUT] fasta [chr:start-end ...]
    python -m pyfastx_amd sample  (-n NUM | -p PROP) [-s SEED] [--sequential-read] [-o OUT] fastx
    python -m pyfastx_amd extract [-l LIST_FILE] [--sequential-read] [-o OUT] fastx [name ...]
"""
import argparse
import math
import random
import re
import sys

import numpy as np

from . import api


def _format_check(path):
    """fastx_format_check (pyfastxcli.py:10-30)."""
    c = api._first_nonspace(path, api._is_gzip(path))
    if c == ord(">"):
        return "fasta"
    if c == ord("@"):
        return "fastq"
    raise Exception("Input file %s is not fasta or fastq" % path)


def _out(args):
    return open(args.out_file, "wb") if args.out_file else sys.stdout.buffer


def _done(args, fw):
    if args.out_file:
        fw.close()
    else:
        fw.flush()


def subseq_records(fa, chroms, starts, ends, by_slice=False):
    """1-based inclusive (chrom, start, end) regions -> the bytes `pyfastx subseq` writes: `>chrom:start-end\\nSEQ\\n` per
    region, every sequence from ONE batched fetch.  Errors as the reference raises them per region: an unknown name
    (NameError from fetch(), KeyError from fa[chrom] for command-line regions), start > end (ValueError)."""
    starts = np.asarray(starts, dtype=np.int64)
    ends = np.asarray(ends, dtype=np.int64)
    t = fa._table()
    ix = t["index"]
    ids = np.empty(len(chroms), dtype=np.int64)
    for k, c in enumerate(chroms):
        i = ix.get(c)
        if i is None:
            raise (KeyError("%s does not exist in fasta file" % c) if by_slice else NameError("Sequence %s does not exists" % c))
        ids[k] = i
    if not by_slice and (starts > ends).any():
        raise ValueError("start position should less than end position")      # fasta.c:425-428
    slen = t["slen"][ids]
    a = np.clip(starts - 1, 0, slen)                          # fa[chrom][start-1:end]: Python slice clamping
    b = np.clip(ends, a, slen)
    buf, offs = fa.fetch_many(ids, a, b)
    view = memoryview(np.ascontiguousarray(buf))
    o = offs.tolist()
    parts = []
    for k, c in enumerate(chroms):
        parts.append((">%s:%d-%d\n" % (c, starts[k], ends[k])).encode("utf-8", "surrogateescape"))
        parts.append(view[o[k]:o[k + 1]])
        parts.append(b"\n")
    return b"".join(parts)


def fastx_subseq(args):
    fa = api.Fasta(args.fastx)
    fw = _out(args)
    if args.region_file or args.bed_file:
        chroms, starts, ends = [], [], []
        with open(args.region_file or args.bed_file) as fh:
            for line in fh:
                chrom, start, end = line.strip().split()
                chroms.append(chrom)
                starts.append(int(start) + (1 if args.bed_file else 0))       # BED: 0-based start (pyfastxcli.py:262)
                ends.append(int(end))
        fw.write(subseq_records(fa, chroms, starts, ends))
    elif args.regions:
        chroms, starts, ends = [], [], []
        for region in args.regions:
            chrom, start, end = re.split("[:-]", region)
            chroms.append(chrom); starts.append(int(start)); ends.append(int(end))
        fw.write(subseq_records(fa, chroms, starts, ends, by_slice=True))
    else:
        raise Exception("no regions or region file provided")
    _done(args, fw)


def _open_fastx(path):
    kind = _format_check(path)
    return api.Fasta(path) if kind == "fasta" else api.Fastq(path)


def fastx_sample(args):
    fx = _open_fastx(args.fastx)
    if args.num is not None and args.num > 0:
        seq_num = min(args.num, len(fx))
    elif args.prop is not None and 0 < args.prop <= 1:
        seq_num = math.ceil(len(fx) * args.prop)
    else:
        raise RuntimeError("specify a right seq number or proportion")
    if args.seed:
        random.seed(args.seed)
    selected = random.sample(range(len(fx)), k=seq_num)       # the reference's draw (pyfastxcli.py:308-310)
    selected.sort()                                            # both of its write loops emit in file order
    buf, _ = fx.raw_many(np.asarray(selected, dtype=np.int64))
    fw = _out(args)
    fw.write(memoryview(np.ascontiguousarray(buf)))
    _done(args, fw)


def fastx_extract(args):
    fx = _open_fastx(args.fastx)
    if args.list_file:
        with open(args.list_file) as fh:
            names = [line.strip() for line in fh]
        if args.sequential_read:                              # file order, every listed name once (pyfastxcli.py:349-360)
            want = set(names)
            if isinstance(fx, api.Fasta):
                ix = fx._table()["index"]
                ids = sorted(ix[n] for n in want if n in ix)
            else:
                got = fx.ids_of(sorted(want))
                ids = sorted(int(i) for i in got if i >= 0)
            buf, _ = fx.raw_many(np.asarray(ids, dtype=np.int64))
        else:
            buf, _ = fx.raw_many(names)
    elif args.names:
        buf, _ = fx.raw_many(list(args.names))
    else:
        raise Exception("no sequence name or list file provided")
    fw = _out(args)
    fw.write(memoryview(np.ascontiguousarray(buf)))
    _done(args, fw)


def main(argv=None):
    parser = argparse.ArgumentParser(prog="pyfastx_amd", usage="python -m pyfastx_amd COMMAND [OPTIONS]",
                                     description="batched subseq / sample / extract on an MI355X")
    sub = parser.add_subparsers(title="Commands", prog="pyfastx_amd", metavar="")
    p = sub.add_parser("subseq", help="get subsequences from fasta file by region")
    p.set_defaults(func=fastx_subseq)
    g = p.add_mutually_exclusive_group()
    g.add_argument("-r", "--region-file", help="tab-delimited file, one region per line, both start and end position are 1-based")
    g.add_argument("-b", "--bed-file", help="tab-delimited BED file, 0-based start position and 1-based end position")
    p.add_argument("-o", "--out-file", help="output file, default: output to stdout")
    p.add_argument("fastx", help="input fasta file, gzip support")
    p.add_argument("regions", nargs="*", help="format is chr:start-end, start and end position is 1-based, multiple regions were separated by space")
    p = sub.add_parser("sample", help="randomly sample sequences from fasta or fastq file")
    p.set_defaults(func=fastx_sample)
    g = p.add_mutually_exclusive_group(required=True)
    g.add_argument("-n", dest="num", type=int, help="number of sequences to be sampled")
    g.add_argument("-p", dest="prop", type=float, help="proportion of sequences to be sampled, 0~1")
    p.add_argument("-s", "--seed", type=int, default=None, help="random seed, default is the current system time")
    p.add_argument("--sequential-read", action="store_true", help="accepted for compatibility: records are always gathered in one batch")
    p.add_argument("-o", "--out-file", help="output file, default: output to stdout")
    p.add_argument("fastx", help="fasta or fastq file, gzip support")
    p = sub.add_parser("extract", help="extract full sequences or reads from fasta/q file")
    p.set_defaults(func=fastx_extract)
    p.add_argument("-l", "--list-file", help="a file containing sequence or read names, one name per line")
    p.add_argument("--reverse-complement", action="store_true", help="accepted and ignored, as in the reference (pyfastxcli.py:330-387 never reads it)")
    p.add_argument("--out-fasta", action="store_true", help="accepted and ignored, as in the reference")
    p.add_argument("--sequential-read", action="store_true", help="write the listed records in file order, each once")
    p.add_argument("-o", "--out-file", help="output file, default: output to stdout")
    p.add_argument("fastx", help="fasta or fastq file, gzip support")
    p.add_argument("names", nargs="*", help="sequence name or read name, multiple names were separated by space")
    args = parser.parse_args(argv)
    if hasattr(args, "func"):
        args.func(args)
    else:
        parser.print_help()


if __name__ == "__main__":
    main()
