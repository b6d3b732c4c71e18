"""CPU, world_size 2 (and 3) over gloo: the N>1 path of the sharded index build
-- local shard summary -> ONE all-gather -> host stitch -> rows -- with the
shard-local scan supplied by the numpy test helper instead of the HIP kernels
(the product path needs a GPU; the collective + stitch logic is what runs here)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, raw, cuts, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import shard_ref
    from pyfastx_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bounds = [0] + list(cuts) + [len(raw)]
    rows, mine = shard_ref.local_scan(raw, bounds[rank], bounds[rank + 1])
    S = shard.allgather_summaries(mine, world, "cpu")                 # the one collective
    fix = shard.stitch_tail(S, rank)
    if fix is not None:
        for k, v in fix.items():
            rows[k][-1] = v
    offs, total = shard.id_offsets(S)
    gathered = [None] * world
    dist.all_gather_object(gathered, (offs[rank], rows))              # result collection, not the data path
    if rank == 0:
        q.put((total, gathered))
    dist.barrier()
    dist.destroy_process_group()


def _run(raw, cuts):
    world = len(cuts) + 1
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, raw, cuts, q)) for r in range(world)]
    for p in procs:
        p.start()
    total, gathered = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    out = {}
    for off, rows in gathered:
        for k, v in rows.items():
            out.setdefault(k, []).extend(v)
    return total, out


def _fasta(seed):
    rng = np.random.default_rng(seed)
    parts = []
    for i in range(9):
        parts.append(b">ctg%d  len=%d\tx\n" % (i, i))
        s = bytes(rng.choice(list(b"ACGTNacgt"), int(rng.integers(50, 3000))).astype(np.uint8))
        parts += [s[p:p + 60] + b"\n" for p in range(0, len(s), 60)]
    return b"".join(parts)


@pytest.mark.parametrize("world,seed", [(2, 1), (2, 2), (3, 3)])
def test_sharded_index_over_gloo(oracle, world, seed):
    raw = _fasta(seed)
    rng = np.random.default_rng(seed)
    cuts = sorted(set(int(x) for x in rng.integers(100, len(raw) - 100, world - 1)))
    total, got = _run(raw, cuts)
    recs, _ = oracle.fasta_index(raw)
    assert total == len(recs)
    for k in ("hoff", "boff", "blen", "slen", "llen", "elen", "norm", "dlen", "name_len"):
        assert got[k] == [int(x) for x in recs[k]], k


def test_cut_inside_header_and_first_line(oracle):
    raw = _fasta(5)
    h = raw.index(b">ctg4")
    for cut in (h + 3, h + 9, raw.index(b"\n", h) + 1 + 17):     # inside name, inside description, inside first seq line
        total, got = _run(raw, [cut])
        recs, _ = oracle.fasta_index(raw)
        for k in ("boff", "blen", "slen", "llen", "elen", "norm", "dlen", "name_len"):
            assert got[k] == [int(x) for x in recs[k]], (cut, k)


def _fetch_worker(rank, world, port, raw, cuts, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    import fxoracle as oracle
    from pyfastx_amd import shard
    from test_host_logic import _OracleShard, _shard_queries
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bases, ends = [0] + list(cuts), list(cuts) + [len(raw)]
    recs, _ = oracle.fasta_index(raw)
    table = {k: recs[k] for k in ("boff", "blen", "slen", "llen", "elen", "norm")}
    ids, st, sp, fl = _shard_queries(np.random.default_rng(77), recs, 400)     # the same batch on every rank
    # a query across each cut
    for c in cuts:
        i = int(np.searchsorted(recs["boff"], c, "right")) - 1
        ids, st, sp, fl = np.append(ids, i), np.append(st, 0), np.append(sp, int(recs["slen"][i])), np.append(fl, np.uint8(6))
    f = shard.ShardFetcher({rank: _OracleShard(oracle, raw, bases[rank], ends[rank])}, bases, ends, table,
                           exchange=shard.allgather_pieces)              # the only exchange: pieces of cross-cut queries
    qidx, buf, offs = f.fetch(ids, st, sp, flags_per_query=fl)
    gathered = [None] * world
    dist.all_gather_object(gathered, (qidx, buf, offs))                   # result collection for the check
    if rank == 0:
        q.put((ids, st, sp, fl, gathered))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,seed", [(2, 11), (3, 12)])
def test_sharded_fetch_over_gloo(oracle, world, seed):
    """One process per shard: every rank answers the queries whose first byte it holds, cross-cut pieces travel by
    all_gather_object (shard.allgather_pieces); together every query is answered exactly once and correctly."""
    from test_host_logic import _expected_fetch
    raw = _fasta(seed)
    rng = np.random.default_rng(seed)
    cuts = sorted(set(int(x) for x in rng.integers(100, len(raw) - 100, world - 1)))
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_fetch_worker, args=(r, world, port, raw, cuts, q)) for r in range(world)]
    for p in procs:
        p.start()
    ids, st, sp, fl, gathered = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    recs, _ = oracle.fasta_index(raw)
    seen = np.zeros(len(ids), dtype=np.int64)
    for qidx, buf, offs in gathered:
        seen[qidx] += 1
        for j, qi in enumerate(qidx.tolist()):
            assert buf[offs[j]:offs[j + 1]].tobytes() == _expected_fetch(oracle, raw, recs, int(ids[qi]), int(st[qi]), int(sp[qi]), int(fl[qi]))
    assert (seen == 1).all()


# ---------------------------------------------------------------------------------- FASTQ over ranks (round 4)
class _OracleFastqShard:
    """What shard.ShardedFastq asks of its staged byte range, answered by the CPU oracle over the whole stream (the kernels'
    part; the collective, the merge of the compositions and the routing are what runs here)."""

    def __init__(self, oracle, raw, lo, hi):
        self.o, self.raw, self.lo, self.hi = oracle, raw, lo, hi
        recs, _, _ = oracle.fastq_index(raw)
        hdr = recs["name_off"] - 1
        self.own = np.nonzero((hdr >= lo) & (hdr < hi))[0]
        self.recs = recs

    def fastq_scan(self):
        core = np.frombuffer(self.raw, dtype=np.uint8)[self.lo:self.hi]
        nl = np.flatnonzero(core == 10)
        return int(nl.size), int(nl[-1] + self.lo) if nl.size else -1

    def fastq_build_ctx(self, line_offset, prev_nl):
        assert line_offset == self.raw[:self.lo].count(b"\n")          # the running line count of the ranks before this one

        class S:
            n_reads = int(self.own.size)
            first_id = int(self.own[0]) if self.own.size else 0
            size = int(self.recs["rlen"][self.own].sum())
        return S

    def fastq_comp(self):
        a = int(self.recs["name_off"][self.own[0]]) - 1
        last = self.own[-1]
        b = int(self.recs["qoff"][last] + self.recs["rlen"][last]) + 2
        c = self.o.fastq_composition(self.raw[a:min(b, len(self.raw))])
        return (np.array([c[k] for k in ("a", "c", "g", "t", "n")], dtype=np.int64),
                np.array([c[k] for k in ("maxlen", "minlen", "minqs", "maxqs", "phred")], dtype=np.int64))

    def fastq_fetch_alloc(self, local_ids, phred=0, seq_flags=0, want=("seq", "qual", "quali")):
        ids = self.own[np.asarray(local_ids, dtype=np.int64)]
        seq = b"".join(self.raw[int(self.recs["soff"][i]):int(self.recs["soff"][i] + self.recs["rlen"][i])] for i in ids)
        qual = b"".join(self.raw[int(self.recs["qoff"][i]):int(self.recs["qoff"][i] + self.recs["rlen"][i])] for i in ids)
        offs = np.zeros(ids.size + 1, dtype=np.int64)
        np.cumsum(self.recs["rlen"][ids], out=offs[1:])
        q = np.frombuffer(qual, dtype=np.uint8)
        return np.frombuffer(seq, dtype=np.uint8), q, (q.astype(np.int16) - (phred or 33)).astype(np.int8), offs

    def close(self):
        pass


def _fastq_worker(rank, world, port, raw, path, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    import fxoracle as oracle
    from pyfastx_amd import _lib, shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _lib.Blob.from_file_range = classmethod(lambda cls, p, off, length, halo=0, device=0: _OracleFastqShard(oracle, raw, off, off + length))
    sq = shard.ShardedFastq(path, rank, world)                          # one all-gather: the cores' line counts
    base, meta = sq.composition()                                        # one more: ten words per rank
    t = torch.tensor([sq.n_local], dtype=torch.int64)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    first = np.concatenate([[0], np.cumsum([int(o) for o in outs])])
    ids = np.random.default_rng(5).integers(0, int(first[-1]), 300)      # the same batch on every rank
    pos, seq, qual, qi, offs = sq.fetch(ids, first, phred=int(meta[4]))
    gathered = [None] * world
    dist.all_gather_object(gathered, (sq.n_local, sq.first_id, base.tolist(), meta.tolist(), pos, seq.tobytes(), qual.tobytes(), offs))
    if rank == 0:
        q.put((ids, gathered))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_fastq_composition_and_fetch_over_gloo(oracle, tmp_path, world):
    """One process per shard of a FASTQ file: the build's all-gather (line counts), the composition's all-gather (five sums, two
    minima, two maxima per rank -> base / meta of the whole file on every rank) and the routed fetch (every read of a batch
    answered by exactly one rank, no bytes between ranks)."""
    rng = np.random.default_rng(world)
    recs_txt = []
    for i in range(400):
        ln = int(rng.integers(1, 120))
        recs_txt.append(b"@r%d some text\n%s\n+\n%s\n" % (i, bytes(rng.choice(list(b"ACGTN"), ln).astype(np.uint8)), bytes(rng.integers(40, 75, ln).astype(np.uint8))))
    raw = b"".join(recs_txt)
    path = str(tmp_path / "g.fq")
    open(path, "wb").write(raw)
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_fastq_worker, args=(r, world, port, raw, path, q)) for r in range(world)]
    for p in procs:
        p.start()
    ids, gathered = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    recs, size, _ = oracle.fastq_index(raw)
    oc = oracle.fastq_composition(raw)
    assert sum(g[0] for g in gathered) == len(recs)
    seen = np.zeros(ids.size, dtype=np.int64)
    for n_local, first_id, base, meta, pos, seq, qual, offs in gathered:
        assert base == [oc[k] for k in ("a", "c", "g", "t", "n")] and meta == [oc[k] for k in ("maxlen", "minlen", "minqs", "maxqs", "phred")]
        seen[pos] += 1
        for j, k in enumerate(pos.tolist()):
            i = int(ids[k])
            s0, q0, l = int(recs["soff"][i]), int(recs["qoff"][i]), int(recs["rlen"][i])
            assert seq[offs[j]:offs[j + 1]] == raw[s0:s0 + l] and qual[offs[j]:offs[j + 1]] == raw[q0:q0 + l]
    assert (seen == 1).all()
