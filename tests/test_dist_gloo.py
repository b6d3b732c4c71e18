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
