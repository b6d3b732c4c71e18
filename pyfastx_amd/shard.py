"""Byte-range sharding of one FASTA stream across GPUs.

Each rank scans only its own byte range [base, base+n) with the HIP kernels
(line table, headers, record table for records that START in the shard), then
ONE all-gather of a fixed 28-word boundary summary (fx_shard_summary) lets every
rank finish the single record of its shard that runs past the cut.  There is
no other data-path exchange: bytes never move between GPUs.

`stitch_tail` is pure integer logic over the gathered summaries and runs on the
host; it is what the world_size-2 gloo tests exercise on CPU.
"""
import json
import os

import numpy as np

FIELDS = ["base", "n_bytes", "is_last", "n_nl", "first_nl", "second_nl", "last_nl", "first_nl_prev",
          "first_byte", "last_byte", "n_hdr", "first_hdr", "last_hdr", "lead_nl", "lead_ws",
          "lead_v1", "lead_c1", "lead_v2", "lead_c2", "tail_e", "tail_first_end", "tail_nl_after", "tail_bad",
          "tail_elen", "tail_dlen", "tail_name_len", "lead_prev_nl", "second_last_nl"]
NWORDS = len(FIELDS)


class Summary(dict):
    __getattr__ = dict.__getitem__

    @classmethod
    def from_array(cls, a):
        return cls(zip(FIELDS, (int(x) for x in a)))

    def to_array(self):
        return np.array([self[k] for k in FIELDS], dtype=np.int64)


def count_ne(u, llen):
    """min(2, number of FULL lead lines of shard u whose len+1 != llen)."""
    full = u.lead_nl - 1
    if full <= 0:
        return 0
    if u.lead_c1 + u.lead_c2 < full:          # >= 3 distinct lengths: at least two differ from any llen
        return 2
    eq = (u.lead_c1 if u.lead_v1 == llen else 0) + (u.lead_c2 if (u.lead_c2 and u.lead_v2 == llen) else 0)
    return min(2, full - eq)


def stream_end(S):
    """`position` after the last line (index.c:231 semantics; virtual EOF newline included)."""
    for u in reversed(S):
        if u.n_nl > 0:
            return u.last_nl + 1
    return 0


def line_regular_rule(boff, blen, slen, llen, elen, norm, byte_at):
    """May slices of this record use the line arithmetic of sequence.c:498-510?  The rule of csrc/fx_kernels.hpp
    (line_regular), restated for the host: `norm` (index.c:342) allows ONE line of another length; the arithmetic is
    right only when that line is the last one.  Decided by the row and one byte of the stream: the record has
    ceil(slen / bpl) lines and the byte in front of its last line is a newline.  byte_at(p): stream[p], or None when
    the caller cannot see it (-> None: undecided)."""
    bpl = llen - elen
    if not norm or bpl <= 0 or elen <= 0:
        return False
    if slen <= bpl:
        return True
    lines = -(-slen // bpl)
    if blen != slen + lines * elen:
        return False
    x = slen - (lines - 1) * bpl
    c = byte_at(boff + blen - (x + elen) - 1)
    return None if c is None else c == 10


def stitch_tail(S, r, full_name=False):
    """Final .fxi columns of the LAST record that starts in shard r, given all
    summaries S (list of Summary in shard order).  None if shard r has no header.
    Mirrors index.c:234-353 for a record whose lines span shard cuts."""
    s = S[r]
    if s.n_hdr == 0:
        return None
    h = s.last_hdr
    e, elen, dlen, name_len = s.tail_e, s.tail_elen, s.tail_dlen, s.tail_name_len
    have_e = e >= 0
    have_llen, llen, nseq, bad = False, 0, 0, 0
    last_nl = s.last_nl
    if have_e:
        nseq = s.tail_nl_after
        if s.tail_first_end >= 0:
            llen, have_llen, bad = s.tail_first_end - e, True, min(2, s.tail_bad)
    hn = None
    t_hdr = -1                                 # the shard that holds the next header line
    ws = -1                                    # whitespace seen in continuation shards while header unterminated
    for t in range(r + 1, len(S)):
        u = S[t]
        if not have_e and ws < 0 and u.lead_ws >= 0:
            ws = u.lead_ws
        if u.lead_nl > 0:
            first = u.first_nl
            full = u.lead_nl - 1
            if not have_e:                     # the header line itself crossed the cut
                pb = u.first_nl_prev if u.first_nl_prev >= 0 else S[t - 1].last_byte
                e = first
                elen = 2 if pb == 13 else 1    # index.c:266-269
                dlen = (e - h) - elen          # index.c:271
                if full_name:
                    name_len = dlen
                elif name_len < 0:
                    name_len = (ws - (h + 1)) if (0 <= ws < e) else dlen
                name_len = min(name_len, dlen)
                have_e = True
                if full >= 1:                  # first full lead line = first sequence line of the record
                    llen, have_llen = u.second_nl - u.first_nl, True
                    bad += count_ne(u, llen)
                nseq += full
            elif not have_llen:                # the first sequence line crossed the cut
                llen, have_llen = first - e, True
                nseq += u.lead_nl
                bad += count_ne(u, llen)
            else:                              # an ordinary line crossed the cut
                bad += int(first - last_nl != llen)
                nseq += u.lead_nl
                bad += count_ne(u, llen)
            bad = min(bad, 2)
            last_nl = u.last_nl if u.n_hdr == 0 else last_nl
        if u.n_hdr > 0:
            hn, t_hdr = u.first_hdr, t
            break
    if hn is None:
        hn = stream_end(S)
    boff = e + 1
    blen = hn - boff                           # index.c:243,348
    slen = blen - elen * nseq
    if nseq <= 0:
        llen = 0
    # line-regular (line_regular_rule): the deciding byte may sit on another rank, so the same question is put to the
    # summaries -- is the record's last line (between the two newlines before hn) x + elen long?
    reg, bpl = 0, llen - elen
    if bad <= 1 and bpl > 0 and elen > 0:
        if bad == 0 or slen <= bpl:
            reg = 1
        else:
            lines = -(-slen // bpl)
            if lines == nseq:
                x = slen - (lines - 1) * bpl
                found, prev = 0, -1
                for t in range(t_hdr if t_hdr >= 0 else len(S) - 1, r - 1, -1):
                    u = S[t]
                    lead = t == t_hdr          # only the newlines before its first header line count there
                    cnt = u.lead_nl if lead else u.n_nl
                    if cnt <= 0:
                        continue
                    if found == 0:
                        if cnt >= 2:
                            prev, found = (u.lead_prev_nl if lead else u.second_last_nl), 2
                        else:
                            found = 1
                    else:
                        prev, found = (u.first_hdr - 1 if lead else u.last_nl), 2
                    if found == 2:
                        break
                reg = int(found == 2 and (hn - 1) - prev == x + elen)
    return {"boff": boff, "blen": blen, "slen": slen, "llen": llen,
            "elen": elen, "norm": 0 if bad > 1 else 1, "dlen": dlen, "name_len": name_len, "reg": reg}


def id_offsets(S):
    out, acc = [], 0
    for u in S:
        out.append(acc)
        acc += u.n_hdr
    return out, acc


def allgather_summaries(mine, world, device="cpu"):
    """THE collective of the sharded build: one all-gather of the fixed-size boundary
    summary (28 x int64 per rank; RCCL over xGMI on GPUs, gloo in the CPU tests)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(mine.to_array()).to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return [Summary.from_array(o.cpu().numpy()) for o in outs]


def pmc_traffic(root, file_bytes=None):
    """HBM bytes per k_span_scan launch from the committed rocprofv3 --pmc summary; None when there is no
    summary or it was collected on a different workload size."""
    p = os.path.join(root, "profiles", "pmc_k_span_scan.json")
    try:
        with open(p) as f:
            j = json.load(f)
        if file_bytes is not None and int(j.get("file_bytes", -1)) != int(file_bytes):
            return None
        return j.get("hbm_bytes_per_launch")
    except Exception:
        return None


class ShardedFasta:
    """One rank's share of a sharded FASTA index build (one process per GPU).

    `piece` is this rank's generated piece of the concatenated stream, already in
    HBM.  For world > 1 the cut is moved DELTA bytes into the next piece (so that
    a record really straddles every boundary): each rank ships its first DELTA
    bytes to the previous rank once, at setup, outside any timed region."""

    DELTA = 100_003

    def __init__(self, piece, nbytes, dev, rank, world, full_name=False, logical=None):
        """logical: None = one process per GPU (torch.distributed); else {"sizes": [...], "next_head": tensor or None}
        -- the same shard built without a process group (G logical shards on one GPU)."""
        import torch
        import torch.distributed as dist
        from . import _lib
        self.rank, self.world, self.dev, self.full_name = rank, world, dev, full_name
        self._torch, self._dist = torch, dist
        self.comm_dev = dev
        if world == 1:
            self.buf, self.n_bytes, self.base, prev, last = piece, nbytes, 0, 10, True
        else:
            d = self.DELTA
            if logical is not None:
                sizes = np.asarray(logical["sizes"], dtype=np.int64)
                tail_in = logical["next_head"]
            else:
                # collectives run on the GPU tensors with RCCL ("nccl"); with gloo (tests) on host copies
                cdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
                self.comm_dev = cdev
                mine = torch.tensor([nbytes], dtype=torch.int64, device=cdev)
                outs = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(outs, mine)
                sizes = np.array([int(o.item()) for o in outs], dtype=np.int64)
                head = piece[:d].to(cdev).clone()
                tail_c = torch.empty(d, dtype=torch.uint8, device=cdev)
                ops = []
                if rank > 0:
                    ops.append(dist.P2POp(dist.isend, head, rank - 1))
                if rank < world - 1:
                    ops.append(dist.P2POp(dist.irecv, tail_c, rank + 1))
                for w in dist.batch_isend_irecv(ops) if ops else []:
                    w.wait()
                tail_in = tail_c.to(dev)
            lo = d if rank > 0 else 0
            hi_extra = d if rank < world - 1 else 0
            n = nbytes - lo + hi_extra
            buf = torch.zeros(n + 131072, dtype=torch.uint8, device=dev)
            buf[:nbytes - lo] = piece[lo:nbytes]
            if hi_extra:
                buf[nbytes - lo:n] = tail_in[:d]
            prev = int(piece[lo - 1]) if lo else 10
            self.buf, self.n_bytes = buf, n
            self.base = int(sizes[:rank].sum()) + lo
            last = rank == world - 1
        torch.cuda.synchronize()
        self.blob = _lib.Blob.from_device(self.buf.data_ptr(), self.n_bytes, device=dev.index, keepalive=self.buf)
        self.blob.set_shard(self.base, prev, last)
        self._init_state()

    def _init_state(self):
        self.summary = None
        self.S = None
        self.n_local = 0
        self._ext = self._mine = self._all = None
        self.force_collective = getattr(self, "force_collective", False)
        self.fxcomm = None

    def _collective(self):
        return self.world > 1 or self.force_collective

    def _use_fxcomm(self):
        """On GPUs the all-gather of the build is the LIBRARY's (fx_comm: ncclAllGather on the handle's own stream, the
        whole build one call, fx_fasta_build_sharded_begin) -- the same entry a C caller uses; torch's
        process group only carries the communicator's id at setup.  FX_COMM=torch keeps the all-gather in
        torch.distributed (all_gather_into_tensor on torch's stream, ordered against the handle's with stream events)."""
        if not self._collective() or self.comm_dev.type != "cuda" or os.environ.get("FX_COMM", "fx") == "torch":
            return False
        if self.fxcomm is None and not getattr(self, "_fxcomm_failed", False):
            from . import _lib
            comm, err = None, ""
            try:
                comm = _lib.Comm.from_process_group(self.dev.index)
            except Exception as e:                            # noqa: BLE001  (no RCCL to bind, ncclCommInitRank failed, ...)
                err = str(e)
            # all ranks take the same path: the library's communicator only if EVERY rank got one
            ok = self._torch.tensor([1 if comm is not None else 0], dtype=self._torch.int32, device=self.comm_dev)
            self._dist.all_reduce(ok, op=self._dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                self.fxcomm = comm
            else:
                if comm is not None:
                    comm.close()
                self._fxcomm_failed = True
                self.fxcomm_error = err or "another rank could not create its communicator"
        return self.fxcomm is not None

    @classmethod
    def from_file(cls, path, dev, rank, world, full_name=False, force_collective=False):
        """This rank's byte range of a FILE ("each GPU ingests only its own range from the host"): the
        uncompressed stream is cut into `world` equal ranges, rank r reads (plain) or reads-and-inflates (BGZF) only
        [size * r / world, size * (r + 1) / world) -- fx_open_file_range -- and nothing moves between GPUs.
        force_collective: run the all-gather + device stitch even with world == 1 (exercises the RCCL path on one GPU)."""
        import torch
        import torch.distributed as dist
        from . import _lib
        size, kind = _lib.stream_size(path)
        if kind == 2:
            raise ValueError("%s is a single gzip stream: it does not shard by byte range (replicas only)" % path)
        lo, hi = size * rank // world, size * (rank + 1) // world
        self = cls.__new__(cls)
        self.rank, self.world, self.dev, self.full_name = rank, world, dev, full_name
        self._torch, self._dist = torch, dist
        self.comm_dev = dev
        if world > 1 or force_collective:
            self.comm_dev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
        self.path, self.stream_bytes = path, size
        self.buf = None
        self.base, self.n_bytes = lo, hi - lo
        self.blob = _lib.Blob.from_file_range(path, lo, hi - lo, 0, device=dev.index)
        self.force_collective = bool(force_collective)
        self._init_state()
        return self

    def _dev_buffers(self):
        torch = self._torch
        if self._ext is None:
            self._ext = torch.cuda.ExternalStream(self.blob.stream, device=self.dev)
            self._mine = torch.zeros(NWORDS, dtype=torch.int64, device=self.dev)
            self._all = torch.zeros(self.world * NWORDS, dtype=torch.int64, device=self.dev)

    def build_begin(self):
        """Enqueue: local scan + tables (+ this shard's boundary summary into the send buffer).  No host sync."""
        self.blob.fasta_build_begin(self.full_name)
        if self.world > 1 or self.force_collective:
            self._dev_buffers()
            self.blob.shard_summary_dev(self._mine.data_ptr())

    def build_end(self):
        """Enqueue, after self._all holds every shard's summary: finish the record that crosses the cut."""
        if self.world > 1 or self.force_collective:
            self.blob.fasta_stitch_dev(self._all.data_ptr(), self.world, self.rank, self.full_name)

    def build_async(self):
        """The whole sharded build enqueued on the device -- scan, summary kernel, RCCL all-gather, stitch kernel,
        ordered with stream events only -- so that device-side consumers (fetch_local) can follow without a host
        round trip; finish() is the one synchronisation.  (gloo, i.e. the CPU tests: the host path, synchronous.)"""
        if (self.world > 1 or self.force_collective) and self.comm_dev.type != "cuda":
            return self.build()
        if self._use_fxcomm():
            return self.blob.fasta_build_sharded_begin(self.fxcomm, self.full_name)       # the ONE collective, under the C ABI
        self.build_begin()
        if self.world > 1 or self.force_collective:
            cur = self._torch.cuda.current_stream(self.dev)
            cur.wait_stream(self._ext)
            self._dist.all_gather_into_tensor(self._all, self._mine)          # the ONE collective (RCCL over xGMI)
            self._ext.wait_stream(cur)
            self.build_end()

    def finish(self):
        s = self.blob.fasta_build_end()
        self.n_local = s.n_seq
        return s

    def build(self):
        if (self.world == 1 and not self.force_collective) or self.comm_dev.type == "cuda":
            self.build_async()
            return self.finish()
        s = self.blob.fasta_build(self.full_name)
        self.n_local = s.n_seq
        self.S = allgather_summaries(self.blob.shard_summary(), self.world, self.comm_dev)   # gloo (CPU tests): host path
        row = stitch_tail(self.S, self.rank, self.full_name)
        if row is not None:
            self.blob.fasta_set_row(self.n_local - 1, **row)
        return s

    def gather_index(self):
        """Every rank's rows and names on every rank (all_gather_object: setup, not the data path) -> the table of the
        WHOLE file in file order: dict of numpy columns + `names` (list of bytes), `seq_len`, `bases`, `ends`.  What rank 0
        writes into the one .fxi (write_index) and what ShardFetcher routes by."""
        part = local_index_part(self.blob, self.n_local, self.base, self.n_bytes)
        if self.world == 1 and not self.force_collective:
            outs = [part]
        else:
            outs = [None] * self.world
            self._dist.all_gather_object(outs, part)
        return merge_index_parts(outs)

    def write_index(self, index_file, table=None):
        """ONE .fxi for the whole file from the gathered table (rank 0 writes; the other ranks return after the gather):
        the same schema and rows a single-GPU build of the file writes."""
        from . import fxi
        table = self.gather_index() if table is None else table
        if self.rank == 0:
            write_merged_index(index_file, table)
        if self.world > 1 or self.force_collective:
            self._dist.barrier()
        return table

    def fetch_local(self, n, d_ids, d_st, d_sp, d_fl, d_out, d_off, d_len):
        self.blob.fasta_fetch_dev(n, d_ids.data_ptr(), d_st.data_ptr(), d_sp.data_ptr(), d_out.data_ptr(),
                                  d_off.data_ptr(), flags_per_query=d_fl.data_ptr(), out_len=d_len.data_ptr())

    def sync(self):
        self.blob.sync()

    def local_rows(self):
        return self.blob.fasta_table(self.n_local)

    def composition(self):
        """Per-record 128-bin composition (fasta.c:901-950) of the records this rank owns, complete across the cuts:
        every rank counts its own bytes once (k_fasta_comp), two more small all-gathers move (a) the boff of each
        shard's last record -- where the next shards' leading bytes start to count -- and (b) the 128 counts of
        the bytes before each shard's first header line to the rank that owns that record."""
        torch, dist = self._torch, self._dist
        n = self.n_local
        if self.world == 1 and not self.force_collective:
            return self.blob.fasta_comp(n)
        rows = self.local_rows()
        mine = torch.tensor([self.base, int(rows["boff"][-1]) if n else -1, n], dtype=torch.int64, device=self.comm_dev)
        allv = [torch.zeros(3, dtype=torch.int64, device=self.comm_dev) for _ in range(self.world)]
        dist.all_gather(allv, mine)
        info = [[int(x) for x in v.cpu()] for v in allv]
        bases, boffs, nh = [v[0] for v in info], [v[1] for v in info], [v[2] for v in info]
        comp, lead = self.blob.fasta_comp_shard(n, comp_lead_from(bases, boffs, self.rank))
        leads = [torch.zeros(128, dtype=torch.int64, device=self.comm_dev) for _ in range(self.world)]
        dist.all_gather(leads, torch.from_numpy(lead).to(self.comm_dev))
        return comp_fold_leads(comp, [v.cpu().numpy() for v in leads], nh, self.rank)

    def fetcher(self, table=None):
        """ShardFetcher over all ranks' shards (SURVEY 8e "Fetch"): the global record table and every shard's byte range
        are collected once (all_gather_object of the small per-rank tables -- setup, not the data path); afterwards
        every rank answers, from its own HBM, the queries whose first byte it holds, and only pieces of queries that
        cross a cut are exchanged.  table: what gather_index returned, when the caller has it already."""
        if table is not None:
            return ShardFetcher({self.rank: self.blob}, table["bases"], table["ends"], table,
                                exchange=allgather_pieces if (self.world > 1 or self.force_collective) else None)
        if self.world == 1:
            t = self.local_rows()
            t["reg"] = self.blob.fasta_line_regular(self.n_local)
            return ShardFetcher({0: self.blob}, [self.base], [self.base + self.n_bytes], t)
        cols = ("boff", "blen", "slen", "llen", "elen", "norm", "reg")
        rows = self.local_rows()
        rows["reg"] = self.blob.fasta_line_regular(self.n_local)
        outs = [None] * self.world
        self._dist.all_gather_object(outs, (self.base, self.n_bytes, {c: np.asarray(rows[c]) for c in cols}))
        table = {c: np.concatenate([o[2][c] for o in outs]) for c in cols}
        return ShardFetcher({self.rank: self.blob}, [o[0] for o in outs], [o[0] + o[1] for o in outs], table,
                            exchange=allgather_pieces)

    def check_against_plan(self, plan, rows, next_plan=None):
        """Analytic ground truth of the generator vs the rows this rank owns."""
        d = self.DELTA if self.world > 1 else 0
        # records of this piece whose '>' lies in this shard: all for world == 1; for world > 1
        # piece-record 0 starts in the previous rank's shard (rank > 0), and the next piece's
        # record 0 starts in ours (rank < world-1) -- checked by its owner's neighbour below.
        first = 1 if (self.world > 1 and self.rank > 0) else 0
        npiece = len(plan["slen"]) - first
        shift = self.base - (d if self.rank > 0 else 0)      # global offset of this piece's byte 0
        ok = True
        for k in ("hoff", "boff"):
            ok &= bool((rows[k][:npiece] == plan[k][first:] + shift).all())
        for k in ("blen", "slen", "llen", "dlen", "name_len"):
            ok &= bool((rows[k][:npiece] == plan[k][first:]).all())
        ok &= bool((rows["norm"][:npiece] == 1).all()) and bool((rows["elen"][:npiece] == 1).all())
        if next_plan is not None:              # the stitched row: first contig of the NEXT piece starts in our shard
            ok &= len(rows["boff"]) == npiece + 1
            nshift = shift + int(plan["n_bytes"])
            j = npiece
            ok &= int(rows["hoff"][j]) == int(next_plan["hoff"][0]) + nshift
            ok &= int(rows["boff"][j]) == int(next_plan["boff"][0]) + nshift
            for k in ("blen", "slen", "llen", "dlen", "name_len"):
                ok &= int(rows[k][j]) == int(next_plan[k][0])
            ok &= int(rows["norm"][j]) == 1 and int(rows["elen"][j]) == 1
        else:
            ok &= len(rows["boff"]) == npiece
        return bool(ok)


# ------------------------------------------------------------------ one index file from many shards
HEAD_BYTES = 4096                  # what a rank shows of its first bytes at least (a name cut by a shard boundary runs on in them)
HEAD_MAX = 64 << 20                # ... and at most: it shows them up to its first newline, where such a name ends at the latest


def local_index_part(blob, n_local, base, n_bytes):
    """What one shard contributes to the index of the whole file: its rows (stitched), line-regular bits, the names it can
    see (the last one may be cut by the shard's end) and its first bytes (where such a name of the previous shard ends)."""
    rows = blob.fasta_table(n_local)
    rows["reg"] = blob.fasta_line_regular(n_local)
    names = []
    if n_local:
        ln = np.maximum(rows["name_len"].astype(np.int64), 0)
        nb, no, ol = blob.fetch_ranges(rows["hoff"] + 1, ln, ln, flags=8)       # FX_RAW; clamped to the bytes held
        raw, o, l = nb.tobytes(), no.tolist(), ol.tolist()
        names = [raw[o[i]:o[i] + l[i]] for i in range(n_local)]
    S = blob.shard_summary()
    first_line = (S.first_nl - base + 1) if S.first_nl >= 0 else n_bytes
    head = blob.read_bytes(base, min(max(HEAD_BYTES, min(first_line, HEAD_MAX)), n_bytes))
    return (int(base), int(n_bytes), {k: np.asarray(v) for k, v in rows.items()}, names, head)


def merge_index_parts(parts):
    """Parts of all shards in shard order -> the table of the whole file (see ShardedFasta.gather_index)."""
    names = []
    for r, (base, nb, rows, nm, _) in enumerate(parts):
        nm = list(nm)
        if nm:
            need = int(max(rows["name_len"][-1], 0))
            t = r + 1
            while len(nm[-1]) < need and t < len(parts):       # the name runs on in the next shard(s)
                head = parts[t][4]
                nm[-1] = nm[-1] + head[:need - len(nm[-1])]
                if parts[t][1] > len(head):
                    break
                t += 1
        names.extend(nm)
    cols = list(parts[0][2])
    table = {c: np.concatenate([p[2][c] for p in parts]) for c in cols}
    table["names"] = names
    table["seq_len"] = int(np.maximum(table["slen"], 0).sum())
    table["bases"] = [p[0] for p in parts]
    table["ends"] = [p[0] + p[1] for p in parts]
    return table


def write_merged_index(index_file, table, key_func=None):
    """The .fxi of the whole file from the merged table (index.c:170-224, 342-372 through fxi.py)."""
    import os
    from . import fxi
    names = table["names"]
    if key_func is not None:
        raise ValueError("key_func needs the whole header line: build on one device")
    if index_file != ":memory:" and not os.path.exists(index_file) and len(names):
        offs = np.zeros(len(names) + 1, dtype=np.int64)
        np.cumsum([len(x) for x in names], out=offs[1:])
        packed = np.frombuffer(b"".join(names), dtype=np.uint8) if offs[-1] else np.zeros(0, dtype=np.uint8)
        try:
            return fxi.write_fasta_bulk(index_file, packed, offs, table, table["seq_len"], order=None)
        except Exception:
            if os.path.exists(index_file):
                os.remove(index_file)
    db = fxi.connect(index_file)
    fxi.write_fasta(db, names, table, table["seq_len"])
    return db


# ------------------------------------------------------------------ composition across shards
def comp_lead_from(bases, last_boffs, r):
    """Global offset from which the bytes before shard r's first header line are counted: they belong to the last
    record of the nearest earlier shard that holds a header line, and start no earlier than that record's boff (its
    header line may cross the cut).  bases[t]: first byte of shard t; last_boffs[t]: boff of the last record that
    starts in shard t AFTER stitching, -1 if none.  -1: nothing before shard r owns them (fasta.c:901-950 drops
    the bytes before the first header)."""
    for t in range(r - 1, -1, -1):
        if last_boffs[t] >= 0:
            return max(int(bases[r]), int(last_boffs[t]))
    return -1


def comp_fold_leads(comp, leads, n_hdrs, r):
    """Add to the last row of shard r's composition the lead rows of the following shards, up to and including the
    first one that holds a header line (whose lead is what precedes that header).  In place; comp may be empty."""
    if n_hdrs[r] == 0:
        return comp
    for t in range(r + 1, len(leads)):
        comp[-1] += leads[t]
        if n_hdrs[t] > 0:
            break
    return comp


# ------------------------------------------------------------------ FASTQ shards
def fastq_contexts(cores):
    """cores[r] = (newlines in shard r's core, offset of the last one or -1), in shard order
    (the payload of the one all-gather).  -> [(line_offset, prev_nl)] for fx_fastq_build_ctx."""
    out, loff, prev = [], 0, -1
    for n_nl, last in cores:
        out.append((loff, prev))
        loff += n_nl
        if n_nl > 0:
            prev = last
    return out


def allgather_fastq_cores(mine, world, device="cpu"):
    """One all-gather of two int64 per rank (gloo in tests, RCCL on GPUs)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor(list(mine), dtype=torch.int64, device=device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return [tuple(int(v) for v in o.cpu()) for o in outs]


# ------------------------------------------------------------------ fetch over byte-range shards ("Fetch")
# The resident stream is range-partitioned across the GPUs, so a query is answered by the rank that holds its bytes.
# Host side, per batch: (record, start, stop) -> byte range of the global stream (the reference's line arithmetic),
# ranges -> pieces per shard (vectorised; almost every query is one piece), every rank fetches the pieces inside its
# own range with the ordinary fetch kernel (global offsets: the kernel clamps to the bytes it holds), and only the
# handful of queries that cross a cut exchange their pieces (one all_gather_object of a few hundred bytes; nothing
# when no query crosses).  A query is ANSWERED by the rank that holds its first byte.
F_UP, F_REV, F_COMP = 1, 2, 4


def slice_ranges(table, ids, starts, stops):
    """(record id, 0-based [start, stop)) -> off, blen, skip, take over the global stream.  Line-regular records
    (table["reg"], fx_fasta_line_regular; a table without that column: norm = 1): exactly the bytes,
    sequence.c:498-510.  Others: the whole record, sliced after despacing (sequence.c:100-110) -- skip = start."""
    ids = np.asarray(ids, dtype=np.int64)
    a = np.asarray(starts, dtype=np.int64)
    b = np.asarray(stops, dtype=np.int64)
    boff, blen = np.asarray(table["boff"], dtype=np.int64)[ids], np.asarray(table["blen"], dtype=np.int64)[ids]
    llen, elen = np.asarray(table["llen"], dtype=np.int64)[ids], np.asarray(table["elen"], dtype=np.int64)[ids]
    bpl = llen - elen
    reg = (np.asarray(table["reg"] if "reg" in table else table["norm"])[ids] != 0) & (bpl > 0)
    safe = np.where(bpl > 0, bpl, 1)
    bs, be = a // safe, b // safe
    off = np.where(reg, boff + a + elen * bs, boff)
    ln = np.where(reg, (b - a) + (be - bs) * elen, blen)
    return off, ln, np.where(reg, 0, a), b - a


def route_ranges(bases, ends, off, blen):
    """Byte ranges [off, off+blen) of the global stream -> their pieces per shard [bases[r], ends[r]).
    -> dict: q (query of each piece), r (shard), poff, plen, cnt (pieces per query), first (shard of the first byte);
    pieces are ordered by (q, r).  Bytes past the end of the stream belong to nobody (fread semantics, index.c:689)."""
    bases, ends = np.asarray(bases, dtype=np.int64), np.asarray(ends, dtype=np.int64)
    off, blen = np.asarray(off, dtype=np.int64), np.asarray(blen, dtype=np.int64)
    n = off.size
    stop = np.minimum(off + np.maximum(blen, 0), ends[-1] if ends.size else 0)
    live = stop > off
    first = np.clip(np.searchsorted(bases, off, "right") - 1, 0, max(bases.size - 1, 0))
    last = np.clip(np.searchsorted(bases, stop - 1, "right") - 1, 0, max(bases.size - 1, 0))
    cnt = np.where(live, last - first + 1, 0).astype(np.int64)
    start = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(cnt, out=start[1:])
    q = np.repeat(np.arange(n, dtype=np.int64), cnt)
    r = first[q] + (np.arange(int(start[-1]), dtype=np.int64) - start[q])
    poff = np.maximum(off[q], bases[r])
    plen = np.minimum(stop[q], ends[r]) - poff
    return {"q": q, "r": r, "poff": poff, "plen": plen, "cnt": cnt, "first": first, "start": start}


def _finish(pieces, skip, take, flags):
    """Pieces of one query (despaced, upper-cased / complemented by the kernel, in stream order) -> its answer:
    slice after despacing, then reverse (util.c:239-269 order: the complement is bytewise, the reversal is not)."""
    s = b"".join(pieces)[skip:skip + take]
    return s[::-1] if flags & F_REV else s


class ShardFetcher:
    """Batched fetch over the byte-range shards of one stream.

    fetchers: {shard index: object with Blob.fetch_ranges} for the shards THIS process holds (one entry per rank in
    the one-process-per-GPU run; all of them for G logical shards on one GPU); bases/ends: every shard's byte range;
    table: the global record table (rows concatenated in shard order).  exchange: None, or a callable that takes this
    process's cross-cut pieces and returns everybody's (all_gather_object over the process group)."""

    def __init__(self, fetchers, bases, ends, table, exchange=None):
        self.f, self.table, self.exchange = dict(fetchers), table, exchange
        self.bases, self.ends = np.asarray(bases, dtype=np.int64), np.asarray(ends, dtype=np.int64)
        reg = table["reg"] if "reg" in table else table["norm"]
        self._cols = {k: np.ascontiguousarray(table[k], dtype=np.int64) for k in ("boff", "blen", "llen", "elen")}
        self._cols["reg"] = np.ascontiguousarray(np.asarray(reg) != 0, dtype=np.uint8)

    def fetch(self, ids, starts, stops, flags=0, flags_per_query=None):
        """-> (qidx, buf, offs): the queries this process answers (those whose first byte it holds; ordered by shard,
        then by position in the batch), their bases back to back, offsets[len(qidx)+1]."""
        from . import _lib
        n = len(ids)
        # line arithmetic, owning shard and routed order of the whole batch: one pass in the library (fx_shard_route)
        R = _lib.shard_route(ids, starts, stops, self._cols, self.bases, self.ends, flags, flags_per_query)
        order, ss, cnt = R["order"], R["shard_start"], R["cnt"]
        answers = {}
        loose = []                                            # (query, shard, bytes): pieces of queries that cross a cut
        for sh, fx in self.f.items():
            a, b = int(ss[sh]), int(ss[sh + 1])
            if a == b:
                continue
            seg = slice(a, b)
            whole = cnt[seg] == 1                             # one piece: the kernel does it all (slice after despacing included); 0: nothing to read
            if whole.all():
                off, ln, tk, kfl, ksk, qq = R["off"][seg], R["len"][seg], R["take"][seg], R["fl"][seg], R["skip"][seg], order[seg]
            else:
                k = a + np.nonzero(whole)[0]
                off, ln, tk, kfl, ksk, qq = R["off"][k], R["len"][k], R["take"][k], R["fl"][k], R["skip"][k], order[k]
            buf, offs, ol = fx.fetch_ranges(off, ln, tk, flags_per_query=kfl, skip=ksk if ksk.any() else None)
            answers[sh] = (qq, buf, offs[:-1] if offs.size == qq.size + 1 else offs, ol)
        # the few queries that cross a cut: their pieces per shard, despaced by the shard that holds them
        cross = np.nonzero(cnt > 1)[0]
        if cross.size:
            P = route_ranges(self.bases, self.ends, R["off"][cross], R["len"][cross])
            for k in range(P["q"].size):
                sh = int(P["r"][k])
                if sh in self.f:
                    j = int(cross[P["q"][k]])
                    buf, offs, ol = self.f[sh].fetch_ranges([P["poff"][k]], [P["plen"][k]], [P["plen"][k]],
                                                            flags=int(R["fl"][j]) & (F_UP | F_COMP))
                    loose.append((int(order[j]), sh, buf[:int(ol[0])].tobytes()))
        skip, take, fl = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.uint8)
        if cross.size:
            skip[order[cross]], take[order[cross]], fl[order[cross]] = R["skip"][cross], R["take"][cross], R["fl"][cross]
        everybody = self.exchange(loose) if self.exchange is not None else loose
        by_q = {}
        for qi, sh, bts in sorted(everybody):
            by_q.setdefault(qi, []).append(bts)
        # ---- what this process answers, in query order: runs of kernel output + the few answers put together here
        own = sorted(self.f.keys())
        mine_q = np.concatenate([order[int(ss[sh]):int(ss[sh + 1])] for sh in own]) if own else np.zeros(0, dtype=np.int64)   # shard by shard: the kernels' outputs stay whole
        m = mine_q.size
        pos = np.full(n, -1, dtype=np.int64)
        pos[mine_q] = np.arange(m, dtype=np.int64)
        bufs, src = [], np.full(m, -1, dtype=np.int64)
        so, sl = np.zeros(m, dtype=np.int64), np.zeros(m, dtype=np.int64)
        for sh, (qq, buf, o, ol) in answers.items():
            pq = pos[qq]
            src[pq], so[pq], sl[pq] = len(bufs), o, ol
            bufs.append(buf)
        for j in np.nonzero(src < 0)[0].tolist():
            qi = int(mine_q[j])
            ans = _finish(by_q.get(qi, []), int(skip[qi]), int(take[qi]), int(fl[qi]))
            src[j], sl[j] = len(bufs), len(ans)
            bufs.append(np.frombuffer(ans, dtype=np.uint8))
        offs = np.zeros(m + 1, dtype=np.int64)
        np.cumsum(sl, out=offs[1:])
        if m == 0:
            return mine_q, np.zeros(0, dtype=np.uint8), offs
        brk = np.nonzero((src[1:] != src[:-1]) | (so[1:] != so[:-1] + sl[:-1]))[0] + 1
        lo = np.concatenate([[0], brk])
        hi = np.concatenate([brk, [m]])
        if lo.size == 1:                                     # the usual batch: the kernel's output is the answer
            return mine_q, bufs[int(src[0])][int(so[0]):int(so[-1] + sl[-1])], offs
        runs = [bufs[int(src[a])][int(so[a]):int(so[e - 1] + sl[e - 1])] for a, e in zip(lo.tolist(), hi.tolist())]
        return mine_q, np.concatenate(runs), offs


def allgather_pieces(pieces):
    """exchange= for ShardFetcher in the one-process-per-GPU run: every rank's cross-cut pieces to every rank (a few
    hundred bytes per cut; the only exchange of the fetch path, and only for queries that cross a cut)."""
    import torch.distributed as dist
    outs = [None] * dist.get_world_size()
    dist.all_gather_object(outs, pieces)
    return [p for o in outs for p in o]


def write_fastq_index_parts(parts, index_file, device=None):
    """ONE .fxi from the ShardedFastq objects of ALL ranks of a build, held by one process (logical ranks on one or several
    devices): the same pages as ShardedFastq.write_index writes over a process group, without one -- every part's leaves are
    formatted on its own device, the names meet on the first part's.  -> rows written."""
    from . import fxi
    parts = sorted(parts, key=lambda p: p.rank)
    w = fxi.PartsWriter(index_file, 1, parts[0].device if device is None else device)
    shapes = []
    for p in parts:
        w.add_local(p.blob, device=p.device)
        shapes.append(p.size)
    db = w.finish()
    n, size = w.rows, int(sum(shapes))
    db.execute("INSERT INTO stat VALUES (?,?,?)", (n, size, size * 1.0 / n if n else float("nan")))     # fastq.c:161
    db.commit()
    db.close()
    return n


# ------------------------------------------------------------------ FASTQ: one file over several ranks (SURVEY 8e "FASTQ")
class ShardedFastq:
    """One rank's share of a sharded FASTQ index build (pyfastx_fastq_create_index, fastq.c:8-182, one process per GPU).

    Rank r stages bytes [size r / R, size (r + 1) / R) of the uncompressed stream plus a halo behind them: a record
    belongs to the rank its HEADER line begins in and is finished from the halo.  The only exchange of the build is the
    line numbering -- one all-gather of two integers per rank (newlines in the core, offset of the last one).  A halo
    that turns out too small for a record of this shard (FX_ERANGE from fx_fastq_build_ctx: a read of a megabyte) is no
    error: the range is opened again with a larger one, the numbering from the all-gather stays valid (it only counts
    the cores).  gather: callable(np.int64[2]) -> int64[world, 2]; default: torch.distributed's all_gather (gloo in the
    CPU tests, RCCL on GPUs) -- or fx_comm (Comm.allgather), the library's own."""

    HALO0 = 1 << 16

    def __init__(self, path, rank, world, device=0, halo=None, gather=None, index_file=None):
        from . import _lib
        size, kind = _lib.stream_size(path)
        if kind == 2:
            raise ValueError("%s is a single gzip stream: it does not shard by byte range (replicas only)" % path)
        self.path, self.rank, self.world, self.device, self.stream_bytes = path, rank, world, device, size
        # index_file: where write_index will put the .fxi -- rank 0 creates it NOW and lets it grow to its estimated size in a
        # thread of the library while the ranks stage and scan (fxi.presize_fastq, as api.Fastq does for one device): the
        # pages are allocated by the time the first leaf is formatted.  Plain files of 1 GiB and more only; best effort.
        self._presized = None
        if index_file and rank == 0 and kind == 0 and not os.path.exists(index_file) and not os.environ.get("FX_FXI_NO_PRESIZE"):
            from . import fxi
            try:
                tok = fxi.presize_fastq(index_file, path, device=device)
                if tok is not None:
                    self._presized = (index_file, tok)
            except Exception:                                 # noqa: BLE001  (no early file: write_index makes it)
                if os.path.exists(index_file):
                    os.remove(index_file)
        self.base, self.end = size * rank // world, size * (rank + 1) // world
        self.halo = int(self.HALO0 if halo is None else halo)
        self.reopened = 0
        ctx = None
        while True:
            h = min(self.halo, size - self.end)
            blob = _lib.Blob.from_file_range(path, self.base, self.end - self.base, h, device=device)
            core = blob.fastq_scan()
            if ctx is None:
                mine = np.array(core, dtype=np.int64)
                if gather is None:
                    allc = allgather_fastq_cores(tuple(int(x) for x in mine), world, "cpu") if world > 1 else [tuple(int(x) for x in mine)]
                else:
                    allc = [tuple(int(v) for v in row) for row in np.asarray(gather(mine)).reshape(world, 2)]
                ctx = fastq_contexts(allc)[rank]
            try:
                self.summary = blob.fastq_build_ctx(*ctx)
                break
            except _lib.FxError as e:
                if e.code != _lib.FX_ERANGE or h >= size - self.end:
                    raise
                blob.close()
                self.halo *= 8
                self.reopened += 1
        self.blob = blob
        self.n_local, self.first_id, self.size = int(self.summary.n_reads), int(self.summary.first_id), int(self.summary.size)

    def local_part(self):
        """This shard's rows (global offsets) and names, as plain arrays (the host loaders' fall-back)."""
        t = self.blob.fastq_table(self.n_local)
        packed, offs = self.blob.names_pack(1, self.n_local, guess=int(np.maximum(t["name_len"], 0).sum()))
        return {"dlen": np.asarray(t["dlen"], np.int64), "rlen": np.asarray(t["rlen"], np.int64), "soff": np.asarray(t["soff"], np.int64),
                "qoff": np.asarray(t["qoff"], np.int64), "names": np.asarray(packed, np.uint8), "name_off": np.asarray(offs, np.int64),
                "size": np.array([self.size, self.n_local], np.int64)}

    def write_index(self, index_file, gather=None, group=None):
        """ONE .fxi for the whole file, every page of its two big b-trees formatted on a device (fxi.PartsWriter, round 6; round 5
        sent every rank's table and names to rank 0 through files and let the host page loader format them: 15 M rows/s).
        Collective over the ranks of the build:
          1. fx_fxi_part_shape on every rank; ONE all-gather of five integers per rank (rows, table leaves, bytes of names, an
             overflow flag, bases);
          2. rank 0 creates the database and, in a thread, makes room for all its pages (table + estimated index);
          3. every rank posts its names to rank 0 (point to point: RCCL over xGMI on GPUs, gloo through the host in tests);
             an all-gather of three words tells everybody where the new pages begin, and every rank formats ITS table
             leaves and copies them into its page range of the file while the names travel;
          4. the first rows of every rank's leaves follow (8 bytes per leaf; their arrival also says the leaves are in
             the file), rank 0 writes the interior levels, the index and the header.
        A row that needs an overflow page (FX_ERANGE on any rank): the host loaders instead, the arrays gathered as objects.
        gather: callable(int64[k]) -> int64[world, k] (default: torch.distributed.all_gather).  -> rows written (rank 0), else None."""
        from . import _lib, fxi
        import time
        world, rank = self.world, self.rank
        marks = [("start", time.perf_counter())]

        def mark(what):
            marks.append((what, time.perf_counter()))
        if world > 1:
            import torch
            import torch.distributed as dist
            nccl = dist.get_backend(group) == "nccl"
        if gather is None:
            def gather(v):
                if world == 1:
                    return np.asarray(v, dtype=np.int64).reshape(1, -1)
                t = torch.tensor([int(x) for x in v], dtype=torch.int64, device=("cuda:%d" % self.device) if nccl else "cpu")
                outs = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(outs, t, group=group)
                return np.stack([o.cpu().numpy() for o in outs])
        # ---- 1. shapes
        bad = 0
        try:
            n, nleaf, nb = self.blob.fxi_part_shape(1, self.first_id)
        except _lib.FxError as e:
            if e.code not in (_lib.FX_ERANGE, _lib.FX_ENOMEM):
                raise
            n, nleaf, nb, bad = self.n_local, 0, 0, 1
        shapes = np.asarray(gather([n, nleaf, nb, bad, self.size]), dtype=np.int64).reshape(world, 5)
        n_total, leaves_total = int(shapes[:, 0].sum()), int(shapes[:, 1].sum())
        mark("shapes_gathered")
        if shapes[:, 3].any():
            return self._write_index_host(index_file, group)
        # ---- 2. the database (rank 0); room for all its pages is made in a thread while the names are packed and posted
        w = None
        head = [0, 0, 0]
        if rank == 0:
            try:
                pre = False
                if self._presized is not None:                # the file made while the ranks staged: schema in place, room allocated
                    pf, tok = self._presized
                    self._presized = None
                    pre = pf == index_file
                    _lib.fxi_presize_end(tok, cancel=not pre)
                    if not pre and os.path.exists(pf):
                        os.remove(pf)
                w = fxi.PartsWriter(index_file, 1, self.device, schema_done=pre)
                w.reserve(leaves_total, n_total, int(shapes[:, 2].sum()), background=True)
                head = [w.first_new_page, w.root["read"], 1]
            except (_lib.FxError, OSError):
                w = None
        mark("database_made")
        # ---- 3. names to rank 0 (posted now, travelling while the leaves are written)
        names = lens = None
        if n:
            if world == 1:
                names, lens = w.dev_buffers(n, nb)
            else:
                names = torch.empty(nb + 64, dtype=torch.uint8, device="cuda:%d" % self.device)
                lens = torch.empty(n, dtype=torch.int32, device="cuda:%d" % self.device)
            self.blob.fxi_part_names(1, names.data_ptr(), lens.data_ptr())
            names[nb:].zero_()
        mark("names_packed")
        reqs, parts = [], []
        if world > 1:
            def ship(t):                                      # what the backend can send: device memory (RCCL) or a host copy (gloo)
                return t if nccl else t.cpu()
            if rank == 0:
                # ONE receive buffer for everything the other ranks send (device memory for RCCL, one pinned block for gloo):
                # per rank its names, then -- 8-byte aligned -- its lengths and the first rows of its leaves
                def up8(x):
                    return (int(x) + 7) & ~7
                need = sum(up8(shapes[r, 2]) + up8(4 * shapes[r, 0]) + 8 * int(shapes[r, 1]) for r in range(1, world))
                pool = torch.empty(max(need, 8), dtype=torch.uint8, device=("cuda:%d" % self.device) if nccl else "cpu", pin_memory=not nccl)
                at = 0
                for r in range(1, world):
                    nr, lr, br = int(shapes[r, 0]), int(shapes[r, 1]), int(shapes[r, 2])
                    if nr == 0:
                        parts.append(None)
                        continue
                    bn = pool[at:at + br]
                    at += up8(br)
                    bl = pool[at:at + 4 * nr].view(torch.int32)
                    at += up8(4 * nr)
                    bf = pool[at:at + 8 * lr].view(torch.int64)
                    at += 8 * lr
                    reqs += [dist.irecv(bn, src=r, group=group), dist.irecv(bl, src=r, group=group)]
                    parts.append((nr, lr, bn, bl, bf))
            elif n:
                reqs += [dist.isend(ship(names[:nb]), dst=0, group=group), dist.isend(ship(lens), dst=0, group=group)]
        mark("names_posted")
        # ---- where the new pages begin: three words from rank 0, once the room is there
        if w is not None:
            try:
                w.reserved()
            except (_lib.FxError, OSError, RuntimeError):
                pass                                          # (best effort: the parts fall back to pwrite)
        mark("room_made")
        head = np.asarray(gather(head), dtype=np.int64).reshape(world, 3)[0]
        if not head[2]:
            for q in reqs:
                q.wait()
            return self._write_index_host(index_file, group)
        first_new_page = int(head[0])
        leaf_base = int(shapes[:rank, 1].sum())               # (a table of one leaf in all: PartsWriter.finish moves it into the root page)
        mark("first_page_known")
        err = None
        box = {}

        def leaves():
            try:
                if n:
                    box["laps"] = self.blob.fxi_part_leaves(1, index_file, first_new_page, leaf_base)
            except _lib.FxError as e:                         # (said in step 4: nobody may be left waiting)
                box["err"] = e
        th = None
        if rank == 0 and world > 1:
            # the writer's own leaves go out in a thread (the copy-out is the host's work, the call releases the interpreter
            # lock): the names of the others arrive and are sorted on the device meanwhile
            import threading
            th = threading.Thread(target=leaves)
            th.start()
        else:
            leaves()
            mark("own_leaves_in_the_file")
        for q in reqs:
            q.wait()
        mark("names_arrived")
        if th is not None:
            try:
                if n:
                    w.add_remote(n, nleaf, None, names[:nb], lens, slack=True)
                for pr in parts:
                    if pr is not None:
                        w.add_remote(pr[0], pr[1], None, pr[2], pr[3])
                w.join_names()
            except _lib.FxError as e:
                box.setdefault("err", e)
            mark("names_sorted")
            th.join()
            mark("own_leaves_in_the_file")
        err = box.get("err")
        if "laps" in box:
            self.index_laps = box["laps"]
        # ---- 4. first rows to rank 0 (their arrival: this rank's leaves are in the file), then the writer's share
        firsts = self.blob.fxi_part_firsts(nleaf) if (n and err is None) else np.zeros(nleaf, dtype=np.int64)
        flag = np.asarray(gather([0 if err is None else 1]), dtype=np.int64).reshape(world)
        if flag.any():
            if w is not None:
                w.abort()
            if err is not None:
                raise err
            raise _lib.FxError(_lib.FX_EIO, "another rank could not write its pages of %s" % index_file)
        out = None
        if world > 1 and rank != 0 and n:
            tf = torch.from_numpy(firsts)
            dist.send(tf.to("cuda:%d" % self.device) if nccl else tf, dst=0, group=group)
        if rank == 0:
            if world == 1:
                if n:
                    w.add_remote(n, nleaf, firsts, names[:nb], lens, slack=True)
            elif n:
                w.set_firsts(0, firsts)
            k = 1 if n else 0
            for r in range(1, world):
                pr = parts[r - 1] if world > 1 else None
                if pr is None:
                    continue
                dist.recv(pr[4], src=r, group=group)
                w.set_firsts(k, pr[4].cpu().numpy())
                k += 1
            mark("first_rows_arrived")
            db = w.finish()
            mark("index_and_levels_written")
            size = int(shapes[:, 4].sum())
            db.execute("INSERT INTO stat VALUES (?,?,?)", (n_total, size, size * 1.0 / n_total if n_total else float("nan")))     # fastq.c:161
            db.commit()
            db.close()
            own = getattr(self, "index_laps", {})
            self.index_laps = dict(w.laps, **{k: v for k, v in own.items() if v})
            out = n_total
        if world > 1:
            dist.barrier(group=group)
        mark("done")
        self.index_steps = {b[0]: round(b[1] - a[1], 4) for a, b in zip(marks[:-1], marks[1:])}
        return out

    def _write_index_host(self, index_file, group=None):
        """The fall-back of write_index (a row that needs an overflow page, no room on a device): every rank's arrays to rank 0
        as objects through the process group, the host page loader there (fxi.write_fastq_bulk; INSERTs when that declines too)."""
        from . import _lib, fxi
        part = self.local_part()
        parts = [part]
        if self.world > 1:
            import torch.distributed as dist
            parts = [None] * self.world
            dist.all_gather_object(parts, part, group=group)
        n_total = None
        if self.rank == 0:
            cols = {k: np.concatenate([p[k] for p in parts]) for k in ("dlen", "rlen", "soff", "qoff")}
            names = np.concatenate([p["names"] for p in parts])
            shift = np.concatenate([[0], np.cumsum([int(p["name_off"][-1]) for p in parts])]).astype(np.int64)
            name_off = np.concatenate([p["name_off"][:-1] + shift[r] for r, p in enumerate(parts)] + [shift[-1:]])
            size, n_total = int(sum(int(p["size"][0]) for p in parts)), int(name_off.size - 1)
            if os.path.exists(index_file):
                os.remove(index_file)
            order, ndup = _lib.sort_packed_names(names, name_off, self.device)
            try:
                db = fxi.write_fastq_bulk(index_file, names, name_off, cols, size, None if ndup else order)
            except _lib.FxError:
                nm, no = names.tobytes(), name_off.tolist()
                db = fxi.connect(index_file)
                fxi.write_fastq(db, [nm[no[i]:no[i + 1]] for i in range(n_total)], cols, size)
            db.commit()
            db.close()
        if self.world > 1:
            dist.barrier(group=group)
        return n_total

    def composition(self, gather=None):
        """base / meta of the WHOLE file on every rank (pyfastx_fastq_calc_composition, fastq.c:663-795): every rank counts the
        reads it owns, one more latency-bound all-gather carries five sums, two minima and two maxima per rank (ten words), and
        the phred rule (fastq.c:768-774) is applied to the merged extremes.  gather: as for the constructor (callable(int64[10])
        -> int64[world, 10]); default torch.distributed, fx_comm through Comm.allgather."""
        from .windows import merge_fastq_meta, finish_fastq_comp
        mine = np.zeros(10, dtype=np.int64)
        mine[5:] = (0, 1 << 62, 104, 33, 0)                                # no read: neutral for max / min / min / max
        if self.n_local:
            bs, mt = self.blob.fastq_comp()
            mine[:5], mine[5:] = bs, mt
        if gather is not None:
            allc = np.asarray(gather(mine)).reshape(self.world, 10)
        elif self.world > 1:
            import torch
            import torch.distributed as dist
            t = torch.from_numpy(mine.copy())
            outs = [torch.empty_like(t) for _ in range(self.world)]
            dist.all_gather(outs, t)
            allc = np.stack([o.numpy() for o in outs])
        else:
            allc = mine.reshape(1, 10)
        base, meta = np.zeros(5, dtype=np.int64), None
        for row in allc:
            base += row[:5]
            if row[6] != (1 << 62):                                          # a rank that owns reads
                meta = merge_fastq_meta(meta, row[5:])
        return finish_fastq_comp(base, meta)

    def fetch(self, ids, first_ids, phred=0, seq_flags=0, want=("seq", "qual", "quali")):
        """The reads of a batch (0-based GLOBAL ids, the same batch on every rank) that THIS rank owns -- read.c:37-45, 152-167,
        237-278 -- -> (positions of those reads in the batch, seq, qual, quali, offsets): every read is answered exactly once
        over the ranks, by the rank whose shard its header line begins in; no bytes move between GPUs.  first_ids: global id of
        every rank's first read (cumulative n_local: what the build's all-gather already told everybody)."""
        ids = np.asarray(ids, dtype=np.int64)
        lo = int(first_ids[self.rank])
        mine = np.nonzero((ids >= lo) & (ids < lo + self.n_local))[0]
        if not mine.size:
            z = np.zeros(0, dtype=np.uint8)
            return mine, (z if "seq" in want else None), (z if "qual" in want else None), (z.view(np.int8) if "quali" in want else None), np.zeros(1, dtype=np.int64)
        seq, qual, qi, offs = self.blob.fastq_fetch_alloc(ids[mine] - lo, phred=phred, seq_flags=seq_flags, want=want)
        return mine, seq, qual, qi, offs
