"""Streams larger than the HBM they may use: byte-range WINDOWS that take turns on one GPU.

The reference indexes files of any size through a 1 MiB buffer (kseq.h:13, index.c:229-230) and reads single ranges with
fseek / zran_seek (index.c:683-692).  Here the stream is resident in HBM -- until it does not fit: its benchmark set goes up
to 32 Gbp genomes and 39 Gbp FASTQ files (benchmark/README.md:18-50), and a GPU may be shared.  When the uncompressed stream
is larger than the budget (FX_HBM_BUDGET, e.g. "64M" / "200G"; default: what is free on the device now, less a margin for the
build's own arrays) the file is cut into W equal byte ranges, and

* the INDEX BUILD runs window after window on the one device with the machinery of the multi-GPU build:
  fx_open_file_range stages a window, the ordinary scan runs over it, its 28-word boundary summary is kept, and when all
  windows have been seen every window's last record is finished from the summaries of the windows behind it
  (shard.stitch_tail -- the same integer logic the all-gather feeds on 8 GPUs); FASTQ windows carry a halo and need only
  the running line count of the windows before them (shard.fastq_contexts), so ONE pass does it.  The rows of all windows
  make ONE .fxi, row for row what a build of the whole file writes;
* FETCHES are routed to the window that holds their bytes (shard.ShardFetcher / route by read id) and the windows they
  need are staged on demand; the budget's worth of windows stays resident, least recently used first out.

A single gzip stream cannot be entered in the middle (no window but the first can be staged without inflating everything
before it): such a file must fit, as before.
"""
import collections
import os

import numpy as np

from . import _lib, shard
from .multi import ShardedBlob

_UNITS = {"K": 1 << 10, "M": 1 << 20, "G": 1 << 30, "T": 1 << 40}


def parse_size(text):
    t = str(text).strip().upper().rstrip("B").rstrip("I")
    if t and t[-1] in _UNITS:
        return int(float(t[:-1]) * _UNITS[t[-1]])
    return int(float(t))


def hbm_budget(device=0):
    """Bytes of ONE stream that may be resident at a time: FX_HBM_BUDGET, else 85 % of what the device has free now."""
    env = os.environ.get("FX_HBM_BUDGET")
    if env:
        return max(parse_size(env), 1 << 16)
    free, _ = _lib.device_memory(device)
    if free * 0.85 < (16 << 30):
        # little room: blocks of closed streams that idle in the library's pool count as used (fx_device_memory) -- give
        # them back and look again, before a stream that fits is sent through windows for nothing
        _lib.lib().fx_release_scratch()
        free, _ = _lib.device_memory(device)
    return int(free * 0.85)


def bgzf_head_ratio(path, head=1 << 18):
    """Inflated / compressed bytes of the BGZF members in the first `head` bytes of a file, or None when the file does not
    begin with BGZF members (header layout of fx_bgzf_walk.hpp: 1f 8b 08 04, XLEN 6, 'B' 'C' 2 0, BSIZE; ISIZE closes a member)."""
    try:
        with open(path, "rb") as f:
            b = f.read(head)
    except OSError:
        return None
    at = cin = cout = 0
    while at + 18 <= len(b) and b[at:at + 4] == b"\x1f\x8b\x08\x04" and b[at + 10:at + 16] == b"\x06\x00BC\x02\x00":
        size = int.from_bytes(b[at + 16:at + 18], "little") + 1
        if size < 26 or at + size > len(b):
            break
        cout += int.from_bytes(b[at + size - 4:at + size], "little")
        cin += size
        at += size
    return cout / cin if cin >= 4096 or (cin and at == len(b)) else None


def plan(path, device=0, factor=1.0, exact=False):
    """-> None when the stream fits (size * factor <= budget; factor: what a build needs beside the stream itself), else
    (size, kind, window bytes, windows resident at a time).  A BGZF file whose size -- estimated from the ratio of its first
    members -- is under a quarter of the budget is taken to fit without the walk over all its members that the exact size
    costs (20 ms for the 46 723 members of C4, a quarter of Fasta(path)); a caller whose open then fails for lack of memory
    asks again with exact=True."""
    if not exact:
        r = bgzf_head_ratio(path)
        if r is not None and os.path.getsize(path) * max(r, 1.0) * 4 * factor <= hbm_budget(device):
            return None
    size, kind = _lib.stream_size(path)
    if kind == 2 or size <= 0:                                  # a single gzip stream does not shard by byte range
        return None
    budget = hbm_budget(device)
    if size * factor <= budget:
        return None
    if not os.environ.get("FX_HBM_BUDGET"):                     # does it fit once the idle blocks of the pool are back with the driver?
        _lib.lib().fx_release_scratch()
        budget = hbm_budget(device)
        if size * factor <= budget:
            return None
    win = max(int(budget / factor) // 4, 1 << 16)               # four windows' worth of stream within the budget
    return size, kind, win, max(2, int(budget / factor) // win)


class WindowCache:
    """The staged windows of one file: at most `capacity` resident, least recently used first out."""

    def __init__(self, path, bounds, device, capacity, halos=None):
        self.devices = list(device) if isinstance(device, (list, tuple)) else [device]       # window w lives on devices[w % len]
        self.path, self.bounds, self.device, self.capacity = path, list(bounds), self.devices[0], max(1, int(capacity))
        self.halos = dict(halos or {})
        self.lru = collections.OrderedDict()
        self.staged = 0                                         # windows staged so far (a measure of the traffic over PCIe)
        self.on_stage = None                                    # callable(w, blob): what a freshly staged window still needs (e.g. its table)

    def device_of(self, w):
        return self.devices[int(w) % len(self.devices)]

    def get(self, w):
        w = int(w)
        b = self.lru.get(w)
        if b is not None:
            self.lru.move_to_end(w)
            return b
        while len(self.lru) >= self.capacity:
            _, old = self.lru.popitem(last=False)
            old.close()
        a, e = self.bounds[w], self.bounds[w + 1]
        halo = min(int(self.halos.get(w, 0)), self.bounds[-1] - e)
        b = _lib.Blob.from_file_range(self.path, a, e - a, halo, self.device_of(w))
        self.lru[w] = b
        self.staged += 1
        if self.on_stage is not None:
            self.on_stage(w, b)
        return b

    def drop(self, w):
        b = self.lru.pop(int(w), None)
        if b is not None:
            b.close()

    def close(self):
        for b in self.lru.values():
            b.close()
        self.lru.clear()


class _LazyBlobs:
    """What multi.ShardedBlob / shard.ShardFetcher index by shard number, with the windows staged when they are asked for."""

    def __init__(self, cache, n):
        self.cache, self.n = cache, n

    def __getitem__(self, w):
        return self.cache.get(w)

    def values(self):
        return list(self.cache.lru.values())


class _WindowFetcher:
    def __init__(self, cache, w):
        self.cache, self.w = cache, w

    def fetch_ranges(self, *a, **k):
        return self.cache.get(self.w).fetch_ranges(*a, **k)


class WindowedBlob(ShardedBlob):
    """ShardedBlob over windows, plus the read gather of the FASTQ iterator (read.c:152-167, 237-278) from byte ranges."""

    def read_fetch(self, soff, qoff, rlen, phred=0, seq_flags=0, want=("seq", "qual", "quali")):
        soff, qoff, rlen = (np.asarray(x, dtype=np.int64) for x in (soff, qoff, rlen))
        seq = qual = qi = None
        offs = np.zeros(rlen.size + 1, dtype=np.int64)
        np.cumsum(rlen, out=offs[1:])
        if "seq" in want:
            seq, _, _ = self.fetch_ranges(soff, rlen, rlen, flags=8 | (int(seq_flags) & 7))
        if "qual" in want or "quali" in want:
            qual, _, _ = self.fetch_ranges(qoff, rlen, rlen, flags=8)
            if "quali" in want:
                qi = (qual.view(np.int8).astype(np.int16) - (int(phred) or 33)).astype(np.int8)          # read.c:268
        return seq, qual if "qual" in want else None, qi, offs


# ================================================================================================ FASTA
class WindowedFasta:
    """The index build of one FASTA file over W windows of one device -> merged table, fetcher, blob adapter: the interface
    of multi.MultiDevice, so api.Fasta drives both the same way."""

    def __init__(self, path, device=0, full_name=False, window=None, capacity=None):
        size, kind = _lib.stream_size(path)
        if kind == 2:
            raise ValueError("%s is a single gzip stream: it cannot be staged in windows" % path)
        if window is None:
            p = plan(path, device, 1.15)
            window, capacity = (p[2], p[3]) if p else (size, 2)
        W = max(1, -(-size // int(window)))
        bounds = [size * i // W for i in range(W + 1)]
        self.cache = WindowCache(path, bounds, device, capacity or 2)
        self.full_name = bool(full_name)
        S, parts, counts = [], [], []
        for w in range(W):
            b = self.cache.get(w)
            n = int(b.fasta_build(full_name).n_seq)
            S.append(b.shard_summary())
            part = shard.local_index_part(b, n, bounds[w], bounds[w + 1] - bounds[w])
            if n and S[-1].tail_e < 0:
                # the last header line of the window is not over at its end: how much of it is the NAME is only known after the
                # stitch, so everything the window holds of it is kept (cut to the final length below, completed from the next
                # windows' first bytes by merge_index_parts)
                h1 = int(part[2]["hoff"][n - 1]) + 1
                part[3][n - 1] = b.read_bytes(h1, bounds[w + 1] - h1)
            parts.append(part)
            counts.append(n)
        self._rows_fix = {}
        for r in range(W):                                      # every window's last record, finished from the windows behind it
            row = shard.stitch_tail(S, r, full_name)
            if row is not None and counts[r]:
                self._rows_fix[r] = row
                for k, v in row.items():
                    if k in parts[r][2]:
                        parts[r][2][k][counts[r] - 1] = v
                parts[r][3][counts[r] - 1] = parts[r][3][counts[r] - 1][:max(int(row["name_len"]), 0)]
        self.table = shard.merge_index_parts(parts)
        if len(self.table["names"]) == 0:
            raise _lib.FxError(_lib.FX_EFORMAT, "no FASTA header line ('>') found")
        self.counts, self.summaries = counts, S
        self.bases, self.ends = bounds[:-1], bounds[1:]
        self.size, self.kind, self.devices, self.windows = size, kind, [device], W
        self.blobs = _LazyBlobs(self.cache, W)
        self.blob = WindowedBlob(self.blobs, self.bases, self.ends, self.table["reg"])

    def fetcher(self):
        return shard.ShardFetcher({w: _WindowFetcher(self.cache, w) for w in range(self.windows)}, self.bases, self.ends, self.table)

    def _built(self, w):
        """Window w resident WITH its record table (a window that was evicted since the build is scanned again)."""
        b = self.cache.get(w)
        if getattr(b, "_n_fasta", None) is None:
            b.fasta_build(self.full_name)
        if w in self._rows_fix and not getattr(b, "_fx_stitched", False):
            b.fasta_set_row(self.counts[w] - 1, **self._rows_fix[w])
            b._fx_stitched = True
        return b

    def composition(self):
        """Per-record composition (fasta.c:851-961) window after window; the bytes before a window's first header line go to
        the record that owns them (shard.comp_lead_from / comp_fold_leads, as across GPUs)."""
        n = self.counts
        first = np.concatenate([[0], np.cumsum(n)])
        boffs = [int(self.table["boff"][first[r + 1] - 1]) if n[r] else -1 for r in range(len(n))]
        comps, leads = [], []
        for r in range(self.windows):
            c, lead = self._built(r).fasta_comp_shard(n[r], shard.comp_lead_from(self.bases, boffs, r))
            comps.append(c)
            leads.append(lead)
        for r in range(len(comps)):
            shard.comp_fold_leads(comps[r], leads, n, r)
        return np.concatenate(comps) if comps else np.zeros((0, 128), dtype=np.int64)


# ================================================================================================ FASTQ
def merge_fastq_meta(meta, mt):
    """meta rows (maxlen, minlen, minqs, maxqs, phred) of two byte ranges -> of both (the phred guess is made at the end)."""
    if meta is None:
        return np.asarray(mt, dtype=np.int64).copy()
    return np.array([max(meta[0], mt[0]), min(meta[1], mt[1]), min(meta[2], mt[2]), max(meta[3], mt[3]), 0], dtype=np.int64)


def finish_fastq_comp(base_sum, meta):
    """-> (base, meta) of the whole file: the phred rule of fastq.c:768-774 on the merged extremes."""
    if meta is None:
        meta = np.array([0, 0, 104, 33, 0], dtype=np.int64)      # fastq.c:667-668: no read at all
    meta = np.asarray(meta, dtype=np.int64).copy()
    phred = 0
    if meta[3] > 74:
        phred = 64
    if meta[2] < 59:
        phred = 33
    meta[4] = phred
    return np.asarray(base_sum, dtype=np.int64), meta


class WindowedFastq:
    """pyfastx_fastq_create_index (fastq.c:8-182) and the read fetch (read.c:37-45) over W byte ranges of one file.

    Range w stages bytes [size w / W, size (w + 1) / W) and a halo behind them; a read belongs to the range its header line
    begins in and is finished from the halo (a halo that proves too small -- FX_ERANGE from fx_fastq_build_ctx: a read of a
    megabyte -- is grown eightfold for that range).  The line numbering a range needs is the running count of the ranges before
    it (shard.fastq_contexts).  Two ways to hold the ranges:

    * windows of ONE device that take turns (a stream larger than the HBM it may use): the build is ONE pass, range after
      range, a budget's worth of them resident, least recently used first out;
    * devices=[...] (`Fastq(path, devices=[0, 1, ...])`, SURVEY 8e in one process): one range per listed device, all resident,
      staged and scanned at the same time by one host thread each, then built at the same time once the counts are known.

    The merged table stays on the host (dlen, rlen, soff, qoff: what the .fxi holds); batches are routed by read id to the
    ranges, which answer with fx_read_fetch."""

    HALO0 = 1 << 16

    def __init__(self, path, device=0, window=None, capacity=None, want_comp=False, devices=None, index_file=None):
        """index_file: where the .fxi will go (a path that does not exist yet).  The table leaves of every range are then formatted
        and written WHILE THE RANGE IS RESIDENT for its build (fxi.PartsWriter.add_local) -- a stream larger than its budget is
        staged once, not once for the build and once more for the index file -- and write_index only has the names to sort and
        the index to write."""
        size, kind = _lib.stream_size(path)
        if kind == 2:
            raise ValueError("%s is a single gzip stream: it cannot be staged by byte range" % path)
        if devices:
            devices = list(devices)[:max(1, min(len(devices), size))]
            W = len(devices)
            capacity = W
        else:
            devices = [device]
            if window is None:
                p = plan(path, device, 1.7)
                window, capacity = (p[2], p[3]) if p else (size, 2)
            W = max(1, -(-size // int(window)))
        bounds = [size * i // W for i in range(W + 1)]
        self.cache = WindowCache(path, bounds, devices, capacity or 2)
        self.path, self.device, self.devices, self.stream_bytes, self.windows = path, devices[0], devices, size, W
        self._want_comp = bool(want_comp)
        self.ctx = [None] * W
        parts = [None] * W
        comps = [None] * W
        self._writer = None
        if index_file and index_file != ":memory:" and not os.path.exists(index_file) and not os.environ.get("FX_FXI_HOST"):
            from . import fxi
            try:
                self._writer = fxi.PartsWriter(index_file, 1, devices[0])
            except (_lib.FxError, OSError):
                self._writer = None
        if len(devices) > 1 or (capacity or 2) >= W > 1:
            self._build_resident(bounds, parts, comps)
            for w in range(W):                                  # (all ranges are resident: their leaves in order)
                self._part_to_index(w, self.cache.lru[w])
        else:
            cores = []
            for w in range(W):                                  # one pass: a range needs only the counts of the ranges before it
                b, core = self._open_scan(w, self.HALO0)
                cores.append(core)
                self.ctx[w] = shard.fastq_contexts(cores)[w]
                parts[w], comps[w] = self._build_range(w, b)
                self._part_to_index(w, self.cache.get(w))       # (the range as _build_range left it: opened again if its halo had to grow)
        self.table = {k: np.concatenate([p[k] for p in parts]) for k in ("dlen", "rlen", "soff", "qoff")}
        self.names = np.concatenate([p["names"] for p in parts]) if parts else np.zeros(0, dtype=np.uint8)
        shift = np.concatenate([[0], np.cumsum([int(p["name_off"][-1]) for p in parts])]).astype(np.int64)
        self.name_off = np.concatenate([p["name_off"][:-1] + shift[r] for r, p in enumerate(parts)] + [shift[-1:]])
        self.first_id = np.concatenate([[0], np.cumsum([p["n"] for p in parts])]).astype(np.int64)     # reads before each range
        self.n_reads, self.size = int(self.first_id[-1]), int(sum(p["size"] for p in parts))
        self.bases, self.ends = bounds[:-1], bounds[1:]
        self.comp = None
        if want_comp:
            base_sum, meta = np.zeros(5, dtype=np.int64), None
            for c in comps:
                if c is not None:
                    base_sum += c[0]
                    meta = merge_fastq_meta(meta, c[1])
            self.comp = finish_fastq_comp(base_sum, meta)
        self.blobs = _LazyBlobs(self.cache, W)
        self.blob = WindowedBlob(self.blobs, self.bases, self.ends, np.zeros(0, dtype=np.int32))

    def _part_to_index(self, w, blob):
        """The table leaves of range w into the index file while the range is resident; its names to the writer's device."""
        if self._writer is None:
            return
        try:
            self._writer.add_local(blob, device=self.cache.device_of(w))
        except _lib.FxError as e:                                # (the writer has removed its file: write_index takes the other routes)
            self._writer = None
            if e.code not in (_lib.FX_ERANGE, _lib.FX_EINVAL, _lib.FX_ENOMEM):
                raise

    # ---- the steps of one range
    def _open_scan(self, w, halo):
        self.cache.drop(w)
        self.cache.halos[w] = halo
        b = self.cache.get(w)
        return b, b.fastq_scan()

    def _build_range(self, w, b):
        """Rows, names (and composition) of range w, whose context is known; a halo too small for one of its reads is grown."""
        halo = int(self.cache.halos.get(w, self.HALO0))
        while True:
            try:
                s = b.fastq_build_ctx(*self.ctx[w])
                break
            except _lib.FxError as e:
                if e.code != _lib.FX_ERANGE or halo >= self.stream_bytes - self.cache.bounds[w + 1]:
                    raise
                halo *= 8
                b, _ = self._open_scan(w, halo)                  # (the counts of the core do not depend on the halo)
        n = int(s.n_reads)
        t = b.fastq_table(n)
        packed, offs = b.names_pack(1, n, guess=int(np.maximum(t["name_len"], 0).sum()))
        part = {"dlen": np.asarray(t["dlen"], np.int64), "rlen": np.asarray(t["rlen"], np.int64), "soff": np.asarray(t["soff"], np.int64),
                "qoff": np.asarray(t["qoff"], np.int64), "names": np.asarray(packed, np.uint8).copy(), "name_off": np.asarray(offs, np.int64),
                "size": int(s.size), "n": n}
        comp = b.fastq_comp() if (self._want_comp and n) else None
        return part, comp

    def _build_resident(self, bounds, parts, comps):
        """All ranges resident (one per device, or few enough for one device): staged and scanned at the same time, built at the
        same time -- one host thread per range, as multi.MultiDevice does for FASTA."""
        import threading
        W = self.windows
        cores, errs = [None] * W, []
        lock = threading.Lock()

        def stage(w):
            try:
                a, e = bounds[w], bounds[w + 1]
                halo = min(self.HALO0, bounds[-1] - e)
                b = _lib.Blob.from_file_range(self.path, a, e - a, halo, self.cache.device_of(w))
                c = b.fastq_scan()
                with lock:
                    self.cache.halos[w] = self.HALO0
                    self.cache.lru[w] = b
                    self.cache.staged += 1
                    cores[w] = c
            except Exception as ex:                               # noqa: BLE001
                errs.append(ex)

        def build(w):
            try:
                parts[w], comps[w] = self._build_range(w, self.cache.lru[w])
            except Exception as ex:                               # noqa: BLE001
                errs.append(ex)

        for fn in (stage, build):
            th = [threading.Thread(target=fn, args=(w,)) for w in range(W)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            if errs:
                raise errs[0]
            if fn is stage:
                ctx = shard.fastq_contexts(cores)
                for w in range(W):
                    self.ctx[w] = ctx[w]

    def _built(self, w):
        b = self.cache.get(w)
        if getattr(b, "_n_fastq", None) is None:
            b.fastq_scan()
            b.fastq_build_ctx(*self.ctx[w])
        return b

    def composition(self):
        """base / meta of the whole file (fastq.c:663-795): five sums, two minima, two maxima over the ranges."""
        if self.comp is None:
            base_sum, meta = np.zeros(5, dtype=np.int64), None
            for w in range(self.windows):
                if self.first_id[w + 1] > self.first_id[w]:
                    bs, mt = self._built(w).fastq_comp()
                    base_sum += bs
                    meta = merge_fastq_meta(meta, mt)
            self.comp = finish_fastq_comp(base_sum, meta)
        return self.comp

    def write_index(self, index_file):
        """ONE .fxi, every page of `read` and `readidx` formatted on a device (fxi.PartsWriter, round 6): range after range --
        staged again if it has been evicted since the build -- formats the table leaves of its rows and hands over its names;
        one sort, the index leaves, the host's levels.  FX_ERANGE / FX_EINVAL (a row that needs an overflow page, a database
        without 4 KiB pages): the host page loader from the merged host arrays, as before."""
        from . import fxi
        try:
            w, self._writer = self._writer, None
            if w is not None and (w.path != index_file or w.rows != int(self.n_reads)):
                w.abort()
                w = None
            if w is None:                                         # (no writer rode along with the build: the ranges once more, staged again where they were evicted)
                w = fxi.PartsWriter(index_file, 1, self.device)
                for r in range(self.windows):
                    if self.first_id[r + 1] > self.first_id[r]:
                        w.add_local(self._built(r), device=self.cache.device_of(r))
            db = w.finish()
            self.index_laps = dict(w.laps)
            n = int(self.n_reads)
            db.execute("INSERT INTO stat VALUES (?,?,?)", (n, int(self.size), self.size * 1.0 / n if n else float("nan")))     # fastq.c:161
            return db
        except _lib.FxError as e:
            if e.code not in (_lib.FX_ERANGE, _lib.FX_EINVAL, _lib.FX_ENOMEM):
                raise
        order, ndup = _lib.sort_packed_names(self.names, self.name_off, self.device)
        return fxi.write_fastq_bulk(index_file, self.names, self.name_off, self.table, self.size, None if ndup else order)

    def fetch(self, ids, phred=0, seq_flags=0, want=("seq", "qual", "quali")):
        """Reads by 0-based id -> (seq, qual, quali, offsets): every range answers the reads it owns (fx_read_fetch on the rows of
        the host table), staged on demand."""
        ids = np.asarray(ids, dtype=np.int64)
        if ids.size and (ids.min() < 0 or ids.max() >= self.n_reads):
            raise IndexError("index out of range")
        rl = self.table["rlen"][ids]
        offs = np.zeros(ids.size + 1, dtype=np.int64)
        np.cumsum(rl, out=offs[1:])
        tot = int(offs[-1])
        out = {k: (np.empty(max(tot, 1), dtype=np.int8 if k == "quali" else np.uint8) if k in want else None) for k in ("seq", "qual", "quali")}
        win = np.searchsorted(self.first_id, ids, "right") - 1
        for w in np.unique(win).tolist():
            m = np.nonzero(win == w)[0]
            q = ids[m]
            sq, ql, qi, o = self.cache.get(w).read_fetch(self.table["soff"][q], self.table["qoff"][q], self.table["rlen"][q], phred=phred,
                                                         seq_flags=seq_flags, want=want)
            ln = rl[m]
            dst = np.repeat(offs[m] - o[:-1], ln) + np.arange(int(o[-1]), dtype=np.int64)
            for k, v in (("seq", sq), ("qual", ql), ("quali", qi)):
                if v is not None:
                    out[k][dst] = v[:int(o[-1])]
        return tuple(None if out[k] is None else out[k][:tot] for k in ("seq", "qual", "quali")) + (offs,)
