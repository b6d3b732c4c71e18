"""One FASTA file over several GPUs held by ONE process: `Fasta(path, devices=[0, 1, ...])`.

The same byte-range sharding as the one-process-per-GPU run (shard.ShardedFasta): device i stages and scans
only bytes [size * i / G, size * (i + 1) / G) of the uncompressed stream (fx_open_file_range: a plain file is read in
that range, of a BGZF file only the covering members), the boundary summaries are exchanged -- here simply collected,
the degenerate all-gather -- every shard finishes the record that crosses its end (shard.stitch_tail), and the rows of
all shards make ONE record table and ONE .fxi.  Bytes never move between GPUs: batched queries are routed to the
device that holds their bytes (shard.ShardFetcher), per-object getters read through `ShardedBlob`, which answers a byte
range from the shard(s) that hold it.  A device may be listed more than once (logical shards on one GPU: how the
single-GPU test box exercises this path).
"""
import threading

import numpy as np

from . import _lib, shard


class ShardedBlob:
    """What api._Staged hands to Sequence / Read code in place of one Blob: the byte-range calls of `_lib.Blob`
    (read_bytes, fetch_one, fetch_ranges) answered from the shards that hold the bytes."""

    _table_ready = True

    def __init__(self, blobs, bases, ends, reg):
        self.blobs, self._reg = blobs, np.asarray(reg, dtype=np.int32)
        self.bases, self.ends = np.asarray(bases, dtype=np.int64), np.asarray(ends, dtype=np.int64)
        self.size = int(self.ends[-1])
        self._n_fasta = len(self._reg)

    def fasta_line_regular(self, n):
        return self._reg

    def _pieces(self, off, n):
        P = shard.route_ranges(self.bases, self.ends, [off], [n])
        return list(zip(P["r"].tolist(), P["poff"].tolist(), P["plen"].tolist()))

    def read_bytes(self, off, n):
        out = b"".join(self.blobs[r].read_bytes(po, pl) for r, po, pl in self._pieces(off, n))
        return out + b"\0" * (max(int(n), 0) - len(out))        # past the end of the stream: zeros, as one Blob gives them

    def fetch_ranges(self, off, blen, slen, flags=0, flags_per_query=None, skip=None):
        off, blen, slen = (np.asarray(x, dtype=np.int64) for x in (off, blen, slen))
        n = off.size
        fl = np.full(n, int(flags), dtype=np.uint8) if flags_per_query is None else np.asarray(flags_per_query, dtype=np.uint8)
        sk = np.zeros(n, dtype=np.int64) if skip is None else np.asarray(skip, dtype=np.int64)
        offs = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.maximum(slen, 0), out=offs[1:])
        dst = np.zeros(max(int(offs[-1]), 1), dtype=np.uint8)
        out_len = np.zeros(n, dtype=np.int64)
        P = shard.route_ranges(self.bases, self.ends, off, blen)
        q, r = P["q"], P["r"]
        one = P["cnt"] == 1
        for sh in np.unique(r).tolist():                        # whole queries inside one shard: its kernel does everything
            m = np.nonzero((r == sh) & one[q])[0]
            if m.size:
                qq = q[m]
                b, o, ol = self.blobs[sh].fetch_ranges(off[qq], blen[qq], slen[qq], flags_per_query=fl[qq],
                                                       skip=sk[qq] if sk[qq].any() else None)
                for k, qi in enumerate(qq.tolist()):
                    dst[offs[qi]:offs[qi] + ol[k]] = b[o[k]:o[k] + ol[k]]
                    out_len[qi] = ol[k]
        for qi in np.nonzero(P["cnt"] > 1)[0].tolist():          # a range across a cut: despaced pieces, put together here
            f = int(fl[qi])
            parts = []
            for k in range(int(P["start"][qi]), int(P["start"][qi + 1])):
                b, o, ol = self.blobs[int(r[k])].fetch_ranges([P["poff"][k]], [P["plen"][k]], [P["plen"][k]], flags=f & 13)
                parts.append(b[:int(ol[0])].tobytes())
            ans = b"".join(parts)[int(sk[qi]):int(sk[qi]) + int(slen[qi])]
            if f & 2:
                ans = ans[::-1]
            dst[offs[qi]:offs[qi] + len(ans)] = np.frombuffer(ans, dtype=np.uint8)
            out_len[qi] = len(ans)
        return dst[:int(offs[-1])], offs, out_len

    def fetch_one(self, off, blen, slen, flags=0, skip=0):
        buf, _, ol = self.fetch_ranges([off], [blen], [slen], flags=flags, skip=[skip] if skip else None)
        return buf[:int(ol[0])].tobytes()

    def close(self):
        for b in self.blobs.values():
            b.close()


class MultiDevice:
    """The sharded build of one FASTA file in this process -> merged table, shard fetcher, ShardedBlob."""

    def __init__(self, path, devices, full_name=False):
        size, kind = _lib.stream_size(path)
        if kind == 2:
            raise ValueError("%s is a single gzip stream: it does not shard by byte range (replicas only)" % path)
        devices = list(devices)[:max(1, min(len(devices), size))]
        G = len(devices)
        bounds = [size * i // G for i in range(G + 1)]
        blobs, errs = [None] * G, []

        def open_range(i):                                      # ranges are staged concurrently: one host thread per shard
            try:
                blobs[i] = _lib.Blob.from_file_range(path, bounds[i], bounds[i + 1] - bounds[i], 0, devices[i])
            except Exception as e:                              # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=open_range, args=(i,)) for i in range(G)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]
        for b in blobs:
            b.fasta_build_begin(full_name)                      # every device scans its range at the same time
        counts = [int(b.fasta_build_end().n_seq) for b in blobs]
        S = [b.shard_summary() for b in blobs]                  # the exchange: 28 words per shard
        for r, b in enumerate(blobs):
            row = shard.stitch_tail(S, r, full_name)
            if row is not None:
                b.fasta_set_row(counts[r] - 1, **row)
        parts = [shard.local_index_part(b, counts[r], bounds[r], bounds[r + 1] - bounds[r]) for r, b in enumerate(blobs)]
        self.table = shard.merge_index_parts(parts)
        if len(self.table["names"]) == 0:
            raise _lib.FxError(_lib.FX_EFORMAT, "no FASTA header line ('>') found")
        self.blobs = dict(enumerate(blobs))
        self.bases, self.ends = bounds[:-1], bounds[1:]
        self.size, self.kind, self.devices = size, kind, devices
        self.blob = ShardedBlob(self.blobs, self.bases, self.ends, self.table["reg"])

    def fetcher(self):
        return shard.ShardFetcher(self.blobs, self.bases, self.ends, self.table)

    def composition(self):
        """Per-record composition across the cuts: every device counts its own bytes, the bytes before a shard's first
        header line are folded into the record that owns them (shard.comp_lead_from / comp_fold_leads)."""
        n = [int((np.asarray(self.table["hoff"]) >= lo).sum() - (np.asarray(self.table["hoff"]) >= hi).sum())
             for lo, hi in zip(self.bases, self.ends)]
        first = np.concatenate([[0], np.cumsum(n)])
        boffs = [int(self.table["boff"][first[r + 1] - 1]) if n[r] else -1 for r in range(len(n))]
        comps, leads = [], []
        for r, b in self.blobs.items():
            c, lead = b.fasta_comp_shard(n[r], shard.comp_lead_from(self.bases, boffs, r))
            comps.append(c)
            leads.append(lead)
        for r in range(len(comps)):
            shard.comp_fold_leads(comps[r], leads, n, r)
        return np.concatenate(comps) if comps else np.zeros((0, 128), dtype=np.int64)
