"""`.fxi` (SQLite3) writer / reader, host side.

The GPU produces offset arrays; this module stores them with exactly the schema,
column order and pragmas the reference uses (index.c:178-224, 342-372;
fasta.c:890-953; fastq.c:29-60, 136-171, 755-784; util.c:442-540), so an index
written here opens in reference pyfastx and vice versa.
"""
import math
import os
import sqlite3
import struct
import time

import numpy as np

FASTA_DDL = """
CREATE TABLE seq (
    ID INTEGER PRIMARY KEY, --seq identifier
    chrom TEXT, --seq name
    boff INTEGER, --seq offset start
    blen INTEGER, --seq byte length
    slen INTEGER, --seq length
    llen INTEGER, --line length
    elen INTEGER, --end length
    norm INTEGER, --line with the same length or not
    dlen INTEGER --description header line length
);
CREATE TABLE stat (
    seqnum INTEGER, --total seq counts
    seqlen INTEGER, --total seq length
    avglen REAL, --average seq length
    medlen REAL, --median seq length
    n50 INTEGER, --N50 seq length
    l50 INTEGER --L50 seq count
);
CREATE TABLE comp (
    ID INTEGER PRIMARY KEY, --comp identifier
    seqid INTEGER, --seq id
    abc INTEGER, --seq letter
    num INTEGER -- letter count
);
CREATE TABLE gzindex (
    ID INTEGER PRIMARY KEY,
    content BLOB
);
"""

FASTQ_DDL = """
CREATE TABLE read (
    ID INTEGER PRIMARY KEY, --read id
    name TEXT, --read name
    dlen INTEGER, --description length
    rlen INTEGER, --read length
    soff INTEGER, --read seq offset
    qoff INTEGER --read qual offset
);
CREATE TABLE gzindex (
    ID INTEGER PRIMARY KEY,
    content BLOB
);
CREATE TABLE stat (
    counts INTEGER, --read counts
    size INTEGER, --all read length
    avglen REAL --average read length
);
CREATE TABLE base (
    a INTEGER,
    c INTEGER,
    g INTEGER,
    t INTEGER,
    n INTEGER
);
CREATE TABLE meta (
    maxlen INTEGER, --maximum read length
    minlen INTEGER, --minimum read length
    minqs INTEGER, --max quality score
    maxqs INTEGER, --min quality score
    phred INTEGER --phred value
);
"""


def connect(path):
    """sqlite3_open of index.c:170 / fastq.c:62; ConnectionError like the reference."""
    try:
        db = sqlite3.connect(path, isolation_level=None, check_same_thread=False)
    except sqlite3.Error:
        raise ConnectionError("Could not open index file %s" % path)
    db.text_factory = lambda b: b.decode("utf-8", "surrogateescape")
    return db


def _name_bytes(names):
    """Names as the bytes the `chrom` / `name` TEXT column stores: raw bytes stay as they are (sqlite3_bind_text in the
    reference, index.c:239-251), str (key_func results) are encoded the way fxi.connect decodes them."""
    return [x if isinstance(x, (bytes, bytearray)) else str(x).encode("utf-8", "surrogateescape") for x in names]


def write_fasta(db, names, cols, seqlen_total):
    """INSERT INTO seq ... (index.c:226-251, 342-372).  names: list of bytes (or str);
    cols: dict of int arrays boff, blen, slen, llen, elen, norm, dlen.  The name is bound as bytes and cast to TEXT,
    so the column holds exactly the bytes of the file, as in the reference and in the bulk loader."""
    names = _name_bytes(names)
    db.executescript(FASTA_DDL)
    db.execute("PRAGMA synchronous=OFF")
    db.execute("PRAGMA locking_mode=EXCLUSIVE")
    db.execute("BEGIN TRANSACTION")
    n = len(names)
    it = zip(names, cols["boff"].tolist(), cols["blen"].tolist(), cols["slen"].tolist(), cols["llen"].tolist(),
             cols["elen"].tolist(), cols["norm"].tolist(), cols["dlen"].tolist())
    db.executemany("INSERT INTO seq VALUES (NULL,CAST(? AS TEXT),?,?,?,?,?,?,?)", it)
    db.execute("PRAGMA locking_mode=NORMAL")
    db.execute("COMMIT")
    try:        # duplicate names: the reference ignores the failure too (index.c:363-366)
        db.execute("CREATE UNIQUE INDEX chromidx ON seq (chrom)")
    except sqlite3.Error:
        pass
    db.execute("INSERT INTO stat (seqnum,seqlen) VALUES (?,?)", (n, int(seqlen_total)))


def write_fasta_comp(db, comp):
    """fasta.c:890-953: non-zero bins per record, then all 128 totals with seqid 0."""
    write_fasta_comp_rows(db, *comp_rows(comp))


def write_fasta_comp_rows(db, seqid, abc, num):
    """The comp table from its rows (comp_rows / fx_fasta_comp_sparse + the 128 totals), one INSERT each."""
    db.execute("PRAGMA synchronous=OFF")
    db.execute("BEGIN TRANSACTION")
    db.executemany("INSERT INTO comp VALUES (NULL,?,?,?)", zip(seqid.tolist(), abc.tolist(), num.tolist()))
    db.execute("CREATE INDEX seqidx ON comp (seqid)")
    db.execute("COMMIT")


def comp_rows(comp):
    """The rows of the `comp` table for a dense per-record composition (fasta.c:890-953): the non-zero bins of
    every record in record order, then all 128 totals with seqid 0.  -> (seqid, abc, num) int64 arrays."""
    import numpy as np
    rec, abc = np.nonzero(comp)
    tot = comp.sum(axis=0) if len(comp) else np.zeros(128, dtype=np.int64)
    seqid = np.concatenate([rec.astype(np.int64) + 1, np.zeros(128, dtype=np.int64)])
    letters = np.concatenate([abc.astype(np.int64), np.arange(128, dtype=np.int64)])
    num = np.concatenate([comp[rec, abc].astype(np.int64), tot.astype(np.int64)])
    return seqid, letters, num


def write_fasta_comp_bulk(path, seqid, abc, num):
    """write_fasta_comp for many rows: the `comp` table and its `seqidx` index written as b-tree pages
    (fx_fxi_bulk_rows without a TEXT column, fx_fxi_bulk_index_int) into a database that has NO open connection and
    an empty comp table; rows as comp_rows returns them (the last 128 are the seqid-0 totals).  Returns an open
    connection."""
    import numpy as np
    from . import _lib
    db = connect(path)
    if db.execute("SELECT count(*) FROM comp").fetchone()[0]:
        db.close()
        raise ValueError("comp table is not empty")
    db.execute("CREATE INDEX seqidx ON comp (seqid)")
    root = dict(db.execute("SELECT name, rootpage FROM sqlite_master").fetchall())
    db.close()
    n = len(seqid)
    _lib.fxi_bulk_rows(path, root["comp"], None, None, [seqid, abc, num])
    # (seqid, rowid) order: the seqid-0 totals (the last 128 rows) first, then the records' rows as they are
    order = np.concatenate([np.arange(n - 128, n, dtype=np.int64), np.arange(0, n - 128, dtype=np.int64)])
    _lib.fxi_bulk_index_int(path, root["seqidx"], seqid, order)
    return connect(path)


def write_fastq(db, names, cols, size):
    """fastq.c:76-171.  names: list of bytes (or str), stored verbatim (see write_fasta)."""
    names = _name_bytes(names)
    db.executescript(FASTQ_DDL)
    db.execute("PRAGMA synchronous = OFF")
    db.execute("PRAGMA locking_mode=EXCLUSIVE")
    db.execute("BEGIN TRANSACTION")
    it = zip(names, cols["dlen"].tolist(), cols["rlen"].tolist(), cols["soff"].tolist(), cols["qoff"].tolist())
    db.executemany("INSERT INTO read VALUES (NULL,CAST(? AS TEXT),?,?,?,?)", it)
    db.execute("PRAGMA locking_mode=NORMAL")
    db.execute("COMMIT")
    try:
        db.execute("CREATE UNIQUE INDEX readidx ON read (name)")
    except sqlite3.Error:
        pass
    n = len(names)
    avg = size * 1.0 / n if n else float("nan")              # fastq.c:161 (0/0 -> nan in C as well)
    db.execute("INSERT INTO stat VALUES (?,?,?)", (n, int(size), avg))


def _bulk_table(path, ddl, table, index_sql, index_name, packed_names, name_off, col_list, order, threads):
    """A NEW index file without one INSERT per record: SQLite creates the file and the schema, fx_fxi_bulk_rows
    writes the b-tree of `table` straight into it (rows arrive in rowid order) and, when `order` (the sorted order
    of distinct names, Blob.names_sort) is given, fx_fxi_bulk_index writes the UNIQUE INDEX b-tree the same way.
    Without `order` SQLite builds the index; a failure on duplicate names is ignored as in index.c:363-366 /
    fastq.c:152-156.  Names are stored as the raw bytes of the file, like sqlite3_bind_text in the reference.
    Returns an open connection.  FX_ERANGE (a row needs an overflow page): the file is removed and the error
    re-raised -- the caller falls back to the INSERT path."""
    from . import _lib
    db = connect(path)
    db.executescript(ddl)
    if order is not None:
        db.execute(index_sql)
    root = dict(db.execute("SELECT name, rootpage FROM sqlite_master").fetchall())
    db.close()
    try:
        _lib.fxi_bulk_rows(path, root[table], packed_names, name_off, col_list)
    except BaseException:                                     # row too large (FX_ERANGE), disk full, ...: no half-written index file
        if os.path.exists(path):
            os.remove(path)
        raise
    reindex = False
    if order is not None:
        try:
            _lib.fxi_bulk_index(path, root[index_name], packed_names, name_off, order)
        except _lib.FxError as e:                             # a name too long for an in-page index entry
            if e.code != _lib.FX_ERANGE:
                if os.path.exists(path):
                    os.remove(path)
                raise
            reindex = True
    db = connect(path)
    db.execute("PRAGMA synchronous = OFF")
    if reindex or order is None:
        db.execute("PRAGMA threads = %d" % int(threads))      # sorter threads of CREATE INDEX / REINDEX
    if reindex:
        db.execute("REINDEX %s" % index_name)                 # SQLite fills the (still empty) index from the table
    if order is None:
        try:
            db.execute(index_sql)
        except sqlite3.Error:
            pass
    return db


def _bulk_table_dev(path, blob, kind, ddl, table, index_sql, index_name, schema_done=False):
    """_bulk_table with the pages formatted ON THE DEVICE (Blob.fxi_dev_build, csrc/fx_fxi_dev.hpp): the record table, the
    names and their sorted order stay in HBM, only finished pages come to the file; the names are sorted while the table's
    leaves cross PCIe.  Duplicate names: the table alone is written and the empty index dropped again -- the reference ignores
    the failure of CREATE UNIQUE INDEX (index.c:363-366, fastq.c:152-156).  schema_done: the file exists with the tables of `ddl` in it (presize_fastq).  Returns (open
    connection, phases in seconds); on FX_ERANGE / FX_EINVAL (a row that needs an overflow page, a database that does not
    have 4 KiB pages) the file is removed and the error re-raised -- the caller falls back to the host loaders."""
    import time
    t0 = time.perf_counter()
    try:
        db = connect(path)
        if not schema_done:
            db.executescript(ddl)
        if not db.execute("SELECT count(*) FROM sqlite_master WHERE name=?", (index_name,)).fetchone()[0]:
            db.execute(index_sql)                             # (the empty index; dropped below if the names are not distinct)
        root = dict(db.execute("SELECT name, rootpage FROM sqlite_master").fetchall())
        db.commit()
        db.close()
        t1 = time.perf_counter()
        ndup, laps = blob.fxi_dev_build(kind, path, root[table], root[index_name])      # the sort runs beside the table's copy-out
        t2 = time.perf_counter()
    except BaseException:
        if os.path.exists(path):
            os.remove(path)
        raise
    db = connect(path)
    db.execute("PRAGMA synchronous = OFF")
    if ndup:                                                  # CREATE UNIQUE INDEX would have failed, and the reference ignores that
        db.execute("DROP INDEX %s" % index_name)
    laps["sqlite_schema"] = t1 - t0
    laps["write_call"] = t2 - t1
    laps["sqlite_reopen"] = time.perf_counter() - t2
    return db, laps


_KINDS = {1: (None, "read", "CREATE UNIQUE INDEX readidx ON read (name)", "readidx"),
          0: (None, "seq", "CREATE UNIQUE INDEX chromidx ON seq (chrom)", "chromidx")}


class PartsWriter:
    """ONE index file whose big table is held by SEVERAL handles -- the byte-range shards of a multi-GPU build, the devices
    of one process, the windows of a stream that does not fit -- with every page formatted on a device (fx_fxi_part_* /
    fx_fxi_join_*, csrc/fxgpu.hip): each part writes the table leaves of ITS rows into its own page range of the file, the
    names of all parts meet on the writer's device, are sorted once and become the index leaves there; the host writes
    the interior levels and the header.  What _bulk_table_dev does for one handle (fastq.c:29-60, 136-171; index.c:178-207,
    239-251, 363).  The parts are added IN ORDER: add_local(blob) for a handle of this process (its leaves are formatted and
    written here and now -- the blob may be closed afterwards), add_remote(...) for a part whose rank wrote its own leaves.
    finish() -> open connection (the caller adds the small tables).  Any FxError: the file is removed and the error
    re-raised -- FX_ERANGE / FX_EINVAL mean the caller should use the host loaders."""

    def __init__(self, path, kind, device, schema_done=False):
        import torch
        self._torch = torch
        self.path, self.kind, self.device = path, int(kind), int(device)
        ddl = FASTQ_DDL if self.kind == 1 else FASTA_DDL
        _, self.table, self.index_sql, self.index_name = _KINDS[self.kind]
        if os.path.exists(path) and not schema_done:
            os.remove(path)
        db = connect(path)
        if not schema_done:
            db.executescript(ddl)
        if not db.execute("SELECT count(*) FROM sqlite_master WHERE name=?", (self.index_name,)).fetchone()[0]:
            db.execute(self.index_sql)                        # (dropped again at the end if the names turn out not to be distinct)
        self.root = dict(db.execute("SELECT name, rootpage FROM sqlite_master").fetchall())
        self.first_new_page = int(db.execute("PRAGMA page_count").fetchone()[0]) + 1
        db.commit()
        db.close()
        self.rows = self.leaves = 0
        self._firsts, self._names, self._lens = [], [], []
        self.laps = {"table_kernels": 0.0, "table_to_file": 0.0}
        self._slack = False

    def reserve(self, leaves_total, rows_total=0, name_bytes_total=0, background=False):
        """Room for the table's pages -- and for the index, estimated from the rows and the bytes of their names -- before the
        parts write them (best effort; fx_fxi_join_grow: on tmpfs the pages are allocated at 16-18 GB/s when nothing else
        writes into the file).  background: in a thread (the call releases the interpreter lock); reserved() waits for it."""
        from . import _lib
        # an index entry: payload size, header (3), the name, the rowid (<= 4 bytes below 2^31 rows), its 2-byte cell pointer
        extra = int((int(name_bytes_total) + 11 * int(rows_total)) * 1.02) + (1 << 16) if rows_total else 0

        def grow():
            first = _lib.fxi_join_grow(self.path, self.root[self.table], int(leaves_total), self.device, extra)
            if first != self.first_new_page:
                raise RuntimeError("the index file %s changed under its writer" % self.path)
        self._grow_err = None
        if not background:
            return grow()
        import threading

        def run():
            try:
                grow()
            except BaseException as e:                        # noqa: BLE001
                self._grow_err = e
        self._grower = threading.Thread(target=run)
        self._grower.start()

    def reserved(self):
        t = getattr(self, "_grower", None)
        if t is not None:
            t.join()
            self._grower = None
            if self._grow_err is not None:
                raise self._grow_err

    def dev_buffers(self, n, name_bytes, device=None):
        t = self._torch
        d = t.device("cuda", self.device if device is None else int(device))
        return t.empty(int(name_bytes) + 64, dtype=t.uint8, device=d), t.empty(max(int(n), 1), dtype=t.int32, device=d)

    def add_local(self, blob, device=None):
        """A part held by a handle of this process: shape, leaves to the file, names to the writer's device."""
        try:
            n, nleaf, nb = blob.fxi_part_shape(self.kind, self.rows)
            if n == 0:
                return 0
            lp = blob.fxi_part_leaves(self.kind, self.path, self.first_new_page, self.leaves)
            for k, v in lp.items():
                self.laps[k] += v
            names, lens = self.dev_buffers(n, nb, device)
            blob.fxi_part_names(self.kind, names.data_ptr(), lens.data_ptr())
            names[nb:].zero_()
            self.add_remote(n, nleaf, blob.fxi_part_firsts(nleaf), names[:nb], lens[:n], slack=True)
            return n
        except BaseException:
            self.abort()
            raise

    def add_remote(self, n, nleaf, firsts, names_dev, lens_dev, slack=False):
        """A part whose own rank formatted and wrote its leaves (at leaf_base = self.leaves as it was before this call):
        its first rows (global, 0-based), its names and their lengths as tensors on a device."""
        if int(n) == 0:
            return
        self._firsts.append(None if firsts is None else np.asarray(firsts, dtype=np.int64))      # (None: set_firsts brings them later)
        self._slack = bool(slack)                             # (64 zero bytes lie behind names_dev in its own storage)
        self._names.append(names_dev)
        self._lens.append(lens_dev)
        self.rows += int(n)
        self.leaves += int(nleaf)

    def set_firsts(self, k, firsts):
        self._firsts[int(k)] = np.asarray(firsts, dtype=np.int64)

    def join_names(self):
        """All parts' names are here: onto the writer's device, back to back, sorted (fx_fxi_join_begin).  May run while the parts
        still write their leaves; finish() does it if nobody has."""
        from . import _lib
        t = self._torch
        try:
            dev = t.device("cuda", self.device)
            names = lens = None
            t0 = time.perf_counter()
            if self.rows:
                # (the sort reads whole words behind the last name: 64 zero bytes follow it -- a lone part's buffer has them, dev_buffers)
                pad = t.zeros(64, dtype=t.uint8, device=dev)
                names = self._names[0] if len(self._names) == 1 and self._names[0].device == dev and self._slack else t.cat([x.to(dev, non_blocking=True) for x in self._names] + [pad])
                lens = self._lens[0].to(dev) if len(self._lens) == 1 else t.cat([x.to(dev, non_blocking=True) for x in self._lens])
                self._names, self._lens = [], []
                t.cuda.synchronize(dev)
            t1 = time.perf_counter()
            self._joined = (names, lens)                      # (kept alive until the index leaves are formatted from them)
            self._join = _lib.FxiJoin(self.device, names.data_ptr() if names is not None else 0, lens.data_ptr() if lens is not None else 0, self.rows)
            self.laps["names_to_device"] = t1 - t0
            self.laps["name_sort"] = time.perf_counter() - t1
        except BaseException:
            self.abort()
            raise

    def abort(self):
        j = getattr(self, "_join", None)
        if j is not None:
            j.close()
            self._join = None
        self._joined = None
        self._names, self._lens = [], []
        if os.path.exists(self.path):
            os.remove(self.path)

    def finish(self):
        from . import _lib
        t = self._torch
        try:
            self.reserved()
            dev = t.device("cuda", self.device)
            root_table, root_index = self.root[self.table], self.root[self.index_name]
            if self.leaves == 1:                              # a table of ONE leaf lives in its root page: the part wrote it as the first new page
                fd = os.open(self.path, os.O_RDWR)
                try:
                    os.pwrite(fd, os.pread(fd, 4096, (self.first_new_page - 1) * 4096), (root_table - 1) * 4096)
                finally:
                    os.close(fd)
            if getattr(self, "_join", None) is None:
                self.join_names()
            j = self._join
            try:
                firsts = np.concatenate(self._firsts) if self._firsts else np.zeros(0, dtype=np.int64)
                lp = j.write(self.path, root_table, 0 if j.n_dup else root_index, self.rows, firsts, self.first_new_page)
            finally:
                j.close()
                self._join = self._joined = None
            self.laps.update(lp)
            self.n_dup = j.n_dup
        except BaseException:
            self.abort()
            raise
        db = connect(self.path)
        db.execute("PRAGMA synchronous = OFF")
        if self.n_dup:                                        # CREATE UNIQUE INDEX would have failed, and the reference ignores that (fastq.c:152-156)
            db.execute("DROP INDEX %s" % self.index_name)
        return db


def _varint_len(v):
    n = 1
    while v > 0x7F and n < 9:
        v >>= 7
        n += 1
    return n


def _int_bytes(v):
    """Bytes of the body of an INTEGER value in a record (fileformat2 2.1)."""
    if v in (0, 1):
        return 0
    for lim, nb in ((127, 1), (32767, 2), (8388607, 3), (2147483647, 4), (140737488355327, 6)):
        if v <= lim:
            return nb
    return 8


def _fastq_window_shape(buf, at_start, full_name):
    """Mean (bytes per record, name length, header-line length, read length) of the four-line records in a window of a
    FASTQ file; the window may begin anywhere (at_start: at the first byte of the file).  None when no phase of four makes
    every complete record of the window look like one ('@' line, then a '+' line two lines on)."""
    import numpy as np
    nl = np.flatnonzero(buf == 10)
    if nl.size < 12:
        return None
    starts = nl[:-1] + 1                                      # starts[i] = first byte of the line that ends at nl[i + 1]
    if at_start:
        starts = np.concatenate([[0], starts])
        ends = nl
    else:
        ends = nl[1:]
    m = starts.size
    first = buf[starts]
    for ph in range(1 if at_start else 4):
        k = (m - ph) // 4
        if k < 2:
            continue
        h = starts[ph: ph + 4 * k: 4]
        if (first[ph: ph + 4 * k: 4] == ord("@")).all() and (first[ph + 2: ph + 4 * k: 4] == ord("+")).all():
            he = ends[ph: ph + 4 * k: 4]
            ss, se = starts[ph + 1: ph + 4 * k: 4], ends[ph + 1: ph + 4 * k: 4]
            per = float(ends[ph + 4 * k - 1] + 1 - h[0]) / k
            kk = min(k, 1024)
            L = dl = 0.0
            for s0, e0 in zip(h[:kk].tolist(), he[:kk].tolist()):
                line = bytes(buf[s0 + 1:e0])
                if line.endswith(b"\r"):
                    line = line[:-1]
                dl += len(line)
                if not full_name:
                    for sep in (b" ", b"\t"):
                        i = line.find(sep)
                        if i >= 0:
                            line = line[:i]
                L += len(line)
            cr = int(buf[se[0] - 1] == 13) if se[0] > 0 else 0
            return per, L / kk, dl / kk, float((se[:kk] - ss[:kk]).mean()) - cr
    return None


def estimate_fastq_index_bytes(path, full_name=False, head=1 << 19, places=8):
    """How large the index file of a plain FASTQ file will be, from `places` windows of `head` bytes spread over it:
    reads are taken as four lines, the name ends at the first white space; cells and entries are sized as
    fx_fxi_dev.hpp sizes them, with the offsets and rowids of each window's place in the file.  None when a window
    does not look like four-line records.  (Only used to set room aside early -- presize_fastq; a wrong guess costs
    time, never correctness.)"""
    import numpy as np
    size = os.path.getsize(path)
    if size < 64:
        return None
    shapes = []
    with open(path, "rb") as f:
        for j in range(places):
            off = 0 if j == 0 else min(int(size * j / places), max(size - head, 0))
            f.seek(off)
            sh = _fastq_window_shape(np.frombuffer(f.read(head), dtype=np.uint8), off == 0, full_name)
            if sh is None:
                return None
            shapes.append((off, sh))
            if size <= head:
                break
    per_mean = sum(sh[0] for _, sh in shapes) / len(shapes)
    n = size / per_mean
    pages = 0.0
    for j, (off, (per, L, dl, rl)) in enumerate(shapes):
        mid = int(off + min(size / len(shapes), size - off) / 2)
        rid = max(int(n * mid / size), 1)
        tl = _varint_len(13 + 2 * int(round(L)))
        payload = 2 + tl + 4 + L + _int_bytes(int(dl)) + _int_bytes(int(rl)) + 2 * _int_bytes(mid)
        cell = _varint_len(int(payload)) + _varint_len(rid) + payload + 2
        ent_payload = 1 + tl + 1 + L + _int_bytes(rid)
        ent = _varint_len(int(ent_payload)) + ent_payload + 2
        rows = n / len(shapes)
        # (the leaves are filled in chunks of 2048 rows, each starting a fresh page -- k_fxi_fill: whole pages per chunk)
        pages += rows / 2048.0 * (math.ceil(2048.0 / int(4088 / cell)) + math.ceil(2048.0 / (int(4088 / ent) + 1)))
    pages *= 1.0 + 1.0 / 250                                  # interior levels
    return int(pages * 4096)


def presize_fastq(path, input_path, full_name=False, device=-1, with_index=False):
    """The index file of a LARGE plain FASTQ input created early -- schema in place -- and grown in the background to
    101 % of its estimated size while the input is staged (fx_fxi_presize_begin; FX_FXI_PRESIZE_FRAC: with 98.5 % the
    missing pages were allocated between the table and the index, 12 ms of C3's constructor; what is left over is cut off).  -> token for _lib.fxi_presize_end, or
    None when nothing was done (a small input, no estimate).  The caller removes the file if the build fails."""
    from . import _lib
    if os.path.getsize(input_path) < int(os.environ.get("FX_FXI_PRESIZE_MIN", 1 << 30)):
        return None
    est = estimate_fastq_index_bytes(input_path, full_name)
    if not est:
        return None
    try:
        st = os.statvfs(os.path.dirname(os.path.abspath(path)) or ".")
        if st.f_bavail * st.f_frsize < est * 1.1 + (64 << 20):
            return None
    except OSError:
        return None
    db = connect(path)
    db.executescript(FASTQ_DDL)
    if with_index:                                            # (the pipelined build: nothing but pages is written while the room is being made)
        db.execute(_KINDS[1][2])
    db.close()
    try:
        return _lib.fxi_presize_begin(path, int(est * float(os.environ.get("FX_FXI_PRESIZE_FRAC", "1.01"))), device)
    except _lib.FxError:
        if os.path.exists(path):                              # (the build must find no file: it makes its own)
            os.remove(path)
        return None


def write_fastq_dev(path, blob, n, size, schema_done=False):
    """The result of write_fastq (fastq.c:76-171) through _bulk_table_dev: `read` and `readidx` from the device."""
    db, laps = _bulk_table_dev(path, blob, 1, FASTQ_DDL, "read", "CREATE UNIQUE INDEX readidx ON read (name)", "readidx", schema_done)
    avg = size * 1.0 / n if n else float("nan")              # fastq.c:161
    db.execute("INSERT INTO stat VALUES (?,?,?)", (int(n), int(size), avg))
    return db, laps


def write_fasta_dev(path, blob, n, seqlen_total):
    """The result of write_fasta (index.c:226-251, 342-372) through _bulk_table_dev: `seq` and `chromidx` from the device."""
    db, laps = _bulk_table_dev(path, blob, 0, FASTA_DDL, "seq", "CREATE UNIQUE INDEX chromidx ON seq (chrom)", "chromidx")
    db.execute("INSERT INTO stat (seqnum,seqlen) VALUES (?,?)", (int(n), int(seqlen_total)))
    return db, laps


def write_fastq_bulk(path, packed_names, name_off, cols, size, order=None, threads=8):
    """The result of write_fastq (fastq.c:76-171) through _bulk_table.  packed_names uint8 + name_off int64[n+1]:
    the read names back to back; cols as for write_fastq."""
    db = _bulk_table(path, FASTQ_DDL, "read", "CREATE UNIQUE INDEX readidx ON read (name)", "readidx", packed_names, name_off,
                     [cols["dlen"], cols["rlen"], cols["soff"], cols["qoff"]], order, threads)
    n = len(name_off) - 1
    avg = size * 1.0 / n if n else float("nan")              # fastq.c:161
    db.execute("INSERT INTO stat VALUES (?,?,?)", (n, int(size), avg))
    return db


def write_fasta_bulk(path, packed_names, name_off, cols, seqlen_total, order=None, threads=8):
    """The result of write_fasta (index.c:226-251, 342-372) through _bulk_table."""
    db = _bulk_table(path, FASTA_DDL, "seq", "CREATE UNIQUE INDEX chromidx ON seq (chrom)", "chromidx", packed_names, name_off,
                     [cols[k] for k in ("boff", "blen", "slen", "llen", "elen", "norm", "dlen")], order, threads)
    db.execute("INSERT INTO stat (seqnum,seqlen) VALUES (?,?)", (len(name_off) - 1, int(seqlen_total)))
    return db


def write_fastq_comp(db, base, meta):
    """fastq.c:755-790 (meta column order: maxlen, minlen, minqs, maxqs, phred)."""
    db.execute("INSERT INTO base VALUES (?,?,?,?,?)", tuple(int(x) for x in base))
    db.execute("INSERT INTO meta VALUES (?,?,?,?,?)", tuple(int(x) for x in meta))


def write_gzindex(db, compressed_size, uncompressed_size, cmp_off=(), uncmp_off=(), spacing=1048576, window=32768, bits=None,
                  has_data=None, windows=None):
    """zran export layout of util.c:461-529: "GZIDX", version 1, flags, compressed_size,
    uncompressed_size, spacing, window_size, npoints, then per point cmp_offset (u64),
    uncmp_offset (u64), bits (u8), has-data flag (u8); one window row per point with data.
    BGZF restart points sit on member boundaries: bits = 0 and no 32 KiB window is needed
    (has-data = 0, which version-1 importers accept, util.c:621-651).  Single-stream gzip: the
    points captured while the stream was inflated (fx_gz_checkpoints) with their bits and windows --
    what the next open inflates in parallel from (read_gzindex -> fx_open_file_indexed).  Checkpoint
    placement is not asserted by any reference test; offsets, bits and windows are: the compiled reference reads through them)."""
    n = len(cmp_off)
    bits = [0] * n if bits is None else [int(x) for x in bits]
    has_data = [0] * n if has_data is None else [int(x) for x in has_data]
    db.execute("BEGIN TRANSACTION")
    rows = [b"GZIDX", struct.pack("<B", 1), struct.pack("<B", 0), struct.pack("<Q", compressed_size),
            struct.pack("<Q", uncompressed_size), struct.pack("<I", spacing), struct.pack("<I", window),
            struct.pack("<I", n)]
    for c, u, bt, hd in zip(cmp_off, uncmp_off, bits, has_data):
        rows += [struct.pack("<Q", int(c)), struct.pack("<Q", int(u)), struct.pack("<B", bt), struct.pack("<B", hd)]
    if windows is not None and sum(has_data):
        w = memoryview(windows).cast("B") if not isinstance(windows, (bytes, bytearray)) else memoryview(windows)
        rows += [bytes(w[k * window:(k + 1) * window]) for k in range(sum(has_data))]
    db.executemany("INSERT INTO gzindex VALUES (NULL,?)", [(sqlite3.Binary(r),) for r in rows])
    db.execute("COMMIT")


def read_gzindex(db):
    """The restart points of a gzindex table written by write_gzindex (or by the reference: the same rows) -> dict
    compressed_size, uncompressed_size, cmp, uncmp, bits, has, windows (numpy uint8, 32768 per point with data; None
    when a row is missing); None when the table holds no points."""
    import numpy as np
    try:
        rows = [bytes(r[0]) for r in db.execute("SELECT content FROM gzindex ORDER BY ID")]
    except sqlite3.Error:
        return None
    if len(rows) < 8 or rows[0] != b"GZIDX" or rows[1][0] > 1:
        return None
    csize, usize = struct.unpack("<Q", rows[3])[0], struct.unpack("<Q", rows[4])[0]
    window, n = struct.unpack("<I", rows[6])[0], struct.unpack("<I", rows[7])[0]
    if n == 0 or len(rows) < 8 + 4 * n:
        return None
    cmp_ = np.array([struct.unpack("<Q", rows[8 + 4 * i])[0] for i in range(n)], dtype=np.int64)
    unc = np.array([struct.unpack("<Q", rows[9 + 4 * i])[0] for i in range(n)], dtype=np.int64)
    bits = np.array([rows[10 + 4 * i][0] for i in range(n)], dtype=np.uint8)
    has = np.array([rows[11 + 4 * i][0] for i in range(n)], dtype=np.uint8)
    wrows = rows[8 + 4 * n:]
    windows = None
    if window == 32768 and len(wrows) == int(has.sum()) and all(len(w) == window for w in wrows):
        windows = np.frombuffer(b"".join(wrows), dtype=np.uint8) if wrows else np.zeros(0, dtype=np.uint8)
    return {"compressed_size": csize, "uncompressed_size": usize, "cmp": cmp_, "uncmp": unc, "bits": bits, "has": has, "windows": windows}


def has_fasta_index(db):
    """pyfastx_load_index's sanity check (index.c:402-411)."""
    try:
        return db.execute("SELECT * FROM seq LIMIT 1").fetchone() is not None
    except sqlite3.Error:
        return False


def exists(path):
    return path != ":memory:" and os.path.exists(path)
