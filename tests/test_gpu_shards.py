"""-m gpu: byte-range shards on ONE GPU (G logical shards, same kernels, same
summary + stitch path as the multi-GPU build; the all-gather is degenerate)."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

COLS = ("hoff", "boff", "blen", "slen", "llen", "elen", "norm", "dlen", "name_len")


def sharded_rows(L, raw, cuts, full_name=False):
    """Build every shard with the HIP path, stitch on the host, return concatenated rows."""
    from pyfastx_amd import shard
    bounds = [0] + list(cuts) + [len(raw)]
    blobs, S, rows = [], [], []
    for i in range(len(bounds) - 1):
        lo, hi = bounds[i], bounds[i + 1]
        b = L.Blob.from_bytes(raw[lo:hi])
        b.set_shard(lo, raw[lo - 1] if lo else 10, hi == len(raw))
        s = b.fasta_build(full_name)
        blobs.append((b, s.n_seq))
        S.append(b.shard_summary())
    for r, (b, n) in enumerate(blobs):
        row = shard.stitch_tail(S, r, full_name)
        if row is not None:
            b.fasta_set_row(n - 1, **row)
        rows.append(dict(b.fasta_table(n), reg=b.fasta_line_regular(n)))
    return {c: np.concatenate([t[c] for t in rows]) for c in COLS + ("reg",)}, S


def sharded_rows_dev(L, raw, cuts, full_name=False):
    """Same, with the device-resident exchange: summaries written by fx_shard_summary_dev into one device
    buffer (what the all-gather would assemble), last records finished by fx_fasta_stitch_dev."""
    import torch
    bounds = [0] + list(cuts) + [len(raw)]
    world = len(bounds) - 1
    allS = torch.zeros(world * 28, dtype=torch.int64, device="cuda:0")
    blobs = []
    for i in range(world):
        lo, hi = bounds[i], bounds[i + 1]
        b = L.Blob.from_bytes(raw[lo:hi])
        b.set_shard(lo, raw[lo - 1] if lo else 10, hi == len(raw))
        s = b.fasta_build(full_name)
        b.shard_summary_dev(allS.data_ptr() + i * 28 * 8)
        b.sync()
        blobs.append((b, s.n_seq))
    rows = []
    for r, (b, n) in enumerate(blobs):
        b.fasta_stitch_dev(allS.data_ptr(), world, r, full_name)
        rows.append(dict(b.fasta_table(n), reg=b.fasta_line_regular(n)))
    return {c: np.concatenate([t[c] for t in rows]) for c in COLS + ("reg",)}


def check(oracle, L, raw, cuts, full_name=False):
    recs, tot = oracle.fasta_index(raw, full_name=full_name)
    got, S = sharded_rows(L, raw, cuts, full_name)
    assert len(got["boff"]) == len(recs), (cuts, len(got["boff"]), len(recs))
    for c in COLS:
        np.testing.assert_array_equal(got[c], recs[c].astype(got[c].dtype), err_msg="%s cuts=%s" % (c, cuts))
    from test_host_logic import _reg_of                     # line-regular: decided across the cuts from the summaries
    np.testing.assert_array_equal(got["reg"], np.array([_reg_of(raw, r) for r in recs], dtype=np.int32), err_msg="reg cuts=%s" % (cuts,))
    dev = sharded_rows_dev(L, raw, cuts, full_name)
    for c in COLS + ("reg",):
        np.testing.assert_array_equal(dev[c], got[c], err_msg="device stitch: %s cuts=%s" % (c, cuts))


@pytest.fixture(scope="module")
def L():
    from pyfastx_amd import _lib
    assert _lib.lib().fx_device_count() >= 1
    return _lib


def test_every_cut_of_every_edge_case(oracle, L):
    g = load_golden("fasta_edge")
    for name, case in g.items():
        if name.endswith(":upper") or name in ("single_long_line", "wide_then_narrow"):
            continue
        raw = case["text"].encode()
        if not raw.lstrip().startswith(b">"):
            continue
        for c in range(1, len(raw)):
            check(oracle, L, raw, [c])
        for c in range(1, len(raw) - 1, 3):          # three shards, the middle one tiny (often no newline at all)
            check(oracle, L, raw, [c, c + 1])
            if c + 4 < len(raw):
                check(oracle, L, raw, [c, c + 4])


@pytest.mark.parametrize("seed", range(8))
def test_random_fasta_random_cuts(oracle, L, seed):
    from test_gpu_kernels import _rand_fasta
    rng = np.random.default_rng(100 + seed)
    raw = _rand_fasta(rng, int(rng.integers(1, 40)), int(rng.integers(3, 90)), crlf=bool(seed & 1),
                      ragged=(seed % 4 == 2), trailing=(seed != 5), lower=True)
    for g in (2, 3, 5, 8):
        cuts = sorted(set(int(x) for x in rng.integers(1, len(raw), g - 1)))
        check(oracle, L, raw, cuts, full_name=bool(seed & 2))


def test_long_lines_across_many_shards(oracle, L):
    raw = (">one  description here\n" + "ACGTTGCA" * 5000 + "\n>two\n" + "GATTACA" * 3000 + "\nAC\n>three\n").encode()
    n = len(raw)
    check(oracle, L, raw, [n // 8 * i for i in range(1, 8)])
    check(oracle, L, raw, [5, 11, 30, 40000, 40010, n - 3])


def _long_header_fasta():
    """A header line of ~100 KiB: the name is 70 000 bytes long, the first white space sits 70 001 bytes behind the '>'."""
    name = ("N" + "x" * 69_999).encode()
    return b">a d\nACGTAC\nGT\n>" + name + b" " + b"d" * 30_000 + b"\nACGTACGT\nACGTACGT\nAC\n>" + b"y" * 100_000 + b"\nGG\n>z\nA\n"


def test_header_lines_of_100_kib_across_cuts(oracle, L):
    """No limit on how far behind a cut a name may end (round 2: 64 KiB, ValueError beyond): cuts at every kind of place of two
    ~100 KiB header lines -- inside the name, at its last byte, at the white space, inside the description, around the line end
    --, two and three shards, with and without full_name; a third shard that holds nothing but a middle piece of the line."""
    raw = _long_header_fasta()
    h1 = raw.index(b">N")
    h2 = raw.index(b">y")
    places = [h1 + 1, h1 + 2, h1 + 35_000, h1 + 69_999, h1 + 70_000, h1 + 70_001, h1 + 70_002, h1 + 85_000, h1 + 100_001, h1 + 100_002,
              h1 + 100_003, h2 + 1, h2 + 50_000, h2 + 100_000, h2 + 100_001, h2 + 100_002]
    for c in places:
        check(oracle, L, raw, [c])
        check(oracle, L, raw, [c], full_name=True)
    for a, b in ((h1 + 10, h1 + 66_000), (h1 + 10, h1 + 70_001), (h1 + 69_000, h1 + 99_000), (h2 + 5, h2 + 99_999), (h1 - 3, h2 + 70_000)):
        check(oracle, L, raw, [a, b])
        check(oracle, L, raw, [a, (a + b) // 2, b], full_name=True)


# ------------------------------------------------------------------ FASTQ shards (halo + line numbering)
def _rand_fastq(rng, n, crlf=False, trailing=True):
    eol = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(n):
        ln = int(rng.integers(1, 200))
        seq = bytes(rng.choice(list(b"ACGTNacgt"), ln).astype(np.uint8))
        qual = bytes(rng.integers(35, 71, ln).astype(np.uint8))
        name = b"@r%d" % i + (b" extra words %d" % (i * 7) if i % 3 else b"")
        out += [name + eol, seq + eol, b"+" + (name[1:] if i % 5 == 0 else b"") + eol, qual + eol]
    raw = b"".join(out)
    return raw if trailing else raw[:-len(eol)]


def fastq_sharded(L, raw, cuts, halo=4096):
    from pyfastx_amd import shard
    bounds = [0] + list(cuts) + [len(raw)]
    blobs, cores = [], []
    for i in range(len(bounds) - 1):
        lo, hi = bounds[i], bounds[i + 1]
        h = min(halo, len(raw) - hi)
        b = L.Blob.from_bytes(raw[lo:hi + h])
        b.set_shard(lo, raw[lo - 1] if lo else 10, hi == len(raw))
        b.set_halo(h)
        cores.append(b.fastq_scan())
        blobs.append(b)
    cols = {k: [] for k in ("name_off", "name_len", "dlen", "rlen", "soff", "qoff")}
    size, nreads, base, meta, next_id = 0, 0, np.zeros(5, np.int64), None, 0
    for b, (loff, prev) in zip(blobs, shard.fastq_contexts(cores)):
        s = b.fastq_build_ctx(loff, prev)
        if s.n_reads:
            assert s.first_id == next_id
        next_id += s.n_reads
        t = b.fastq_table(s.n_reads)
        for k in cols:
            cols[k].append(t[k])
        size += s.size
        nreads += s.n_reads
        bs, mt = b.fastq_comp()
        base += bs
        meta = mt.copy() if meta is None else np.array([max(meta[0], mt[0]), min(meta[1], mt[1]), min(meta[2], mt[2]),
                                                        max(meta[3], mt[3]), 0])
    return {k: np.concatenate(v) for k, v in cols.items()}, size, nreads, base, meta


@pytest.mark.parametrize("seed", range(6))
def test_fastq_shards(oracle, L, seed):
    rng = np.random.default_rng(500 + seed)
    raw = _rand_fastq(rng, int(rng.integers(5, 300)), crlf=bool(seed & 1), trailing=(seed != 4))
    if seed == 5:
        raw += b"@tail\nACGT\n+\n"                       # incomplete trailing record
    recs, size, ln = oracle.fastq_index(raw)
    c = oracle.fastq_composition(raw)
    for g in (1, 2, 3, 7):
        cuts = sorted(set(int(x) for x in rng.integers(1, len(raw), g - 1)))
        got, gsize, n, base, meta = fastq_sharded(L, raw, cuts)
        assert n == len(recs) and gsize == size, (seed, cuts)
        for k in got:
            np.testing.assert_array_equal(got[k], recs[k].astype(got[k].dtype), err_msg="%s cuts=%s" % (k, cuts))
        assert base.tolist() == [c["a"], c["c"], c["g"], c["t"], c["n"]], (seed, cuts)
        assert meta[:4].tolist() == [c["maxlen"], c["minlen"], c["minqs"], c["maxqs"]], (seed, cuts)


def test_fastq_every_cut_small(oracle, L):
    raw = b"@r1 d1\r\nACGT\r\n+\r\nIIII\r\n@r2\r\nGGCCA\r\n+r2\r\n#!5AB\r\n@r3 x y\r\nA\r\n+\r\nI\r\n"
    recs, size, ln = oracle.fastq_index(raw)
    oc = oracle.fastq_composition(raw)
    for c in range(1, len(raw)):
        got, gsize, n, base, meta = fastq_sharded(L, raw, [c], halo=64)
        assert n == len(recs) and gsize == size, c
        for k in got:
            np.testing.assert_array_equal(got[k], recs[k].astype(got[k].dtype), err_msg="%s cut=%d" % (k, c))
        # composition across every cut (round 4): five sums, two minima, two maxima -- each read counted by the shard that owns it
        assert base.tolist() == [oc["a"], oc["c"], oc["g"], oc["t"], oc["n"]], c
        assert meta[:4].tolist() == [oc["maxlen"], oc["minlen"], oc["minqs"], oc["maxqs"]], c


def test_fastq_halo_too_small_is_an_error(L):
    raw = _rand_fastq(np.random.default_rng(1), 50)
    with pytest.raises(L.FxError) as e:
        fastq_sharded(L, raw, [len(raw) // 2], halo=3)
    assert e.value.code == L.FX_ERANGE


def test_device_exchange_stream_ordering(oracle, L):
    """The plumbing of ShardedFasta.build's device path on one GPU: summaries are produced on the library's
    stream, 'gathered' by torch on ITS stream (a plain copy stands in for the RCCL all-gather), stitched on
    the library's stream again -- ordered by stream events only (torch ExternalStream), no host sync."""
    import torch
    from test_gpu_kernels import _rand_fasta
    rng = np.random.default_rng(7)
    raw = _rand_fasta(rng, 12, 61, crlf=False, ragged=False, trailing=True, lower=True)
    cuts = sorted(set(int(x) for x in rng.integers(1, len(raw), 3)))
    bounds = [0] + cuts + [len(raw)]
    world = len(bounds) - 1
    dev = torch.device("cuda", 0)
    allS = torch.zeros(world * 28, dtype=torch.int64, device=dev)
    cur = torch.cuda.current_stream(dev)
    blobs = []
    for i in range(world):
        lo, hi = bounds[i], bounds[i + 1]
        b = L.Blob.from_bytes(raw[lo:hi])
        b.set_shard(lo, raw[lo - 1] if lo else 10, hi == len(raw))
        s = b.fasta_build()
        ext = torch.cuda.ExternalStream(b.stream, device=dev)
        mine = torch.zeros(28, dtype=torch.int64, device=dev)
        b.shard_summary_dev(mine.data_ptr())
        cur.wait_stream(ext)
        allS[i * 28:(i + 1) * 28].copy_(mine)               # stands in for all_gather_into_tensor
        blobs.append((b, s.n_seq, ext, mine))
    rows = []
    for r, (b, n, ext, _) in enumerate(blobs):
        ext.wait_stream(cur)
        b.fasta_stitch_dev(allS.data_ptr(), world, r)
        rows.append(b.fasta_table(n))
    got = {c: np.concatenate([t[c] for t in rows]) for c in COLS}
    recs, _ = oracle.fasta_index(raw)
    for c in COLS:
        np.testing.assert_array_equal(got[c], recs[c].astype(got[c].dtype), err_msg=c)


def sharded_comp(L, raw, cuts, fused=False):
    """k_fasta_comp on every shard + the exchange of shard.py (lead_from / lead rows), with G logical shards."""
    from pyfastx_amd import shard
    bounds = [0] + list(cuts) + [len(raw)]
    blobs, S = [], []
    for i in range(len(bounds) - 1):
        lo, hi = bounds[i], bounds[i + 1]
        b = L.Blob.from_bytes(raw[lo:hi])
        b.set_shard(lo, raw[lo - 1] if lo else 10, hi == len(raw))
        s = b.fasta_build(False, comp=fused)                 # fused: the composition counters ride on the scan
        blobs.append((b, s.n_seq))
        S.append(b.shard_summary())
    last_boff = []
    for r, (b, n) in enumerate(blobs):
        row = shard.stitch_tail(S, r, False)
        if row is not None:
            b.fasta_set_row(n - 1, **row)
        last_boff.append(int(b.fasta_table(n)["boff"][-1]) if n else -1)
    bases, nh = bounds[:-1], [n for _, n in blobs]
    comps, leads = [], []
    for r, (b, n) in enumerate(blobs):
        c, lead = b.fasta_comp_shard(n, shard.comp_lead_from(bases, last_boff, r))
        comps.append(c); leads.append(lead)
    out = [shard.comp_fold_leads(comps[r], leads, nh, r) for r in range(len(blobs))]
    return np.concatenate([c for c in out if len(c)]) if any(len(c) for c in out) else np.zeros((0, 128), dtype=np.int64)


def test_composition_across_cuts(oracle, L):
    """Per-record composition with the stream cut into shards equals the composition of the whole stream: every cut
    of the edge cases (header lines, CRLF, blank lines crossing the cut), two adjacent cuts, and random cuts of a
    larger random file whose records span several shards."""
    for name, case in load_golden("fasta_edge").items():
        if name.endswith(":upper"):
            continue
        raw = case["text"].encode()
        if len(raw) > 300 or name in ("single_long_line", "wide_then_narrow") or not raw.lstrip().startswith(b">"):
            continue
        n = len(oracle.fasta_index(raw)[0])
        want = oracle.fasta_comp(raw, n)
        for c in range(1, len(raw)):
            np.testing.assert_array_equal(sharded_comp(L, raw, [c]), want, err_msg="%s cut %d" % (name, c))
        for c in range(1, len(raw) - 3, 3):
            np.testing.assert_array_equal(sharded_comp(L, raw, [c, c + 2]), want, err_msg="%s cuts %d,%d" % (name, c, c + 2))
    rng = np.random.default_rng(8)
    parts = []
    for i in range(9):
        parts.append(b">rec%d some description that is long enough to cross a cut now and then\n" % i)
        s = np.frombuffer(b"ACGTNacgtnRY", dtype=np.uint8)[rng.integers(0, 12, int(rng.integers(1, 90000)))].tobytes()
        parts += [s[p:p + 60] + b"\n" for p in range(0, len(s), 60)]
    raw = b"".join(parts)
    want = oracle.fasta_comp(raw, 9)
    for g in (2, 3, 5, 16):
        for _ in range(4):
            cuts = sorted(set(int(x) for x in rng.integers(1, len(raw), g - 1)))
            np.testing.assert_array_equal(sharded_comp(L, raw, cuts), want, err_msg=str(cuts))
            np.testing.assert_array_equal(sharded_comp(L, raw, cuts, fused=True), want, err_msg="fused " + str(cuts))


def test_mixed_shapes_in_shards(oracle, L):
    """A 20 MB stream of every FASTA shape cut into 2 .. 16 shards at random offsets: the stitched rows (host and
    device stitch) and the composition folded across the cuts equal those of the whole stream."""
    from test_gpu_kernels import _mixed_fasta
    rng = np.random.default_rng(31)
    raw = _mixed_fasta(rng, nbig=6, ntiny=8000)
    n = len(oracle.fasta_index(raw)[0])
    want = oracle.fasta_comp(raw, n)
    for g in (2, 3, 8, 16):
        cuts = sorted(set(int(x) for x in rng.integers(1, len(raw), g - 1)))
        check(oracle, L, raw, cuts)
        np.testing.assert_array_equal(sharded_comp(L, raw, cuts), want, err_msg=str(cuts))
        np.testing.assert_array_equal(sharded_comp(L, raw, cuts, fused=True), want, err_msg="fused " + str(cuts))


@pytest.mark.parametrize("seed", range(4))
def test_fetch_over_shards(oracle, L, seed):
    """SURVEY 8e "Fetch" through the HIP kernels: every shard a Blob over its own bytes, queries routed by
    shard.ShardFetcher (the same object the multi-GPU run uses, all shards held by this process), answers equal to the
    oracle's on the whole stream -- including the queries that cross a cut and records that are not line-regular."""
    from pyfastx_amd import shard
    from test_host_logic import _shard_queries, _expected_fetch
    from test_gpu_kernels import _rand_fasta
    rng = np.random.default_rng(4200 + seed)
    raw = _rand_fasta(rng, 30, int(rng.integers(20, 90)), crlf=bool(seed & 1), ragged=False, trailing=(seed != 2), lower=True)
    raw += _rand_fasta(rng, 4, 50, crlf=bool(seed & 1), ragged=True, trailing=True)      # a few norm=0 records
    from test_gpu_kernels import _odd_line_fasta
    raw += _odd_line_fasta(rng, bool(seed & 1))                                           # norm=1 records with one odd line
    recs, _ = oracle.fasta_index(raw)
    G = (2, 4, 8, 16)[seed]
    cuts = sorted(set(int(x) for x in rng.integers(1, len(raw) - 1, G - 1)))
    got, _ = sharded_rows(L, raw, cuts)
    table = {k: got[k] for k in ("boff", "blen", "slen", "llen", "elen", "norm", "reg")}
    bases, ends = [0] + cuts, cuts + [len(raw)]
    blobs = {}
    for r in range(len(bases)):
        b = L.Blob.from_bytes(raw[bases[r]:ends[r]])
        b.set_shard(bases[r], raw[bases[r] - 1] if bases[r] else 10, ends[r] == len(raw))
        blobs[r] = b
    ids, st, sp, fl = _shard_queries(rng, recs, 3000)
    for c in cuts:                                            # windows across every cut that lies inside a sequence
        i = int(np.searchsorted(recs["boff"], c, "right")) - 1
        if i >= 0 and recs["slen"][i] > 300 and recs["norm"][i] and recs["boff"][i] < c < recs["boff"][i] + recs["blen"][i]:
            mid = int((c - recs["boff"][i]) // int(recs["llen"][i])) * int(recs["llen"][i] - recs["elen"][i])
            mid = min(mid, int(recs["slen"][i]) - 1)
            ids, st, sp = np.append(ids, i), np.append(st, max(mid - 150, 0)), np.append(sp, min(mid + 150, int(recs["slen"][i])))
            fl = np.append(fl, np.uint8(rng.integers(0, 8)))
    f = shard.ShardFetcher(blobs, bases, ends, table)
    qidx, buf, offs = f.fetch(ids, st, sp, flags_per_query=fl)
    assert sorted(qidx.tolist()) == list(range(len(ids)))
    for j, qi in enumerate(qidx.tolist()):
        assert buf[offs[j]:offs[j + 1]].tobytes() == _expected_fetch(oracle, raw, recs, int(ids[qi]), int(st[qi]), int(sp[qi]), int(fl[qi])), (seed, qi)
    # one process per shard: each answers what it holds; together exactly once
    pool, seen = [], np.zeros(len(ids), dtype=np.int64)
    for r in blobs:
        shard.ShardFetcher({r: blobs[r]}, bases, ends, table, exchange=lambda mine: (pool.extend(mine), [])[1]).fetch(ids, st, sp, flags_per_query=fl)
    for r in blobs:
        qidx, buf, offs = shard.ShardFetcher({r: blobs[r]}, bases, ends, table, exchange=lambda mine: list(pool)).fetch(ids, st, sp, flags_per_query=fl)
        seen[qidx] += 1
        for j, qi in enumerate(qidx.tolist()):
            assert buf[offs[j]:offs[j + 1]].tobytes() == _expected_fetch(oracle, raw, recs, int(ids[qi]), int(st[qi]), int(sp[qi]), int(fl[qi]))
    assert (seen == 1).all()


@pytest.mark.parametrize("crlf", [False, True])
def test_line_regular_across_cuts(oracle, L, crlf):
    """Records with one odd line, a long last line, ragged lines: the line-regular bit of the record that crosses a cut
    comes out of the summaries (k_stitch_tail / shard.stitch_tail) as the rule gives it on the whole stream -- for a
    cut at every byte of the first records and random cuts over all of them."""
    from test_gpu_kernels import _odd_line_fasta
    rng = np.random.default_rng(31 + crlf)
    raw = _odd_line_fasta(rng, crlf)
    first = raw[:raw.index(b">o4 ")]
    for c in range(1, len(first)):
        check(oracle, L, first, [c])
    for c in range(2, len(first) - 2, 5):
        check(oracle, L, first, [c, c + 1])
        check(oracle, L, first, [c - 1, c + 2])
    for g in (2, 3, 7, 12):
        for _ in range(6):
            check(oracle, L, raw, sorted(set(int(x) for x in rng.integers(1, len(raw), g - 1))))


def _logical_ranks(path, world, halo=None):
    """ShardedFastq for every rank of a `world`, in one process: the all-gather is the list of every range's core counts."""
    from pyfastx_amd import _lib, shard
    size, _ = _lib.stream_size(path)
    cores = []
    for r in range(world):
        lo, hi = size * r // world, size * (r + 1) // world
        b = _lib.Blob.from_file_range(path, lo, hi - lo, 0)
        cores.append(b.fastq_scan())
        b.close()
    table = np.array(cores, dtype=np.int64)
    return [shard.ShardedFastq(path, r, world, halo=halo, gather=lambda mine, t=table: t) for r in range(world)]


@pytest.mark.parametrize("world", [2, 3, 5])
def test_sharded_fastq_with_a_read_of_a_megabyte(oracle, L, tmp_path, world):
    """shard.ShardedFastq on a FILE: a halo too small for a record of the shard is no error any more (round 2: FX_ERANGE) --
    the range is opened again with a larger one.  One read of 1 MiB among ordinary ones, placed so that it begins just in
    front of a cut; the rows of all ranks, concatenated, are the oracle's; ONE .fxi is written with every rank's
    table leaves formatted from its own handle and the name index from one GPU sort of all ranks' names."""
    import sqlite3
    from pyfastx_amd import shard
    rng = np.random.default_rng(world)
    big = bytes(rng.choice(list(b"ACGT"), 1 << 20).astype(np.uint8))
    head = _rand_fastq(rng, 400)
    tail = _rand_fastq(rng, 300).replace(b"@r", b"@t")
    # the big read begins ~100 bytes in front of the first cut of the file
    raw0 = head + b"@big one\n" + big + b"\n+\n" + bytes(rng.integers(35, 71, 1 << 20).astype(np.uint8)) + b"\n" + tail
    want_cut = len(raw0) // world
    pad = want_cut - len(head) - 100
    filler = b""
    i = 0
    while len(filler) < max(pad, 0):
        filler += b"@f%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i
        i += 1
    raw = head + filler + raw0[len(head):]
    p = tmp_path / "big.fq"
    p.write_bytes(raw)
    recs, size, ln = oracle.fastq_index(raw)
    ranks = _logical_ranks(str(p), world)
    assert sum(r.reopened for r in ranks) >= 1                    # somebody's 64 KiB halo was too small
    assert sum(r.n_local for r in ranks) == len(recs) and sum(r.size for r in ranks) == size
    nxt = 0
    for r in ranks:
        assert r.n_local == 0 or r.first_id == nxt
        nxt += r.n_local
    for k in ("name_off", "name_len", "dlen", "rlen", "soff", "qoff"):
        got = np.concatenate([r.blob.fastq_table(r.n_local)[k] for r in ranks])
        np.testing.assert_array_equal(got, recs[k].astype(got.dtype), err_msg=k)
    fxi_path = str(tmp_path / "big.fq.fxi")
    n = shard.write_fastq_index_parts(ranks, fxi_path)             # every rank's leaves from its own handle, one sort of all names (round 6)
    assert n == len(recs)
    db = sqlite3.connect(fxi_path)
    assert db.execute("PRAGMA integrity_check").fetchone()[0] == "ok"
    rows = db.execute("SELECT name, dlen, rlen, soff, qoff FROM read ORDER BY ID").fetchall()
    names = [raw[int(r["name_off"]):int(r["name_off"]) + int(r["name_len"])].decode() for r in recs]
    assert rows == [(names[i], int(recs["dlen"][i]), int(recs["rlen"][i]), int(recs["soff"][i]), int(recs["qoff"][i])) for i in range(len(recs))]
    assert db.execute("SELECT counts, size FROM stat").fetchone() == (len(recs), size)
    for nm in ("big", names[3], names[-1]):
        assert db.execute("SELECT ID FROM read WHERE name=?", (nm,)).fetchone()[0] == names.index(nm) + 1
    db.close()


@pytest.mark.parametrize("world", [2, 3, 5])
def test_sharded_fastq_composition_and_routed_fetch(oracle, L, tmp_path, world):
    """Round 4: what was left of multi-GPU FASTQ.  ShardedFastq.composition -- every rank counts the reads it owns, ONE more
    all-gather of ten words, the phred rule on the merged extremes -- gives the base / meta rows of the whole file on every
    rank; ShardedFastq.fetch answers the reads of a batch that the rank owns, every read exactly once over the ranks."""
    rng = np.random.default_rng(40 + world)
    raw = _rand_fastq(rng, 700, crlf=bool(world & 1))
    p = tmp_path / "s.fq"
    p.write_bytes(raw)
    recs, size, ln = oracle.fastq_index(raw)
    oc = oracle.fastq_composition(raw)
    ranks = _logical_ranks(str(p), world)
    mine = []
    for r in ranks:                                               # what each rank would put into the all-gather
        box = []
        r.composition(gather=lambda m, box=box: (box.append(m.copy()), np.tile(m, (world, 1)))[1])
        mine.append(box[0])
    table = np.stack(mine)
    for r in ranks:
        base, meta = r.composition(gather=lambda m, t=table: t)
        assert base.tolist() == [oc["a"], oc["c"], oc["g"], oc["t"], oc["n"]]
        assert meta.tolist() == [oc["maxlen"], oc["minlen"], oc["minqs"], oc["maxqs"], oc["phred"]]
    first = np.concatenate([[0], np.cumsum([r.n_local for r in ranks])])
    ids = rng.integers(0, len(recs), 2000)
    seen = np.zeros(ids.size, dtype=np.int64)
    for r in ranks:
        pos, seq, qual, qi, offs = r.fetch(ids, first, phred=int(oc["phred"]))
        seen[pos] += 1
        for j, k in enumerate(pos.tolist()):
            i = int(ids[k])
            s0, q0, l = int(recs["soff"][i]), int(recs["qoff"][i]), int(recs["rlen"][i])
            assert seq[offs[j]:offs[j + 1]].tobytes() == raw[s0:s0 + l] and qual[offs[j]:offs[j + 1]].tobytes() == raw[q0:q0 + l]
            assert (qi[offs[j]:offs[j + 1]] == oracle.quali(raw, q0, l, int(oc["phred"]))).all()
    assert (seen == 1).all()

