"""-m gpu: BGZF inputs are inflated by k_bgzf_inflate (one work-item per member) and
then indexed / fetched exactly like the plain file."""
import gzip
import os
import sqlite3
import struct

import numpy as np
import pytest

from conftest import DATA, fixture_bytes, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from pyfastx_amd import _lib
    assert _lib.lib().fx_device_count() >= 1
    return _lib


def _write(tmp_path, name, payload):
    p = tmp_path / name
    p.write_bytes(payload)
    return str(p)


@pytest.mark.parametrize("level,block", [(6, 65280), (1, 65280), (9, 4096), (0, 65280), (6, 700)])
def test_inflate_matches_zlib(L, tmp_path, level, block):
    from pyfastx_amd import synth
    rng = np.random.default_rng(level * 7 + block)
    parts = [fixture_bytes("test.fa"), bytes(rng.integers(0, 256, 200_000, dtype=np.uint8)),     # incompressible
             b"A" * 300_000, fixture_bytes("test.fq")[:150_000], b""]                            # long runs (dist < len)
    raw = b"".join(parts)
    bg = synth.bgzf_compress(raw, block=block, level=level)
    assert gzip.decompress(bg) == raw                     # the framing itself is valid gzip
    p = _write(tmp_path, "x.fa.gz", bg)
    b = L.Blob.from_file(p)
    assert b.is_gzip and b.size == len(raw)
    got = b.read_bytes(0, len(raw))
    assert got == raw
    c, u, cs = b.gz_points(spacing=100_000)
    import zlib
    assert cs == len(bg) and c[0] == 18 and u[0] == 0 and len(c) >= 2
    for co, uo in zip(c, u):                              # each point is the first deflate byte of a member: what a raw inflate
        assert bg[co - 18:co - 14] == b"\x1f\x8b\x08\x04"  # (zran_seek: inflateInit2(-15) at cmp_offset) starts from
        assert zlib.decompressobj(-15).decompress(bg[co:co + 70_000], 64) == raw[uo:uo + 64]


def _genome_like(rng, n):
    """Soft-masked four-letter text in 60-column lines with runs of N: what a FASTA member looks like to a compressor."""
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.choice(4, n, p=[.295, .205, .205, .295])]
    low = np.repeat(rng.random(n // 5000 + 1) < 0.4, 5000)[:n]
    letters = np.where(low, letters + 32, letters).astype(np.uint8)
    for _ in range(max(1, n // 400_000)):
        a = int(rng.integers(0, n - 1))
        letters[a:a + int(rng.integers(1000, 120_000))] = ord("N")
    rows = letters[:n - n % 60].reshape(-1, 60)
    return b"\n".join(r.tobytes() for r in rows) + b"\n"


@pytest.mark.parametrize("level", [1, 6, 9])
def test_members_are_decoded_by_the_wave_kernel(L, tmp_path, level):
    """k_bgzf_decode_par (one wave per member, the lanes at 64 bit positions of a block, hand-over by Huffman
    self-synchronisation) decodes ordinary members ITSELF -- genome text at three compression levels (level 1: two or three
    deflate blocks per member; N runs: members of a few hundred symbols that use a handful of lanes), FASTQ, protein --
    and hands nothing to the serial kernel; the bytes equal the input and every member passes its CRC-32."""
    from pyfastx_amd import synth
    rng = np.random.default_rng(level)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    raw = b"".join([_genome_like(rng, 3_000_000), fixture_bytes("test.fq") * 3, b">p\n" + aa[rng.integers(0, 20, 700_000)].tobytes() + b"\n"])
    bg = synth.bgzf_compress(raw, level=level)
    p = _write(tmp_path, "w.gz", bg)
    b = L.Blob.from_file(p)
    assert b.size == len(raw) and b.read_bytes(0, len(raw)) == raw
    members, handed, reason = b.bgzf_counts()
    assert members == (len(raw) + 65279) // 65280 + 1 and handed == 0, (members, handed, reason)


def test_damaged_and_odd_members_go_to_the_serial_kernel(L, tmp_path):
    """What the wave kernel does not take on -- here: a payload that ends in the middle of a block -- is handed over, and the
    serial kernel names the error."""
    from pyfastx_amd import synth
    raw = _genome_like(np.random.default_rng(5), 400_000)
    bg = bytearray(synth.bgzf_compress(raw))
    good = bytes(bg)
    # cut 40 bytes out of the second member's deflate data and fix its BSIZE: the stream ends early
    import struct as st
    m0 = st.unpack("<H", good[16:18])[0] + 1
    m1 = st.unpack("<H", good[m0 + 16:m0 + 18])[0] + 1
    bad = good[:m0 + 16] + st.pack("<H", m1 - 40 - 1) + good[m0 + 18:m0 + m1 - 48] + good[m0 + m1 - 8:]
    p = _write(tmp_path, "bad.gz", bad)
    with pytest.raises(L.FxError):
        L.Blob.from_file(p)
    b = L.Blob.from_file(_write(tmp_path, "good.gz", good))
    assert b.read_bytes(0, len(raw)) == raw and b.bgzf_counts()[1] == 0


def test_fixed_huffman_and_stored_members(L, tmp_path):
    import zlib
    raws = [b"hello hello hello hello\n", bytes(range(256)) * 3, b""]
    out = []
    for i, chunk in enumerate(raws):
        co = zlib.compressobj(6, zlib.DEFLATED, -15, 9, zlib.Z_FIXED if i == 0 else zlib.Z_DEFAULT_STRATEGY)
        cd = co.compress(chunk) + co.flush()
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cd) + 25) + cd +
                   struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    p = _write(tmp_path, "f.gz", b"".join(out))
    b = L.Blob.from_file(p)
    assert b.read_bytes(0, b.size) == b"".join(raws)


def test_matches_and_literals_at_member_ends(L, tmp_path):
    """The decoder stores a symbol as four bytes at its place (a literal, or the 3-byte token of a match in the first
    bytes the match will fill) except within four bytes of the member's end, and the copy pass walks a map of one bit per
    output byte: members of 1 .. 9 bytes, members that END in a shortest match, in a run (distance 1), in literals after
    a match, full-size members of repeats, every combination next to a neighbour whose first bytes must survive; and
    the scratch pool of the opens is emptied and refilled on the way (fx_release_scratch)."""
    import zlib
    rng = np.random.default_rng(5)
    dna = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 70_000)])
    chunks = [b"A", b"AC", b"ACG", b"ACGT", b"ACGTA", b"AAAAAAAAA", b"ABCABC", b"XYZ" + b"ABCDEFG" + b"XYZ", b"QABCABCABC", b"ABCDABCDABCDAB",
              dna[:65280], (dna[:997] * 70)[:65280], dna[:300] + b"N" * 5000 + dna[:300], (b"ACGTTGCA" * 9000)[:65279] + b"Z",
              dna[:1000] + dna[:1000][-3:], dna[:1000] + dna[500:503], b"".join(dna[i:i + 60] + b"\n" for i in range(0, 60_000, 60))]
    for lvl in (9, 6, 1):
        out, raw = [], []
        order = list(range(len(chunks)))
        rng.shuffle(order)
        for i in order + order[::-1]:
            chunk = chunks[i]
            co = zlib.compressobj(lvl, zlib.DEFLATED, -15)
            cd = co.compress(chunk) + co.flush()
            out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cd) + 25) + cd +
                       struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
            raw.append(chunk)
        raw = b"".join(raw)
        p = _write(tmp_path, "ends%d.gz" % lvl, b"".join(out))
        for k in range(3):
            b = L.Blob.from_file(p)
            assert b.size == len(raw)
            got = b.read_bytes(0, b.size)
            assert got == raw, (lvl, k, next(i for i in range(len(raw)) if got[i] != raw[i]))
            b.close()
            if k == 1:
                L.check(L.lib().fx_release_scratch())


def test_corrupt_member_is_reported(L, tmp_path):
    from pyfastx_amd import synth
    bg = bytearray(synth.bgzf_compress(fixture_bytes("test.fa"), block=20000))
    bg[40] ^= 0xFF; bg[41] ^= 0x55; bg[60] ^= 0xFF
    with pytest.raises(L.FxError) as e:
        L.Blob.from_file(_write(tmp_path, "bad.fa.gz", bytes(bg)))
    assert e.value.code == L.FX_EIO and "BGZF member" in str(e.value)


def test_crc_of_every_member_is_checked(L, tmp_path):
    """a member that inflates to the right length with wrong bytes is an error (zlib's gzread checks the CRC-32 in
    the reference): stored members (level 0) with one payload byte flipped inflate cleanly -- only the CRC kernel notices."""
    from pyfastx_amd import synth
    raw = fixture_bytes("test.fa")
    bg = bytearray(synth.bgzf_compress(raw, block=20000, level=0))
    good = L.Blob.from_file(_write(tmp_path, "ok.fa.gz", bytes(bg)))
    assert good.read_bytes(0, len(raw)) == raw
    pos = bg.index(raw[30000:30040]) + 7                   # inside the stored payload of the second member
    bg[pos] ^= 0x20
    with pytest.raises(L.FxError) as e:
        L.Blob.from_file(_write(tmp_path, "flip.fa.gz", bytes(bg)))
    assert e.value.code == L.FX_EIO and "CRC-32" in str(e.value) and "member 1 " in str(e.value)


def test_bgzf_fasta_through_the_api(tmp_path):
    import pyfastx_amd as fx
    from pyfastx_amd import synth
    g = load_golden("fasta_fixture")["test.fa"]
    p = _write(tmp_path, "test.fa.gz", synth.bgzf_compress(fixture_bytes("test.fa"), block=8000))
    fa = fx.Fasta(p, full_index=True)
    assert fa.is_gzip and len(fa) == g["count"] and fa.size == g["size"]
    db = sqlite3.connect(p + ".fxi")
    assert [list(r) for r in db.execute("SELECT * FROM seq")] == g["seq"]
    assert [list(r[1:]) for r in db.execute("SELECT * FROM comp")] == g["comp"]
    blobs = [r[0] for r in db.execute("SELECT content FROM gzindex ORDER BY ID")]
    assert blobs[0] == b"GZIDX" and struct.unpack("<Q", blobs[3])[0] == os.path.getsize(p)
    npoints = struct.unpack("<I", blobs[7])[0]
    assert npoints >= 1 and len(blobs) == 8 + 4 * npoints
    for f in g["fetches"][:100]:
        sub = fa[f["id"] - 1][f["start"]:f["stop"]]
        assert sub.seq == f["seq"] and sub.antisense == f["antisense"]


def test_plain_gzip_still_goes_through_host_zlib(L):
    b = L.Blob.from_file(os.path.join(DATA, "test.fa.gz"))     # single-member gzip, not BGZF
    assert b.read_bytes(0, b.size) == fixture_bytes("test.fa.gz")
    assert b.gz_points()[0].size == 0
    p = b.gz_checkpoints()                                      # a small file: only the point at the start of the deflate data
    assert p["cmp"].size == 1 and p["uncmp"][0] == 0 and p["has"][0] == 0 and p["windows"].size == 0


def _big_fasta(rng, nrec=40, width=70):
    parts = []
    for i in range(nrec):
        parts.append(b">rec%d some description %d\n" % (i, i * 7))
        s = bytes(rng.choice(list(b"ACGTNacgtn"), int(rng.integers(50_000, 400_000))).astype(np.uint8))
        parts.append(b"\n".join(s[p:p + width] for p in range(0, len(s), width)) + b"\n")
    return b"".join(parts)


@pytest.mark.parametrize("shape", ["one_member", "pigz_style", "two_members", "stored_blocks"])
def test_single_stream_gzip_checkpoints(L, tmp_path, shape):
    """SURVEY a13 / VERDICT r1 #8: a single gzip stream is inflated serially ONCE -- restart points (compressed offset, bits,
    uncompressed offset, 32 KiB window; the rows of util.c:461-529) are captured at deflate block boundaries >= 1 MiB apart
    -- and every later open inflates the segments between the points in parallel (fx_open_file_indexed): same bytes.
    Every point is checked the way zran uses it: a raw inflate primed with its bits and window continues the stream."""
    import zlib
    from pyfastx_amd import synth, fxi
    rng = np.random.default_rng(len(shape))
    raw = _big_fasta(rng)
    if shape == "one_member":
        gz = gzip.compress(raw, 6)
    elif shape == "pigz_style":
        gz = synth.gzip_single_stream(raw, piece=1 << 20)
    elif shape == "two_members":
        gz = gzip.compress(raw[:len(raw) // 3], 6) + gzip.compress(raw[len(raw) // 3:], 1)
    else:
        gz = gzip.compress(raw[:3_000_000], 0) + gzip.compress(raw[3_000_000:], 6)      # level 0: stored blocks (64 KiB each)
    assert len(raw) > 6 * 1048576
    p = _write(tmp_path, "big.fa.gz", gz)
    b = L.Blob.from_file(p)
    assert b.size == len(raw) and b.read_bytes(0, len(raw)) == raw
    pts = b.gz_checkpoints()
    n = pts["cmp"].size
    assert n >= 5 and pts["uncmp"][0] == 0 and pts["has"][0] == 0 and int(pts["has"].sum()) == n - 1
    assert pts["windows"].size == (n - 1) * 32768 and (np.diff(pts["uncmp"]) >= 1048576).all()
    k = 0
    for i in range(n):                                         # zran's use of a point (zran_seek -> inflate from there)
        d = zlib.decompressobj(-15)
        c, u, bits = int(pts["cmp"][i]), int(pts["uncmp"][i]), int(pts["bits"][i])
        data = gz[c:c + 200_000]
        if pts["has"][i]:
            win = pts["windows"][k * 32768:(k + 1) * 32768].tobytes()
            k += 1
            assert win == raw[u - 32768:u] if u >= 32768 else win[-u:] == raw[:u]
            d = zlib.decompressobj(-15, zdict=win)
        if bits:                                               # python's zlib has no inflatePrime (and a shifted copy of the stream breaks
            continue                                           # the byte alignment of stored blocks): the parallel re-open below uses
        got = d.decompress(data, 50_000)                       # every point with the real inflatePrime
        assert got == raw[u:u + len(got)] and len(got) > 0, (shape, i)
    # the index file carries them; the next open inflates in parallel and sees the same stream
    import pyfastx_amd as fx
    fa = fx.Fasta(p)
    db = sqlite3.connect(p + ".fxi")
    rows = [bytes(r[0]) for r in db.execute("SELECT content FROM gzindex ORDER BY ID")]
    assert rows[0] == b"GZIDX" and struct.unpack("<I", rows[7])[0] == n and len(rows) == 8 + 4 * n + (n - 1)
    assert struct.unpack("<Q", rows[3])[0] == len(gz) and struct.unpack("<Q", rows[4])[0] == len(raw)
    got = fxi.read_gzindex(db)
    assert (got["cmp"] == pts["cmp"]).all() and (got["bits"] == pts["bits"]).all() and got["windows"].tobytes() == pts["windows"].tobytes()
    whole = [s.seq for s in fa]
    del fa
    fb = fx.Fasta(p)                                           # index exists: parallel inflate from its points
    assert fb._st.blob.size == len(raw) and fb._st.blob.read_bytes(0, len(raw)) == raw
    again = fb._st.blob.gz_checkpoints()
    assert again["cmp"].size == n and again["windows"].size == 0          # the parallel path ran (a serial inflate captures windows again)
    assert [s.seq for s in fb] == whole
    # points that do not describe the file: the serial path, silently, same bytes
    bad = dict(got, cmp=got["cmp"] + np.where(np.arange(n) == 2, 3, 0))
    bb = L.Blob.from_file(p, gzindex=bad)
    assert bb.read_bytes(0, len(raw)) == raw and bb.gz_checkpoints()["windows"].size == (n - 1) * 32768


def test_first_open_of_a_large_single_stream_runs_on_all_cores(L, tmp_path):
    """A single gzip stream of >= 32 MiB is inflated by fx_pgzip.hpp on its FIRST open (mode 3; the serial zlib pass of round 2
    is mode 2): the bytes are the input's, the restart points it captured on the way are valid zran points -- the second open
    inflates from them in parallel (mode 4) and sees the same stream, and the compiled reference reads its slices THROUGH them."""
    import glob
    import sys
    import zlib
    from conftest import ROOT
    rng = np.random.default_rng(77)
    parts = []
    for i in range(24):
        parts.append(b">rec%d some description\n" % i)
        s = bytes(rng.choice(list(b"ACGTNacgtn"), int(rng.integers(3_000_000, 6_000_000))).astype(np.uint8))
        parts.append(b"\n".join(s[p:p + 70] for p in range(0, len(s), 70)) + b"\n")
    raw = b"".join(parts)
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    gz = co.compress(raw) + co.flush()
    assert len(gz) >= 32 << 20
    p = _write(tmp_path, "large.fa.gz", gz)
    b = L.Blob.from_file(p)
    assert b.gz_open_mode == 3 and b.size == len(raw)
    for a in (0, len(raw) // 3, len(raw) - 5_000_000):
        assert b.read_bytes(a, 5_000_000) == raw[a:a + 5_000_000]
    pts = b.gz_checkpoints()
    n = pts["cmp"].size
    assert n >= len(raw) // (2 << 20) and pts["uncmp"][0] == 0 and pts["has"][0] == 0 and int(pts["has"].sum()) == n - 1
    assert (np.diff(pts["uncmp"]) >= 1048576).all() and pts["windows"].size == (n - 1) * 32768
    k = 0
    for i in range(n):                                         # every point the way zran uses it (those on a byte boundary: python has no inflatePrime)
        c, u, bits = int(pts["cmp"][i]), int(pts["uncmp"][i]), int(pts["bits"][i])
        d = zlib.decompressobj(-15)
        if pts["has"][i]:
            win = pts["windows"][k * 32768:(k + 1) * 32768].tobytes()
            k += 1
            assert win == raw[u - 32768:u]
            d = zlib.decompressobj(-15, zdict=win)
        if bits == 0:
            got = d.decompress(gz[c:c + 200_000], 50_000)
            assert got == raw[u:u + len(got)] and len(got) > 0
    del b
    import pyfastx_amd as fx
    fa = fx.Fasta(p)                                           # first open again (no index yet): parallel, writes the points
    assert fa._st.blob.gz_open_mode == 3
    del fa
    fb = fx.Fasta(p)
    some = [fb[5][1000:1400].seq, fb[23][-300:].antisense, fb[0][:80].seq]
    assert fb._st.blob.gz_open_mode == 4 and fb._st.blob.read_bytes(len(raw) - 1000, 1000) == raw[-1000:]
    if not glob.glob(os.path.join(ROOT, "oracle", "_ref", "pyfastx*.so")):
        pytest.fail("oracle/_ref is missing")
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    import pyfastx
    so, stats, hits = _shim(pyfastx)
    so.fxshim_reset()
    rf = pyfastx.Fasta(p)
    assert [rf[5][1000:1400].seq, rf[23][-300:].antisense, rf[0][:80].seq] == some
    for _ in range(300):
        kk = int(rng.integers(0, 24))
        a = int(rng.integers(0, len(rf[kk]) - 500))
        assert rf[kk][a:a + 300].seq == fb[kk][a:a + 300].seq
    st = stats()
    assert st["errors"] == 0 and st["built_points"] == 0 and st["from_point"] >= 200


def _shim(pyfastx):
    """The counters of the compiled reference's zran work-alike (oracle/refshim/zran.c), through ctypes."""
    import ctypes
    so = ctypes.CDLL(pyfastx.__file__)
    so.fxshim_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    so.fxshim_point_hits.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32]

    def stats():
        a = (ctypes.c_uint64 * 6)()
        so.fxshim_stats(a)
        return dict(zip(("seeks", "from_point", "from_start", "continued", "built_points", "errors"), (int(x) for x in a)))

    def hits(n):
        a = (ctypes.c_uint32 * n)()
        so.fxshim_point_hits(a, n)
        return [int(x) for x in a]
    return so, stats, hits


@pytest.mark.parametrize("shape", ["zlib6", "pigz_style", "two_members", "stored_then_deflate", "bgzf"])
def test_reference_reads_through_our_restart_points(tmp_path, shape):
    """SURVEY a13 pinned THROUGH the reference: the compiled reference opens an index file the PRODUCT wrote, imports its
    gzindex rows (pyfastx_gzip_index_import, util.c:542-726: id, version, sizes, window >= 32768, spacing >= window, the
    points and their windows) and answers 1 000 random slices by zran_seek / zran_read (index.c:685-686) -- served by the
    zran work-alike of oracle/refshim FROM THE IMPORTED POINTS: last point at or before the offset, raw inflate primed
    with the point's bits and window.  The shim's counters prove it: no point was built in this process, every imported
    point started at least one seek, no inflate error -- and every answer equals the product's and the plain bytes."""
    import glob
    import sys
    from conftest import ROOT
    from pyfastx_amd import synth
    if not glob.glob(os.path.join(ROOT, "oracle", "_ref", "pyfastx*.so")):
        pytest.fail("oracle/_ref is missing")
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    import pyfastx
    import pyfastx_amd as fx
    rng = np.random.default_rng(3)
    N = 40
    raw = _big_fasta(rng, nrec=N)
    if shape == "zlib6":
        gz = gzip.compress(raw, 6)
    elif shape == "pigz_style":
        gz = synth.gzip_single_stream(raw, piece=1 << 20)
    elif shape == "two_members":
        gz = gzip.compress(raw[:len(raw) // 3], 6) + gzip.compress(raw[len(raw) // 3:], 1)
    elif shape == "stored_then_deflate":
        gz = gzip.compress(raw[:3_000_000], 0) + gzip.compress(raw[3_000_000:], 6)
    else:
        gz = synth.bgzf_compress(raw)
    p = _write(tmp_path, "r.fa.gz", gz)
    fa = fx.Fasta(p)                                           # the product writes the index file, gzindex included
    db = sqlite3.connect(p + ".fxi")
    npts = struct.unpack("<I", bytes(db.execute("SELECT content FROM gzindex WHERE ID=8").fetchone()[0]))[0]
    pts = fxi_points(db)
    db.close()
    assert npts >= 5 and len(pts) == npts
    if shape != "bgzf":
        assert any(b for _, _, b, _ in pts) or shape == "stored_then_deflate"      # points in the middle of a byte are in play
    so, stats, hits = _shim(pyfastx)
    so.fxshim_reset()
    rf = pyfastx.Fasta(p)                                      # loads OUR index file: import, no build
    assert len(rf) == len(fa) == N and stats()["built_points"] == 0
    # slices: two that start just behind every point's offset (so that each point has to serve a seek), the rest random
    recs = [(s.name, len(s)) for s in fa]
    tab = sqlite3.connect(p + ".fxi").execute("SELECT boff, blen, slen FROM seq ORDER BY ID").fetchall()
    q = []
    for (_, u, _, _) in pts:
        for k, (boff, blen, slen) in enumerate(tab):
            if boff <= u + 200 < boff + blen - 400:
                a = ((u + 200 - boff) * 70) // 71                # a base whose byte lies a little behind the point
                q.append((k, a, min(a + 120, slen)))
                break
    while len(q) < 1000:
        k = int(rng.integers(0, N))
        a = int(rng.integers(0, recs[k][1] - 1))
        q.append((k, a, min(recs[k][1], a + int(rng.integers(1, 400)))))
    order = rng.permutation(len(q))                            # random order: every seek starts over
    for j in order.tolist():
        k, a, b = q[j]
        want = fa[k][a:b]
        got = rf[recs[k][0]][a:b]
        assert got.seq == want.seq and got.antisense == want.antisense, (shape, k, a, b)
    assert rf[7].seq == fa[7].seq and rf[N - 1].raw == fa[N - 1].raw
    st = stats()
    h = hits(npts)
    assert st["errors"] == 0 and st["built_points"] == 0 and st["from_start"] == 0
    assert st["from_point"] >= 1000 * 0.5 and all(x >= 1 for x in h), (st, h)


def fxi_points(db):
    """(cmp, uncmp, bits, has_data) of every point of a gzindex table, straight from its rows (util.c:461-529)."""
    rows = [bytes(r[0]) for r in db.execute("SELECT content FROM gzindex ORDER BY ID")]
    n = struct.unpack("<I", rows[7])[0]
    out = []
    for i in range(n):
        c, u, b, f = rows[8 + 4 * i:12 + 4 * i]
        out.append((struct.unpack("<Q", c)[0], struct.unpack("<Q", u)[0], b[0], f[0]))
    return out


@pytest.mark.parametrize("group", [8192, 65536, 1 << 20])
def test_open_in_groups_behind_the_staging(L, tmp_path, monkeypatch, group):
    """Round 4: a large BGZF file is inflated group by group while its later bytes are still on their way to the device
    (bgzf_open_pipelined; FX_BGZF_GROUP makes a few MB "large").  Members that straddle two groups, groups that hold no
    complete member, the 28-byte EOF member alone in the last group: the bytes, the restart points and the counts are the
    one-shot open's."""
    from pyfastx_amd import synth
    rng = np.random.default_rng(group)
    raw = b">chr1 six million\n" + _genome_like(rng, 6_000_000) + b">fixture\n" + fixture_bytes("test.fa")[:200_000]
    bg = synth.bgzf_compress(raw, block=65280 if group > 8192 else 30000)
    p = _write(tmp_path, "g.fa.gz", bg)
    monkeypatch.setenv("FX_BGZF_GROUP", "0")
    one = L.Blob.from_file(p)
    want_pts = one.gz_points(spacing=100_000)
    want_counts = one.bgzf_counts()
    monkeypatch.setenv("FX_BGZF_GROUP", str(group))
    monkeypatch.setenv("FX_TRACE_BGZF", "1")
    b = L.Blob.from_file(p)
    assert b.is_gzip and b.size == len(raw) == one.size
    assert b.read_bytes(0, len(raw)) == raw
    got_pts = b.gz_points(spacing=100_000)
    for x, y in zip(got_pts, want_pts):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    assert b.bgzf_counts() == want_counts
    s = b.fasta_build()                                    # the blob is a stream like any other (its allocation is a little larger than it)
    s1 = one.fasta_build()
    assert (s.n_seq, s.seq_len) == (s1.n_seq, s1.seq_len) and s.n_seq == 2 + fixture_bytes("test.fa")[:200_000].count(b">")


def test_groups_give_way_when_the_file_inflates_further_than_its_head_promised(L, tmp_path, monkeypatch):
    """The blob of a pipelined open is allocated from the ratio of the first group (+ 10 %): a file whose tail is a run of N
    (ratio ~1000) outgrows it and is opened all at once instead -- same bytes; a member damaged in a LATER group is reported
    with its number in the file."""
    from pyfastx_amd import synth
    rng = np.random.default_rng(5)
    raw = _genome_like(rng, 1_500_000) + b"N" * 30_000_000
    bg = synth.bgzf_compress(raw)
    monkeypatch.setenv("FX_BGZF_GROUP", "65536")
    b = L.Blob.from_file(_write(tmp_path, "n.fa.gz", bg))
    assert b.size == len(raw) and b.read_bytes(1_400_000, 200_000) == raw[1_400_000:1_600_000] and b.read_bytes(len(raw) - 70_000, 70_000) == raw[-70_000:]
    raw2 = _genome_like(rng, 3_000_000)
    bg2 = bytearray(synth.bgzf_compress(raw2, level=0))   # stored members: a flipped payload byte only shows in the CRC
    member = 2_000_000 // 65280                          # a stored member: 18 bytes of header, 5 of block header, 65280 of payload, 8 of trailer
    pos = member * (65280 + 31) + 23 + (2_000_000 - member * 65280)
    assert bg2[pos] == raw2[2_000_000]
    bg2[pos] ^= 0x20
    with pytest.raises(L.FxError) as e:
        L.Blob.from_file(_write(tmp_path, "flip2.fa.gz", bytes(bg2)))
    assert e.value.code == L.FX_EIO and "CRC-32" in str(e.value) and ("member %d " % member) in str(e.value)
