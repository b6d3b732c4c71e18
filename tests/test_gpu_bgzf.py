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
    assert cs == len(bg) and c[0] == 0 and u[0] == 0 and len(c) >= 2
    for co, uo in zip(c, u):                              # each point really is a member start
        assert bg[co:co + 4] == b"\x1f\x8b\x08\x04"
        assert gzip.decompress(bg[co:])[:64] == raw[uo:uo + 64]


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


def test_corrupt_member_is_reported(L, tmp_path):
    from pyfastx_amd import synth
    bg = bytearray(synth.bgzf_compress(fixture_bytes("test.fa"), block=20000))
    bg[40] ^= 0xFF; bg[41] ^= 0x55; bg[60] ^= 0xFF
    with pytest.raises(L.FxError) as e:
        L.Blob.from_file(_write(tmp_path, "bad.fa.gz", bytes(bg)))
    assert e.value.code == L.FX_EIO and "BGZF member" in str(e.value)


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
