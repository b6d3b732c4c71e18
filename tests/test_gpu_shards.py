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
        rows.append(b.fasta_table(n))
    return {c: np.concatenate([t[c] for t in rows]) for c in COLS}, S


def check(oracle, L, raw, cuts, full_name=False):
    recs, tot = oracle.fasta_index(raw, full_name=full_name)
    got, S = sharded_rows(L, raw, cuts, full_name)
    assert len(got["boff"]) == len(recs), (cuts, len(got["boff"]), len(recs))
    for c in COLS:
        np.testing.assert_array_equal(got[c], recs[c].astype(got[c].dtype), err_msg="%s cuts=%s" % (c, cuts))


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
