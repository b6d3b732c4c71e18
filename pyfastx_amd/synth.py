"""Synthetic workloads of BASELINE.json's shapes, generated directly in HBM with
torch (plumbing only) together with their ANALYTIC ground truth, so that
full-size runs can be checked without a CPU pass over 3 GB.

C2: hg38-shaped FASTA -- 24 chromosomes proportional to hg38 + 176 scaffolds
(log-uniform 1 kbp..500 kbp), 60-column LF lines, ~50 % soft-masked blocks,
telomere / centromere N runs, `>name  AC:.. LN:.. rl:..` headers.
"""
import numpy as np

HG38_MBP = [248, 242, 198, 190, 181, 171, 159, 145, 138, 134, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47,
            51, 156, 57]
CHR_NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]


def fasta_plan(total_bp=3_000_000_000, n_scaffolds=176, width=60, seed=20260612, tag=""):
    """Contig names / lengths / headers and the analytic index rows."""
    rng = np.random.default_rng(seed)
    scaff = np.exp(rng.uniform(np.log(1_000), np.log(500_000), n_scaffolds)).astype(np.int64)
    chrom_total = total_bp - int(scaff.sum())
    w = np.array(HG38_MBP, dtype=np.float64)
    chrom = np.floor(w / w.sum() * chrom_total).astype(np.int64)
    chrom[0] += chrom_total - int(chrom.sum())
    if total_bp < 50_000_000:          # tiny test plans: keep at least one line per chromosome
        chrom = np.maximum(chrom, 1)
    lens = np.concatenate([chrom, scaff])
    names = [tag + n for n in CHR_NAMES] + ["%schrUn_KI%06dv1" % (tag, 270000 + i) for i in range(n_scaffolds)]
    headers = []
    for i, (nm, ln) in enumerate(zip(names, lens)):
        sep = "\t" if i % 7 == 3 else "  "            # exercise the space-or-tab name split (index.c:289-293)
        kind = "Chromosome" if i < 24 else "unplaced-scaffold"
        headers.append((">%s%sAC:CM%06d.2  gi:%d  LN:%d  rl:%s  M5:%032x  AS:synthetic" %
                        (nm, sep, 663 + i, 568336000 + i, ln, kind, int(rng.integers(0, 2**62)))).encode())
    # analytic rows (index.c:230-372 semantics for a well-formed LF file)
    n = len(lens)
    hoff = np.zeros(n, np.int64); boff = np.zeros(n, np.int64); blen = np.zeros(n, np.int64)
    llen = np.zeros(n, np.int64); dlen = np.zeros(n, np.int64); name_len = np.zeros(n, np.int64)
    pos = 0
    for i in range(n):
        L = int(lens[i])
        nlines = (L + width - 1) // width
        hoff[i] = pos
        boff[i] = pos + len(headers[i]) + 1
        blen[i] = L + nlines
        llen[i] = (min(L, width) + 1) if L > 0 else 0
        dlen[i] = len(headers[i]) - 1
        name_len[i] = len(names[i])
        pos = int(boff[i] + blen[i])
    return {"names": names, "headers": headers, "slen": lens.astype(np.int64), "hoff": hoff, "boff": boff,
            "blen": blen, "llen": llen, "dlen": dlen, "name_len": name_len, "width": width,
            "n_bytes": pos, "n_chrom": 24, "seed": seed}


def fasta_generate(plan, device, keep_flat=True):
    """-> (blob uint8[n_bytes(+pad)] on `device`, flat bases uint8[sum slen] or None, flat_start int64[n])."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(plan["seed"]))
    rng = np.random.default_rng(plan["seed"] + 1)
    nb = int(plan["n_bytes"])
    pad = (-nb) % 65536 + 65536
    blob = torch.zeros(nb + pad, dtype=torch.uint8, device=device)
    total = int(plan["slen"].sum())
    flat = torch.empty(total, dtype=torch.uint8, device=device) if keep_flat else None
    flat_start = np.zeros(len(plan["slen"]), np.int64)
    # byte -> base LUT with P(A,C,G,T) ~= .295/.205/.205/.295
    lut = torch.empty(256, dtype=torch.uint8, device=device)
    lut[:76] = ord("A"); lut[76:128] = ord("C"); lut[128:180] = ord("G"); lut[180:] = ord("T")
    w = plan["width"]
    fpos = 0
    for i, L in enumerate(plan["slen"].tolist()):
        hdr = plan["headers"][i]
        ho, bo = int(plan["hoff"][i]), int(plan["boff"][i])
        blob[ho:ho + len(hdr)] = torch.frombuffer(bytearray(hdr), dtype=torch.uint8).to(device)
        blob[ho + len(hdr)] = 10
        flat_start[i] = fpos
        if L == 0:
            continue
        seq = lut[torch.randint(0, 256, (L,), dtype=torch.uint8, device=device, generator=g).long()] \
            if L < (1 << 22) else _lut_big(lut, L, device, g)
        # soft-masked (lower-case) blocks of 1..50 kb, about half of them masked
        nblk = max(1, L // 25_000 + 1)
        blens = torch.from_numpy(rng.integers(1_000, 50_001, nblk * 2)).to(device)
        bits = torch.from_numpy((rng.random(nblk * 2) < 0.5).astype(np.uint8)).to(device)
        mask = torch.repeat_interleave(bits, blens)[:L]
        if mask.numel() < L:
            mask = torch.cat([mask, torch.zeros(L - mask.numel(), dtype=torch.uint8, device=device)])
        seq = seq | (mask << 5)                        # 0x20 = lower case
        del mask, blens, bits
        if i < plan["n_chrom"] and L > 4_000_000:      # telomeres + one centromere block of N
            seq[:10_000] = ord("N"); seq[L - 10_000:] = ord("N")
            c0 = int(rng.integers(L // 3, L // 2)); cl = int(rng.integers(1_000_000, 3_000_001))
            seq[c0:c0 + cl] = ord("N")
        if keep_flat:
            flat[fpos:fpos + L] = seq
        fpos += L
        full = L // w
        if full:
            view = blob[bo:bo + full * (w + 1)].view(full, w + 1)
            view[:, :w] = seq[:full * w].view(full, w)
            view[:, w] = 10
        rem = L - full * w
        if rem:
            p = bo + full * (w + 1)
            blob[p:p + rem] = seq[full * w:]
            blob[p + rem] = 10
        del seq
    return blob, flat, flat_start


def _lut_big(lut, L, device, g):
    import torch
    out = torch.empty(L, dtype=torch.uint8, device=device)
    step = 1 << 26
    for a in range(0, L, step):
        b = min(L, a + step)
        out[a:b] = lut[torch.randint(0, 256, (b - a,), dtype=torch.uint8, device=device, generator=g).long()]
    return out


def fasta_queries(plan, n=1_000_000, qlen=100, seed=12345, skip_first=False):
    """contig ~ length, start uniform in [0, slen-qlen], 50 % '-' strand."""
    rng = np.random.default_rng(seed)
    slen = plan["slen"]
    ok = np.nonzero(slen >= qlen)[0]
    if skip_first:
        ok = ok[ok > 0]
    p = slen[ok].astype(np.float64)
    ids = ok[rng.choice(ok.size, n, p=p / p.sum())]
    start = (rng.random(n) * (slen[ids] - qlen + 1)).astype(np.int64)
    strand = (rng.random(n) < 0.5).astype(np.uint8)     # 1 = '-'
    return ids.astype(np.int64), start, start + qlen, strand


def expected_fetch(flat, flat_start, ids, start, qlen, strand, device):
    """Reference answer on device from the un-wrapped bases (torch indexing, not our kernels)."""
    import torch
    base = torch.from_numpy(flat_start[ids] + start).to(device)
    idx = base[:, None] + torch.arange(qlen, device=device)[None, :]
    out = flat[idx]                                       # [n, qlen]
    comp = torch.arange(256, dtype=torch.uint8, device=device)
    for a, b in ("AT", "CG", "MK", "RY", "VB", "HD"):
        for x, y in ((a, b), (b, a)):
            comp[ord(x)] = ord(y); comp[ord(x) + 32] = ord(y) + 32
    comp[ord("U")] = ord("A"); comp[ord("u")] = ord("a")
    neg = torch.from_numpy(strand.astype(bool)).to(device)
    rc = comp[out.long()].flip(1)
    return torch.where(neg[:, None], rc, out)


# --------------------------------------------------------------------------- C3: FASTQ
def fastq_generate(n_reads, device, rlen=150, seed=7):
    """Fixed-shape synthetic FASTQ in HBM: `@SYN:1:FC:1:<tile4>:<x5>:<i9> 1:N:0:ACGT` headers (unique
    names, one space), rlen bases i.i.d. ACGT with 0.1 % N, '+', qualities uniform 35..70, LF.
    -> (blob uint8, dict of analytic columns as numpy int64)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    head = b"@SYN:1:FC:1:0000:00000:000000000 1:N:0:ACGT\n"
    hl = len(head)
    rec = hl + rlen + 1 + 2 + rlen + 1
    t = torch.empty((n_reads, rec), dtype=torch.uint8, device=device)
    t[:, :hl] = torch.frombuffer(bytearray(head), dtype=torch.uint8).to(device)
    idx = torch.arange(n_reads, device=device, dtype=torch.int64)
    for k in range(9):                                   # <i9>
        t[:, 31 - k] = (48 + (idx // (10 ** k)) % 10).to(torch.uint8)
    tile = (idx // 50_000) % 10_000
    for k in range(4):
        t[:, 15 - k] = (48 + (tile // (10 ** k)) % 10).to(torch.uint8)
    x = (idx * 7919) % 100_000
    for k in range(5):
        t[:, 21 - k] = (48 + (x // (10 ** k)) % 10).to(torch.uint8)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    step = 1 << 20
    for a in range(0, n_reads, step):
        b = min(n_reads, a + step)
        r = torch.randint(0, 4000, (b - a, rlen), device=device, generator=g)
        bases = lut[(r & 3).long()]
        bases[r >= 3996] = ord("N")
        t[a:b, hl:hl + rlen] = bases
        t[a:b, hl + rlen + 3:hl + 2 * rlen + 3] = torch.randint(35, 71, (b - a, rlen), device=device, generator=g,
                                                                dtype=torch.uint8)
    t[:, hl + rlen] = 10
    t[:, hl + rlen + 1] = ord("+")
    t[:, hl + rlen + 2] = 10
    t[:, rec - 1] = 10
    n_bytes = n_reads * rec
    blob = torch.zeros(n_bytes + 131072, dtype=torch.uint8, device=device)
    blob[:n_bytes] = t.view(-1)
    i = np.arange(n_reads, dtype=np.int64)
    cols = {"name_off": i * rec + 1, "name_len": np.full(n_reads, 31, np.int64), "dlen": np.full(n_reads, hl - 1, np.int64),
            "rlen": np.full(n_reads, rlen, np.int64), "soff": i * rec + hl, "qoff": i * rec + hl + rlen + 3,
            "n_bytes": n_bytes, "rec": rec}
    return blob, cols


# --------------------------------------------------------------------------- C4: BGZF framing
def bgzf_compress(raw, block=65280, level=6):
    """bgzip-compatible framing of `raw` (SAM spec 4.1): independent raw-deflate members of
    <= `block` input bytes, 18-byte header with the 'BC' BSIZE subfield, CRC32 + ISIZE trailer,
    and the 28-byte empty EOF member.  (No bgzip binary is needed.)"""
    import struct
    import zlib
    out = []
    for a in list(range(0, len(raw), block)) + [None]:
        chunk = b"" if a is None else raw[a:a + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        cd = co.compress(chunk) + co.flush()
        bsize = len(cd) + 25
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + cd +
                   struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    return b"".join(out)


# --------------------------------------------------------------------------- single-stream gzip, compressed in parallel
def _deflate_piece(args):
    import zlib
    chunk, last, level = args
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    return co.compress(chunk) + co.flush(zlib.Z_FINISH if last else zlib.Z_FULL_FLUSH)


def gzip_single_stream(raw, pool=None, piece=8 << 20, level=6):
    """ONE gzip member (one header, one deflate stream, one trailer) holding `raw`, compressed piece by piece the way pigz
    does it: every piece is deflated on its own and ends on a full flush (byte aligned, no final block), the last one
    finishes the stream; concatenated they are a single valid deflate stream.  pool: a multiprocessing pool (setup speed
    only).  Not BGZF: there are no member boundaries to restart from -- the shape zran-style checkpoints exist for."""
    import struct
    import zlib
    mv = memoryview(raw)
    jobs = [(bytes(mv[a:a + piece]), a + piece >= len(mv), level) for a in range(0, max(len(mv), 1), piece)]
    parts = pool.map(_deflate_piece, jobs, chunksize=1) if pool is not None else [_deflate_piece(j) for j in jobs]
    crc = 0
    for a in range(0, len(mv), 64 << 20):
        crc = zlib.crc32(mv[a:a + (64 << 20)], crc)
    return b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff" + b"".join(parts) + struct.pack("<II", crc & 0xFFFFFFFF, len(mv) & 0xFFFFFFFF)


# --------------------------------------------------------------------------- the same two, over all cores of the host (setup speed only)
# Threads, not processes: zlib releases the GIL while it deflates, so a thread pool compresses on all cores without a fork of
# the bench process (96 forks of a process with a HIP context and tens of GB mapped: 10 s of page-table copies) and without
# 3 GB going through pipes.
def _threads(n=None):
    import os
    from multiprocessing.pool import ThreadPool
    return ThreadPool(n or min(128, os.cpu_count() or 8))


_NATIVE = None


def _native():
    """csrc/libfxsynth.so (fxsynth.c: the same zlib calls from plain threads) or None when it has not been built."""
    global _NATIVE
    if _NATIVE is None:
        import ctypes as C
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libfxsynth.so")
        try:
            L = C.CDLL(path)
            i64, vp, i32 = C.c_int64, C.c_void_p, C.c_int
            L.fxs_bgzf_bound.restype = i64; L.fxs_bgzf_bound.argtypes = [i64, i32]
            L.fxs_gzip_bound.restype = i64; L.fxs_gzip_bound.argtypes = [i64, i64]
            L.fxs_bgzf_compress.restype = i64; L.fxs_bgzf_compress.argtypes = [vp, i64, vp, i64, i32, i32, i32]
            L.fxs_gzip_stream.restype = i64; L.fxs_gzip_stream.argtypes = [vp, i64, vp, i64, i64, i32, i32]
            _NATIVE = L
        except OSError:
            _NATIVE = False
    return _NATIVE or None


def _ncpu(procs):
    import os
    return int(procs or min(192, os.cpu_count() or 8))


def bgzf_compress_parallel(raw, procs=None, step=65280 * 16, block=65280, level=6):
    """`raw` (numpy uint8) BGZF-framed as bgzf_compress would frame it -> bytes-like (numpy uint8 from the native helper)."""
    L = _native()
    if L is not None:
        src = np.ascontiguousarray(np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw)
        dst = np.empty(int(L.fxs_bgzf_bound(src.size, block)), dtype=np.uint8)
        w = int(L.fxs_bgzf_compress(src.ctypes.data, src.size, dst.ctypes.data, dst.size, block, level, _ncpu(procs)))
        if w < 0:
            raise RuntimeError("fxs_bgzf_compress: %d" % w)
        return dst[:w]
    mv = memoryview(raw)
    with _threads(procs) as pool:
        parts = pool.map(lambda ab: bgzf_compress(mv[ab[0]:ab[1]], block, level)[:-28], [(a, min(a + step, len(raw))) for a in range(0, len(raw), step)], chunksize=1)
    return b"".join(parts) + bgzf_compress(b"")


def gzip_single_stream_parallel(raw, procs=None, piece=8 << 20, level=6):
    """gzip_single_stream of `raw` (numpy uint8), the pieces deflated and the CRC-32 of the trailer folded by a thread pool."""
    import struct
    import zlib
    L = _native()
    if L is not None:
        src = np.ascontiguousarray(np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw)
        dst = np.empty(int(L.fxs_gzip_bound(src.size, piece)), dtype=np.uint8)
        w = int(L.fxs_gzip_stream(src.ctypes.data, src.size, dst.ctypes.data, dst.size, piece, level, _ncpu(procs)))
        if w < 0:
            raise RuntimeError("fxs_gzip_stream: %d" % w)
        return dst[:w]
    mv = memoryview(raw)
    n = len(raw)
    with _threads(procs) as pool:
        res = pool.map_async(lambda a: _deflate_piece((mv[a:a + piece], a + piece >= n, level)), list(range(0, max(n, 1), piece)), chunksize=1)
        crc = 0
        for a in range(0, n, 64 << 20):
            crc = zlib.crc32(mv[a:a + (64 << 20)], crc)
        parts = res.get()
    return b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff" + b"".join(parts) + struct.pack("<II", crc & 0xFFFFFFFF, n & 0xFFFFFFFF)
