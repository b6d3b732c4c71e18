"""TEST HELPER: numpy restatement of what one shard's HIP kernels produce
(line table, header list, local record rows, fx_shard_summary), so the
host-side stitch and the torch.distributed plumbing can be tested on CPU
(gloo) without a GPU.  Not part of the product."""
import numpy as np

from pyfastx_amd.shard import FIELDS, Summary


def local_scan(raw, lo, hi, full_name=False):
    """-> (rows dict of lists for records whose '>' lies in [lo,hi), Summary)."""
    n_total = len(raw)
    a = np.frombuffer(raw, dtype=np.uint8)[lo:hi]
    n = hi - lo
    is_last = hi == n_total
    prev = raw[lo - 1] if lo else 10
    nl = (np.flatnonzero(a == 10) + lo).tolist()
    if is_last and n and a[-1] != 10:
        nl.append(n_total)                                 # virtual EOF newline
    gt = np.flatnonzero(a == ord(">"))
    hdr = [int(p) + lo for p in gt if (raw[lo + p - 1] if (lo + p) > 0 else 10) == 10 and (p > 0 or prev == 10)]
    nl_arr = np.array(nl, dtype=np.int64)
    rows = {k: [] for k in ("hoff", "boff", "blen", "slen", "llen", "elen", "norm", "dlen", "name_len", "reg")}

    def byte_at(p):                                        # what this shard can see of the stream
        return raw[p] if lo <= p < hi else None
    tail = dict(tail_e=-1, tail_first_end=-1, tail_nl_after=0, tail_bad=0, tail_elen=0, tail_dlen=-1, tail_name_len=-1)
    for k, h in enumerate(hdr):
        L = int(np.searchsorted(nl_arr, h, side="left"))
        if L >= len(nl):                                   # header unterminated in this shard
            seg = raw[h + 1:hi]
            ws = -1
            if not full_name:
                for j, c in enumerate(seg):
                    if c in (32, 9):
                        ws = j
                        break
            vals = dict(hoff=h, boff=0, blen=0, slen=0, llen=0, elen=0, norm=1, dlen=-1, name_len=ws, reg=0)
            for kk, v in vals.items():
                rows[kk].append(v)
            tail.update(tail_dlen=-1, tail_name_len=ws)
            continue
        e = nl[L]
        elen = 2 if raw[e - 1] == 13 else 1
        dlen = (e - h) - elen
        name_len = dlen
        if not full_name:
            for j in range(dlen):
                if raw[h + 1 + j] in (32, 9):
                    name_len = j
                    break
        if k + 1 < len(hdr):
            hn = hdr[k + 1]
            Ln = int(np.searchsorted(nl_arr, hn, side="left"))
        else:
            hn, Ln = nl[-1] + 1, len(nl)
        nseq = Ln - L - 1
        boff = e + 1
        blen = hn - boff
        llen = nl[L + 1] - nl[L] if nseq > 0 else 0
        bad = sum(1 for i in range(L + 2, Ln) if nl[i] - nl[i - 1] != llen)
        from pyfastx_amd.shard import line_regular_rule
        reg = line_regular_rule(boff, blen, blen - elen * nseq, llen, elen, 0 if bad > 1 else 1, byte_at)
        vals = dict(hoff=h, boff=boff, blen=blen, slen=blen - elen * nseq, llen=llen, elen=elen,
                    norm=0 if bad > 1 else 1, dlen=dlen, name_len=name_len, reg=int(bad == 0) if reg is None else int(reg))
        for kk, v in vals.items():
            rows[kk].append(v)
        if k == len(hdr) - 1:
            tail.update(tail_e=e, tail_first_end=nl[L + 1] if L + 1 < len(nl) else -1, tail_nl_after=len(nl) - L - 1,
                        tail_bad=bad, tail_elen=elen, tail_dlen=dlen, tail_name_len=name_len)
    lead_nl = int(np.searchsorted(nl_arr, hdr[0], side="left")) if hdr else len(nl)
    d = [nl[i] - nl[i - 1] for i in range(1, lead_nl)]
    v1 = c1 = v2 = c2 = 0
    if d:
        v1 = d[0]
        c1 = sum(1 for x in d if x == v1)
        rest = [x for x in d if x != v1]
        if rest:
            v2 = rest[0]
            c2 = sum(1 for x in d if x == v2)
    first_nl = nl[0] if nl else -1
    lim = (first_nl - lo) if first_nl >= 0 else n
    ws = -1
    for j in range(lim):
        if a[j] in (32, 9):
            ws = lo + j
            break
    s = dict(base=lo, n_bytes=n, is_last=int(is_last), n_nl=len(nl), first_nl=first_nl,
             second_nl=nl[1] if len(nl) > 1 else -1, last_nl=nl[-1] if nl else -1,
             first_nl_prev=int(raw[first_nl - 1]) if first_nl > lo else -1,
             first_byte=int(a[0]), last_byte=int(a[-1]), n_hdr=len(hdr), first_hdr=hdr[0] if hdr else -1,
             last_hdr=hdr[-1] if hdr else -1, lead_nl=lead_nl, lead_ws=ws, lead_v1=v1, lead_c1=c1, lead_v2=v2,
             lead_c2=c2, lead_prev_nl=nl[lead_nl - 2] if lead_nl >= 2 else -1,
             second_last_nl=nl[-2] if len(nl) >= 2 else -1)
    s.update(tail)
    return rows, Summary((k, int(s[k])) for k in FIELDS)


def stitched_rows(raw, cuts, full_name=False, gather=None):
    """All shards on this process (gather=None) -> concatenated final rows."""
    from pyfastx_amd import shard
    bounds = [0] + list(cuts) + [len(raw)]
    local = [local_scan(raw, bounds[i], bounds[i + 1], full_name) for i in range(len(bounds) - 1)]
    S = [s for _, s in local]
    out = {k: [] for k in local[0][0]}
    for r, (rows, _) in enumerate(local):
        fix = shard.stitch_tail(S, r, full_name)
        if fix is not None:
            for k, v in fix.items():
                rows[k][-1] = v
        for k in out:
            out[k].extend(rows[k])
    return out
