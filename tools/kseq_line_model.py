"""Executable model of k_kseq_walk (pyfastx_amd/csrc/fx_kseq.hpp): kseq_read (kseq.c:138-179) restated over a LINE
TABLE instead of a byte stream, with the two 64-line window steps the kernel takes -- 16 four-line FASTQ records at
once, a run of FASTA header / sequence lines at once -- and the one-line-at-a-time step for everything else; in front of
it the passes that take a file's regular prefix with every line judged by itself (prefix()).

A development tool: the kernel is a transliteration of walk() below, and `python tools/kseq_line_model.py [seed] [n]`
fuzzes the model against the byte-level oracle (oracle/fx_oracle.c: fxo_kseq, itself pinned against the compiled
reference's Fastx), so the line formulation is checked on the CPU before it runs on a GPU.  Not imported by the product.
"""
import os
import random
import sys

SEEK, HDR, SEQ, QUAL = 0, 1, 2, 3
C_SEQ, C_QUAL = 1, 2
F_FASTQ, F_UNTOUCHED, F_HDR_UNTERM = 1, 2, 4
BIG = 1 << 25
REG_MAX = 1 << 20
W = 64


def line_table(data):
    n = len(data)
    nl = [i for i, b in enumerate(data) if b == 10]
    if n > 0 and data[-1] != 10:
        nl.append(n)                                   # the last, unterminated line
    lines, start = [], 0
    for e in nl:
        ln = e - start
        lines.append((start, ln, data[start] if ln else 0, data[e - 1] if ln else 0, e == n))
        start = e + 1
    return lines


def con0(t):
    return t[1] - (1 if (t[3] == 13 and t[1] > 1) else 0)


def is_hdr(t):
    return t[1] >= 1 and t[2] in (62, 64)


def prefix(T, recs, ldst, lcls, lcon):
    """The regular prefix, every line judged by itself (k_kq_classify, k_kq_fq_*, k_kq_fa_*): -> the walker's entry state
    (j, st, S, cur, acc), records and line entries filled in."""
    L = len(T)
    firstq = firsta = L
    for i, t in enumerate(T):
        s, ln, f, la, un = t
        role = i & 3
        if role == 0:
            ok = is_hdr(t)
        elif role == 1:
            ok = ln >= 1 and f not in (62, 64, 43)
        elif role == 2:
            ok = ln >= 1 and f == 43 and not un
        else:
            ok = con0(t) == con0(T[i - 2])
        if ln >= REG_MAX:
            ok = False
        if not ok:
            firstq = min(firstq, i)
        if (ln >= 1 and f == 43) or (ln == 1 and f == 13) or un or ln >= REG_MAX:
            firsta = min(firsta, i)
    R = firstq // 4
    if R >= 1:
        S = 0
        for r in range(R):
            h, c = T[4 * r], con0(T[4 * r + 1])
            recs.append(dict(hdr_off=h[0] + 1, hdr_len=h[1] - 1, hdr_line=4 * r, flags=F_FASTQ, seq_len=c, s_n=1, q_n=1, seq_cum=S))
            ldst[4 * r + 1], lcls[4 * r + 1], lcon[4 * r + 1] = S, C_SEQ, c
            ldst[4 * r + 3], lcls[4 * r + 3], lcon[4 * r + 3] = S, C_QUAL, c
            S += c
        return 4 * R, SEEK, S, None, 0
    if L and is_hdr(T[0]) and firsta >= 1:
        n = firsta
        H = [is_hdr(T[i]) for i in range(n)]
        con = [0 if (H[i] or T[i][1] == 0) else T[i][1] - (1 if T[i][3] == 13 else 0) for i in range(n)]
        coff = [0]
        for c in con:
            coff.append(coff[-1] + c)
        hpos = [i for i in range(n) if H[i]]
        for i in range(n):
            if not H[i] and T[i][1] > 0:
                ldst[i], lcls[i], lcon[i] = coff[i], C_SEQ, con[i]
        for r, i in enumerate(hpos[:-1]):
            i2 = hpos[r + 1]
            recs.append(dict(hdr_off=T[i][0] + 1, hdr_len=T[i][1] - 1, hdr_line=i, flags=0, seq_len=coff[i2] - coff[i], s_n=i2 - i - 1, q_n=0, seq_cum=coff[i]))
        i = hpos[-1]
        return n, SEQ, coff[i], dict(hdr_off=T[i][0] + 1, hdr_len=T[i][1] - 1, hdr_line=i, flags=0), coff[n] - coff[i]
    return 0, SEEK, 0, None, 0


def walk(data, fast=True, use_prefix=True):
    n = len(data)
    T = line_table(data)
    L = len(T)
    ldst, lcls, lcon = [0] * L, [0] * L, [0] * L
    recs = []
    st, j, S, code = SEEK, 0, 0, None
    cur = None                                           # open record: dict(hdr_off, hdr_len, hdr_line, flags)
    acc = qacc = qn = sn = tcr = lastc = 0
    if use_prefix:
        j, st, S, cur, acc = prefix(T, recs, ldst, lcls, lcon)

    def emit(seq_len, s_n, q_n, flags, seq_cum):
        r = dict(cur)
        r.update(seq_len=seq_len, s_n=s_n, q_n=q_n, seq_cum=seq_cum)
        r["flags"] |= flags
        recs.append(r)

    while j < L and code is None:
        win = T[j:j + W]
        big = any(t[1] >= BIG for t in win)
        if fast and st == SEEK and not big:
            # ---- 16 four-line records at once
            ok = []
            for k, (s, ln, f, la, un) in enumerate(win):
                role = k & 3
                con = ln - (1 if (la == 13 and ln > 1) else 0)
                if role == 0:
                    ok.append(ln >= 1 and f in (62, 64))
                elif role == 1:
                    ok.append(ln >= 1 and f not in (62, 64, 43))
                elif role == 2:
                    ok.append(ln >= 1 and f == 43 and not un)
                else:
                    p = win[k - 2]
                    ok.append(con == p[1] - (1 if (p[3] == 13 and p[1] > 1) else 0))
            R = 0
            while 4 * R + 3 < len(win) and all(ok[4 * R:4 * R + 4]):
                R += 1
            if R:
                for r in range(R):
                    h, b, d = win[4 * r], win[4 * r + 1], win[4 * r + 3]
                    con = b[1] - (1 if (b[3] == 13 and b[1] > 1) else 0)
                    cur = dict(hdr_off=h[0] + 1, hdr_len=h[1] - 1, hdr_line=j + 4 * r, flags=0)
                    ldst[j + 4 * r + 1], lcls[j + 4 * r + 1], lcon[j + 4 * r + 1] = S, C_SEQ, con
                    ldst[j + 4 * r + 3], lcls[j + 4 * r + 3], lcon[j + 4 * r + 3] = S, C_QUAL, con
                    emit(con, 1, 1, F_FASTQ, S)
                    S += con
                j += 4 * R
                continue
        if st == SEEK:
            s, ln, f, la, un = T[j]
            if ln >= 1 and f in (62, 64):
                st = HDR
                continue
            p = next((q for q in range(s, s + ln) if data[q] in (62, 64)), -1)
            if p >= 0:
                if p + 1 >= n:
                    code = -1
                    break
                cur = dict(hdr_off=p + 1, hdr_len=s + ln - (p + 1), hdr_line=j, flags=F_HDR_UNTERM if un else 0)
                st, acc = SEQ, 0
            j += 1
            continue
        if fast and st in (HDR, SEQ) and not big:
            # ---- a run of header / sequence lines at once
            m = 0
            for (s, ln, f, la, un) in win:
                if (ln >= 1 and f == 43) or (ln == 1 and f == 13) or un:
                    break
                m += 1
            if m:
                H = [win[k][1] >= 1 and win[k][2] in (62, 64) for k in range(m)]
                con = [0 if (H[k] or win[k][1] == 0) else win[k][1] - (1 if win[k][3] == 13 else 0) for k in range(m)]
                pin, t = [], 0
                for c in con:
                    t += c
                    pin.append(t)
                pex = [pin[k] - con[k] for k in range(m)]
                hs = [k for k in range(m) if H[k]]
                carried = st == SEQ
                carry = acc if carried else 0
                for k in range(m):
                    if not H[k] and win[k][1] > 0:
                        ldst[j + k], lcls[j + k], lcon[j + k] = S + carry + pex[k], C_SEQ, con[k]
                if hs:
                    if carried:
                        emit(acc + pex[hs[0]], (j + hs[0]) - cur["hdr_line"] - 1, 0, 0, S)
                    for r, h in enumerate(hs):
                        cur = dict(hdr_off=win[h][0] + 1, hdr_len=win[h][1] - 1, hdr_line=j + h, flags=0)
                        if r + 1 < len(hs):
                            h2 = hs[r + 1]
                            emit(pex[h2] - pin[h], h2 - h - 1, 0, 0, S + carry + pin[h])
                    S = S + carry + pin[hs[-1]]
                    acc = pin[m - 1] - pin[hs[-1]]
                else:
                    acc += pin[m - 1]
                st = SEQ
                j += m
                continue
        # ---- one line
        s, ln, f, la, un = T[j]
        if st == HDR:
            if un and ln == 1:
                code = -1
                break
            cur = dict(hdr_off=s + 1, hdr_len=ln - 1, hdr_line=j, flags=F_HDR_UNTERM if un else 0)
            st, acc = SEQ, 0
            j += 1
        elif st == SEQ:
            if ln == 0:
                j += 1
            elif f in (62, 64):
                emit(acc, j - cur["hdr_line"] - 1, 0, 0, S)
                S += acc
                st = HDR
            elif f == 43:
                if un:
                    code = -2
                    break
                sn = j - cur["hdr_line"] - 1
                st, qacc, qn, tcr, lastc = QUAL, 0, 0, 0, j
                j += 1
            else:
                # the first byte goes in by itself (kseq.c:156), the strip happens in the call for the rest of the line --
                # which returns early when nothing at all is left: a lone CR as the last byte of the stream stays
                con = ln - (1 if (la == 13 and acc + ln > 1 and not (un and ln == 1)) else 0)
                ldst[j], lcls[j], lcon[j] = S + acc, C_SEQ, con
                acc += con
                j += 1
        else:                                            # QUAL
            # ks_getuntil2 strips ONE trailing CR per call from a string longer than one byte (kseq.c:106) -- also in a
            # call that appends nothing: an empty line takes a CR that a line ending in "\r\r" left behind.  tcr = the
            # run of CRs the quality string ends with, lastc = the last line that holds bytes of it.
            tr = 0
            while tr < ln and data[s + ln - 1 - tr] == 13:
                tr += 1
            con = ln
            qacc += ln
            tcr = tcr + ln if tr == ln else tr
            if tcr >= 1 and qacc > 1:
                qacc -= 1
                tcr -= 1
                if ln:
                    con -= 1
                else:
                    while lcon[lastc] == 0:
                        lastc -= 1
                    lcon[lastc] -= 1
            ldst[j], lcls[j], lcon[j] = S + qacc - con, C_QUAL, con
            if con:
                lastc = j
            qn += 1
            j += 1
            if qacc >= acc:
                if qacc != acc:
                    code = -2
                    break
                emit(acc, sn, qn, F_FASTQ, S)
                S += acc
                st = SEEK
    if code is None:
        if st == SEQ:
            emit(acc, L - cur["hdr_line"] - 1, 0, 0, S)
            S += acc
            code = -1
        elif st == QUAL:
            if qn == 0 and acc == 0:
                emit(0, sn, 0, F_FASTQ | F_UNTOUCHED, S)
                code = -1
            else:
                code = -2
        else:
            code = -1
    return T, recs, ldst, lcls, lcon, S, code


def materialise(data, fast=True, use_prefix=True):
    """-> list of (header bytes, flags, seq bytes, qual bytes or None), end code: through the per-line arrays, as
    k_kseq_gather copies."""
    T, recs, ldst, lcls, lcon, S, code = walk(data, fast, use_prefix)
    out = []
    for r in recs:
        seq, qual = bytearray(r["seq_len"]), bytearray(r["seq_len"])
        last = r["hdr_line"] + r["s_n"] + ((1 + r["q_n"]) if r["flags"] & F_FASTQ else 0)
        for i in range(r["hdr_line"] + 1, last + 1):
            if lcls[i]:
                o = ldst[i] - r["seq_cum"]
                piece = data[T[i][0]:T[i][0] + lcon[i]]
                (seq if lcls[i] == C_SEQ else qual)[o:o + lcon[i]] = piece
        hdr = bytes(data[r["hdr_off"]:r["hdr_off"] + r["hdr_len"]])
        out.append((hdr, r["flags"], bytes(seq), bytes(qual) if r["flags"] & F_FASTQ else None))
    return out, code


SPACE = b" \t\n\r\x0b\x0c"


def header_parts(hdr, unterminated):
    """(name, comment or None = comment buffer untouched): kseq.c:148-149."""
    for i, c in enumerate(hdr):
        if c in SPACE:
            rest = hdr[i + 1:]
            if unterminated and not rest:
                return hdr[:i], None
            if len(rest) > 1 and rest.endswith(b"\r"):
                rest = rest[:-1]
            return hdr[:i], rest
    return hdr, None


def main():
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sys.path.insert(0, os.path.join(root, "oracle"))
    sys.path.insert(0, os.path.join(root, "tests"))
    import fxoracle
    from kseq_cases import gen
    rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    bad = 0
    for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2000):
        b = gen(rng)
        recs, seq, qual, code = fxoracle.kseq(b)
        want = []
        for r in recs:
            com = None if r["com_len"] < 0 else b[r["com_off"]:r["com_off"] + r["com_len"]]
            q = None if r["qual_len"] == -1 else bytes(qual[r["qual_off"]:r["qual_off"] + max(int(r["qual_len"]), 0)])
            want.append((b[r["name_off"]:r["name_off"] + r["name_len"]], com, bytes(seq[r["seq_off"]:r["seq_off"] + r["seq_len"]]), q,
                         r["qual_len"] == -2))
        for fast, pre in ((True, True), (True, False), (False, False)):
            got, gcode = materialise(b, fast, pre)
            have = []
            for hdr, fl, s, q in got:
                nm, cm = header_parts(hdr, bool(fl & F_HDR_UNTERM))
                have.append((nm, cm, s, q, bool(fl & F_UNTOUCHED)))
            if have != want or gcode != code:
                bad += 1
                if bad < 5:
                    print("MISMATCH fast=%s" % fast, repr(b), "\n want", want, code, "\n have", have, gcode)
    print("bad", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
