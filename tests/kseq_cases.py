"""Inputs for the Fastx (kseq_read) parity tests: random files in which everything kseq tolerates happens -- FASTA and
FASTQ records mixed, sequence and quality over several lines, empty lines, CR and CRLF line ends, lines that are one CR,
white space inside the lines, '@' / '>' / '+' at the start of quality lines, text between records, a last line without
a newline, files cut off anywhere -- and FIXED, hand-written cases for the rules that random text rarely meets.
Used by tests/ (oracle against the reference here, the HIP path against the oracle on the GPU box) and by
tools/kseq_line_model.py."""


def gen(rng):
    out = []

    def nl():
        return rng.choice(["\n", "\n", "\n", "\r\n"])

    def rline(alphabet, lo, hi):
        return "".join(rng.choice(alphabet) for _ in range(rng.randint(lo, hi)))

    mode = rng.random()
    if rng.random() < 0.2:
        out.append(rline("xyz >@+\r", 0, 5) + nl())
    for _ in range(rng.randint(0, 40 if mode < 0.5 else 8)):
        t = rng.random()
        name = rng.choice([">", "@"]) + rline("abc12", 0, 4)
        if rng.random() < 0.5:
            name += rng.choice([" ", "\t", "\r", "\x0b", " \r"]) + rline("desc \t", 0, 5)
        if mode < 0.25:                                  # mostly regular four-line records
            e = nl()
            ln = rng.randint(0 if rng.random() < 0.1 else 1, 9)
            s = rline("ACGT", ln, ln)
            q = rline("IJ@>+#", ln, ln) if rng.random() < 0.95 else rline("IJ", 0, 9)
            out.append(name + e + s + e + "+" + e + q + e)
            continue
        out.append(name + nl())
        for _ in range(rng.randint(0, 3 if mode > 0.5 else 70)):
            out.append(rline("ACGTN acgt\r\t", 0, 6) + nl())
            if rng.random() < 0.1:
                out.append(nl())
        if t < (0.6 if mode > 0.5 else 0.1):
            out.append("+" + rline("xy", 0, 2) + nl())
            for _ in range(rng.randint(0, 3)):
                out.append(rline("IJ@>+#\r", 0, 7) + nl())
        if rng.random() < 0.1:
            out.append(rline("junk@>", 0, 6) + nl())
    s = "".join(out)
    if rng.random() < 0.3 and s.endswith("\n"):
        s = s[:-1]
    if rng.random() < 0.1:
        s = s[:rng.randint(0, len(s))]
    return s.encode()




FIXED = [
    b"", b"\n", b">", b"@", b">\n", b"@a", b">a\n", b">a\nACGT", b">a\nACGT\n", b"junk\n>a\nAC\n", b"xx>a b\nAC\n",
    b">a\nAC\n\nGT\n>b\n\n\n", b">a \r\nAC\r\n", b">a\r\nAC\r\n>b c\r\nG\r\n",
    b"@r\nACGT\n+\nIIII\n", b"@r\nACGT\n+\nIIII", b"@r\nACGT\n+\nIII\n", b"@r\nACGT\n+\nIIIII\n", b"@r\nACGT\n+",
    b"@r\nACGT\n+\n", b"@r\nAC\nGT\n+r\nII\nII\n@s\nA\n+\nI\n", b"@r\nAC\nGT\n+\n@I\n>I\n@s\nA\n+\n+\n",
    b"@r\n\n+\n\n@s\nA\n+\nI\n", b"@r\n+\n", b"@r\nA\n+\nI\n@s\n+\n", b">f\nAC\n@q\nA\n+\nI\n>g\nT\n", b"@q\nA\n+\nI\n>g\nT\n+\n",
    b"@r\n\r\n+\n\r\n", b"@r\nA\r\r\n+\nI\r\r\n", b"@r\nAAAA\n+\nI\r\r\n\nII\n", b"@r\nAAA\n+\nI\r\r\n\n\nI\n", b">a\nAC\n\r",
    b">a\n\r\nA\n\r\n", b"@r x\ty\nA\n+\nI\n", b"@r\x0bz\nA\n+\nI\n", b">a\x00b c\nAC\x00GT\n", b"@a \nA\n+\nI\n@b\nC\n+\nJ\n",
    b"@a\nA\n+\nI\njunk@b\nC\n+\nJ\n", b"@a\nA\n+\nI\njunk\nmore @\n", b">a\n+\n", b">a\nAC GT\tN\n",
]
