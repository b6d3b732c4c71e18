// tools/deflate_symbol_mix.c -- what a BGZF member of genome text is made of (VERDICT r4 #2 proposed root-table entries that
// carry up to three literals): literal / match symbols, how many literals stand next to another literal, bits per symbol.
// Text: 65280-byte members of hg38-shaped FASTA (60-column lines, i.i.d. bases with the genome's skew, soft-masked stretches,
// N runs), raw deflate at a zlib level, parsed with a plain bit-by-bit inflate of its own.   usage: deflate_symbol_mix [level] [members]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
typedef struct { short count[16], symbol[320]; } huff;
static const uint8_t *in; static long inbits;
static int bit(long p) { return p < inbits ? (in[p >> 3] >> (p & 7)) & 1 : 0; }
static long bits(long *p, int n) { long v = 0; for (int i = 0; i < n; i++) v |= (long)bit((*p)++) << i; return v; }
static int decode(long *p, const huff *h) { int code = 0, first = 0, index = 0; for (int len = 1; len <= 15; len++) { code |= bit((*p)++); int c = h->count[len]; if (code - c < first) return h->symbol[index + (code - first)]; index += c; first += c; first <<= 1; code <<= 1; } return -10; }
static void construct(huff *h, const short *length, int n) { short offs[16]; memset(h->count, 0, sizeof h->count); for (int s = 0; s < n; s++) h->count[length[s]]++; offs[1] = 0; for (int len = 1; len < 15; len++) offs[len + 1] = offs[len] + h->count[len]; for (int s = 0; s < n; s++) if (length[s]) h->symbol[offs[length[s]]++] = s; }
static const short lens[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const short lext[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const short dext[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
int main(int argc, char **argv) {
    int level = argc > 1 ? atoi(argv[1]) : 6, nmem = argc > 2 ? atoi(argv[2]) : 300;
    srand(7);
    long nlit = 0, nmatch = 0, litbits = 0, matchbits = 0, matchbytes = 0, lit_after_lit = 0, lit_runs = 0, inrun = 0, cbytes = 0, run3 = 0;
    for (int m = 0; m < nmem; m++) {
        static uint8_t raw[65280], cmp[70000];
        int lower = 0, col = 0, nrun = 0;
        for (int i = 0; i < 65280; i++) {
            if (col == 60) { raw[i] = '\n'; col = 0; continue; }
            if (rand() % 20000 == 0) lower = !lower;
            if (rand() % 100000 == 0) nrun = rand() % 20000;
            char c = "ACGT"[(rand() % 1000 < 295) ? 0 : (rand() % 705 < 205 ? 1 : (rand() % 500 < 205 ? 2 : 3))];
            if (nrun > 0) { c = 'N'; nrun--; }
            raw[i] = lower ? c + 32 : c; col++;
        }
        z_stream z; memset(&z, 0, sizeof z); deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        z.next_in = raw; z.avail_in = 65280; z.next_out = cmp; z.avail_out = sizeof cmp; deflate(&z, Z_FINISH); long clen = z.total_out; deflateEnd(&z);
        cbytes += clen; in = cmp; inbits = clen * 8; long p = 0; int last;
        do {
            last = bits(&p, 1); int type = bits(&p, 2);
            if (type == 0) { p = (p + 7) & ~7L; long len = bits(&p, 16); bits(&p, 16); p += len * 8; continue; }
            huff lc, dc; short lengths[320];
            if (type == 1) { int s = 0; for (; s < 144; s++) lengths[s] = 8; for (; s < 256; s++) lengths[s] = 9; for (; s < 280; s++) lengths[s] = 7; for (; s < 288; s++) lengths[s] = 8; construct(&lc, lengths, 288); for (s = 0; s < 30; s++) lengths[s] = 5; construct(&dc, lengths, 30); }
            else {
                int nlen = bits(&p, 5) + 257, ndist = bits(&p, 5) + 1, ncode = bits(&p, 4) + 4; static const short order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
                int idx; for (idx = 0; idx < ncode; idx++) lengths[order[idx]] = bits(&p, 3); for (; idx < 19; idx++) lengths[order[idx]] = 0; construct(&lc, lengths, 19); idx = 0;
                while (idx < nlen + ndist) { int s = decode(&p, &lc); if (s < 16) lengths[idx++] = s; else { int len = 0, rep; if (s == 16) { len = lengths[idx - 1]; rep = 3 + bits(&p, 2); } else if (s == 17) rep = 3 + bits(&p, 3); else rep = 11 + bits(&p, 7); while (rep--) lengths[idx++] = len; } }
                construct(&lc, lengths, nlen); construct(&dc, lengths + nlen, ndist);
            }
            inrun = 0;
            for (;;) {
                long q0 = p; int s = decode(&p, &lc);
                if (s < 0) { fprintf(stderr, "decode error\n"); return 1; }
                if (s == 256) break;
                if (s < 256) { nlit++; litbits += p - q0; if (inrun) lit_after_lit++; else lit_runs++; inrun++; if (inrun == 3) run3++; }
                else { s -= 257; int len = lens[s] + bits(&p, lext[s]); int d = decode(&p, &dc); bits(&p, dext[d]); nmatch++; matchbits += p - q0; matchbytes += len; inrun = 0; }
            }
        } while (!last);
    }
    long nsym = nlit + nmatch;
    printf("zlib level %d, %d members of 65280 bytes -> %.1f %% of their size\n", level, nmem, 100.0 * cbytes / (65280.0 * nmem));
    printf("symbols per member %.0f: literals %.1f %% (%.2f bits each), matches %.1f %% (%.2f bits, %.2f bytes each)\n", (double)nsym / nmem, 100.0 * nlit / nsym,
           (double)litbits / (nlit ? nlit : 1), 100.0 * nmatch / nsym, (double)matchbits / (nmatch ? nmatch : 1), (double)matchbytes / (nmatch ? nmatch : 1));
    printf("output bytes from literals %.1f %%; literals that follow a literal: %.1f %% of all symbols (runs of literals: %.2f long on average, %.1f %% of them reach three)\n",
           100.0 * nlit / (nlit + matchbytes), 100.0 * lit_after_lit / nsym, (double)nlit / (lit_runs ? lit_runs : 1), 100.0 * run3 / (lit_runs ? lit_runs : 1));
    printf("=> a table entry that carries up to 3 literals saves at most %.1f %% of the decode steps\n", 100.0 * lit_after_lit / nsym);
    return 0;
}
