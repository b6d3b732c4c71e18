// compbench.hip -- stand-alone check + timing harness for k_fasta_comp (fx_comp.hpp).  Not part of the product.
// (1) v_perm_b32 table self-test; (2) a host-built buffer with every awkward shape (tiny records, header lines that
// cross granules, CRLF, IUPAC / protein letters, '*', bytes >= 128, NUL, long N runs, soft-masked runs, no trailing
// newline) is counted on the GPU and compared bin by bin with a scalar CPU count; (3) a 3 GB hg38-shaped buffer
// is timed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/compbench tools/compbench.hip && /tmp/compbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include "../pyfastx_amd/csrc/fx_comp.hpp"

using namespace fx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static uint32_t rng_state = 777;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

__global__ void k_selftest(const uint32_t *in, uint32_t *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t h, hl, d;
    comp_classify(in[i], h, hl, d);
    out[3 * i] = h; out[3 * i + 1] = hl; out[3 * i + 2] = d;
}

static int selftest() {
    std::vector<uint32_t> in;
    for (uint32_t b = 0; b < 256; ++b) in.push_back(b | ((255 - b) << 8) | (((b * 7 + 3) & 255) << 16) | (((b * 13 + 101) & 255) << 24));
    uint32_t *di, *dout;
    CK(hipMalloc((void **)&di, in.size() * 4)); CK(hipMalloc((void **)&dout, in.size() * 12));
    CK(hipMemcpy(di, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_selftest, dim3(1), dim3(256), 0, 0, di, dout, (int)in.size());
    std::vector<uint32_t> out(in.size() * 3);
    CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (size_t i = 0; i < in.size(); ++i)
        for (int k = 0; k < 4; ++k) {
            const uint8_t b = (in[i] >> (8 * k)) & 255;
            const uint8_t h = (out[3 * i] >> (8 * k)) & 255, hl = (out[3 * i + 1] >> (8 * k)) & 255, d = (out[3 * i + 2] >> (8 * k)) & 255;
            const char *set = "ACGTNacgtn\n\r";
            const bool valid = b && strchr(set, b);
            uint8_t wh = 0;
            switch (b & 0xDF) { case 'A': wh = 1; break; case 'C': wh = 2; break; case 'G': wh = 4; break; case 'T': wh = 8; break; case 'N': wh = 16; break; }
            if (b == 13) wh = 32;
            const bool lower = valid && b >= 'a';
            if (valid ? (d != 0 || h != wh || hl != (lower ? wh : 0)) : d == 0) {
                if (bad++ < 10) printf("selftest byte %02x: h %02x hl %02x d %02x (valid %d)\n", b, h, hl, d, (int)valid);
            }
        }
    printf("selftest: %s\n", bad ? "FAILED" : "ok");
    return bad != 0;
}

struct Built { std::vector<uint8_t> b; std::vector<int64_t> hdr, boff; };

static void add_record(Built &B, const std::string &name, size_t len, int width, int kind) {
    B.hdr.push_back((int64_t)B.b.size());
    B.b.push_back('>');
    B.b.insert(B.b.end(), name.begin(), name.end());
    const bool crlf = kind == 3;
    if (crlf) B.b.push_back('\r');
    B.b.push_back('\n');
    B.boff.push_back((int64_t)B.b.size());
    const char dna[4] = {'A', 'C', 'G', 'T'};
    const char *iupac = "RYKMSWBDHVU-*.xX";
    const char *prot = "ACDEFGHIKLMNPQRSTVWY*";
    size_t col = 0;
    bool lower = false;
    size_t nrun = 0;
    for (size_t i = 0; i < len; ++i) {
        if ((rnd() & 255) == 0) lower = !lower;
        if (nrun == 0 && (rnd() & 8191) == 0) nrun = 1 + rnd() % 3000;
        uint8_t c = (uint8_t)dna[rnd() & 3];
        if (nrun) { c = 'N'; --nrun; }
        if (lower) c |= 0x20;
        if (kind == 1 && (rnd() & 63) == 0) c = (uint8_t)iupac[rnd() & 15];
        if (kind == 2) c = (uint8_t)prot[rnd() % 21];
        if (kind == 4 && (rnd() & 1023) == 0) c = (uint8_t)(rnd() & 255);        // anything, incl. NUL, >= 128, '>' inside a line
        if (c == '\n') c = 'A';
        if (kind == 4 && c == '>' && col == 0) c = 'A';
        B.b.push_back(c);
        if (++col == (size_t)width) { if (crlf) B.b.push_back('\r'); B.b.push_back('\n'); col = 0; }
    }
    if (col) { if (crlf) B.b.push_back('\r'); B.b.push_back('\n'); }
}

static void cpu_count(const Built &B, std::vector<uint64_t> &comp) {
    const int64_t n = (int64_t)B.b.size(), nr = (int64_t)B.hdr.size();
    comp.assign((size_t)nr * 128, 0);
    for (int64_t r = 0; r < nr; ++r) {
        const int64_t e = r + 1 < nr ? B.hdr[r + 1] : n;
        for (int64_t p = B.boff[r]; p < e; ++p) { const uint8_t c = B.b[p]; if (c != '\n' && c < 128) ++comp[r * 128 + c]; }
    }
}

static int run(const Built &B, bool verify, int reps, const char *what) {
    const int64_t n = (int64_t)B.b.size(), nr = (int64_t)B.hdr.size();
    const int64_t ngran = n / FX_GRAN + 1;
    std::vector<int64_t> hp((size_t)ngran + 1, 0);
    for (int64_t r = 0; r < nr; ++r) ++hp[(size_t)(B.hdr[r] / FX_GRAN) + 1];
    for (int64_t g = 0; g < ngran; ++g) hp[g + 1] += hp[g];
    uint8_t *d; int64_t *dh, *db, *dp; unsigned long long *dc; int32_t *de;
    CK(hipMalloc((void **)&de, (size_t)(ngran + 2) * 4));
    CK(hipMalloc((void **)&d, (size_t)n + 64)); CK(hipMalloc((void **)&dh, nr * 8)); CK(hipMalloc((void **)&db, nr * 8));
    CK(hipMalloc((void **)&dp, (ngran + 1) * 8)); CK(hipMalloc((void **)&dc, nr * 128 * 8));
    CK(hipMemcpy(d, B.b.data(), n, hipMemcpyHostToDevice)); CK(hipMemcpy(dh, B.hdr.data(), nr * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, B.boff.data(), nr * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dp, hp.data(), (ngran + 1) * 8, hipMemcpyHostToDevice));
    int rc = 0;
    for (int gpw : {4 * COMP_DEPTH, 5, 1, 8 * COMP_DEPTH, 2 * COMP_DEPTH}) {
        if (!verify && (gpw == 5 || gpw == 1)) continue;
        const int64_t waves = (ngran + gpw - 1) / gpw;
        const unsigned nb = (unsigned)((waves + COMP_WPB - 1) / COMP_WPB);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int it = 0; it < reps; ++it) {
            CK(hipMemset(dc, 0, nr * 128 * 8));
            CK(hipEventRecord(e0));
            CK(hipMemsetAsync(de, 0, 4, 0));
            hipLaunchKernelGGL(k_fasta_comp<true>, dim3(nb), dim3(COMP_WPB * 64), 0, 0, d, n, (int64_t)0, dh, db, nr, dp, ngran, gpw, de, (int64_t)-1, dc);
            hipLaunchKernelGGL(k_fasta_comp<false>, dim3(nb), dim3(COMP_WPB * 64), 0, 0, d, n, (int64_t)0, dh, db, nr, dp, ngran, gpw, de, (int64_t)-1, dc);
            hipLaunchKernelGGL(k_fasta_comp_small, dim3(2048), dim3(BLOCK), 0, 0, d, n, (int64_t)0, dh, db, nr, dc);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        }
        CK(hipGetLastError());
        int32_t nedge = 0; CK(hipMemcpy(&nedge, de, 4, hipMemcpyDeviceToHost));
        printf("%s: n = %.3f GB, %lld records, gpw %d (%d of %lld runs to the edge kernel): %.3f ms (%.2f TB/s)\n", what, n / 1e9, (long long)nr, gpw, nedge, (long long)waves, best, n / (best * 1e-3) / 1e12);
        if (verify && !FX_COMP_PROBE) {
            std::vector<uint64_t> want, got((size_t)nr * 128);
            cpu_count(B, want);
            CK(hipMemcpy(got.data(), dc, got.size() * 8, hipMemcpyDeviceToHost));
            int bad = 0;
            for (size_t i = 0; i < got.size(); ++i)
                if (got[i] != want[i] && bad++ < 12) printf("  rec %zu byte %zu ('%c'): got %llu want %llu\n", i / 128, i % 128, (int)(i % 128) >= 32 ? (int)(i % 128) : '?',
                                                           (unsigned long long)got[i], (unsigned long long)want[i]);
            printf("  verify: %s (%d differing bins)\n", bad ? "FAILED" : "ok", bad);
            rc |= bad != 0;
        }
    }
    hipFree(de); hipFree(d); hipFree(dh); hipFree(db); hipFree(dp); hipFree(dc);
    return rc;
}

int main(int argc, char **argv) {
    int rc = selftest();
    {
        Built B;
        const char pre[] = "junk before the first header\nACGT\n";
        B.b.insert(B.b.end(), pre, pre + sizeof pre - 1);
        int rec = 0;
        while (B.b.size() < (size_t)96 << 20) {
            const int kind = rec % 7 == 6 ? 4 : rec % 7 % 5;            // 0 plain 1 iupac 2 protein 3 crlf 4 noise
            const int shape = rec % 11;
            size_t len = shape == 0 ? (4u << 20) + rnd() % 5000 : shape == 1 ? 0 : shape == 2 ? 1 + rnd() % 40 : shape == 3 ? 4000 + rnd() % 300 :
                         shape == 4 ? 70000 + rnd() % 70000 : shape == 5 ? 300000 : 200 + rnd() % 3000;
            const int width = shape == 5 ? 1000000 : shape == 6 ? 7 : shape == 7 ? 4095 : 60;
            std::string name = "seq" + std::to_string(rec) + " kind " + std::to_string(kind);
            if (shape == 8) name += std::string(9000, 'x');            // a header line longer than two granules
            add_record(B, name, len, width, kind);
            ++rec;
        }
        B.b.push_back('A'); B.b.push_back('c');                        // unterminated last line
        rc |= run(B, true, 2, "edge shapes");
    }
    if (argc > 1 && !strcmp(argv[1], "check")) return rc;
    {
        Built B;
        B.b.reserve((size_t)3100 << 20);
        for (int c = 0; c < 24; ++c) add_record(B, "chr" + std::to_string(c + 1), (size_t)125000000, 60, 0);
        rc |= run(B, false, 5, "3 Gbp, 24 records, 60 columns");
    }
    return rc;
}
