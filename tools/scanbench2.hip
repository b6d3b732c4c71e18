// scanbench2.hip -- stand-alone check + timing harness for k_span_scan (fx_spanscan.hpp).
// Not part of the product.  (1) a 160 MB host-built buffer with every awkward shape
// (dense headers, blank lines, CRLF, '>' inside lines, a 300 KB line, no trailing newline)
// is summarised on the GPU and compared span by span with a scalar CPU reference;
// (2) a 3 GB 60-column buffer is timed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/scanbench2 tools/scanbench2.hip && /tmp/scanbench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include <algorithm>
#include "../pyfastx_amd/csrc/fx_spanscan.hpp"

using namespace fx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static uint32_t rng_state = 12345;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

static void build_host(std::vector<uint8_t> &b, size_t target, int nk) {
    const char B[8] = {'A', 'C', 'G', 'T', 'a', 'c', 'g', 'N'};
    int rec = 0;
    while (b.size() < target) {
        const int kind = nk == 9 ? rec % 9 : 1 + rec % 8 == 5 ? 1 : (1 + rec % 8 == 7 ? 3 : 1 + rec % 8);
        char hdr[128];
        int hl = snprintf(hdr, sizeof hdr, ">seq%d some description %u", rec, rnd());
        const bool crlf = kind == 4;
        b.insert(b.end(), hdr, hdr + hl);
        if (crlf) b.push_back('\r');
        b.push_back('\n');
        size_t len = kind == 0 ? (size_t)(20 << 20) + rnd() % 1000 : kind == 1 ? 50 + rnd() % 500 : kind == 2 ? 0 :
                     kind == 3 ? 5000 : kind == 5 ? 300000 : kind == 6 ? 700 : kind == 7 ? (size_t)(3 << 20) : 130;
        const int width = kind == 5 ? 400000 : kind == 6 ? 7 : kind == 7 ? 80 : 60;
        size_t col = 0;
        for (size_t i = 0; i < len; ++i) {
            uint8_t c = (uint8_t)B[rnd() & 7];
            if (kind == 3 && (i % 977) == 5) c = '>';         // '>' inside a line
            b.push_back(c);
            if (++col == (size_t)width) { if (crlf) b.push_back('\r'); b.push_back('\n'); col = 0;
                if (kind == 6 && (i % 91) == 0) b.push_back('\n'); }   // blank lines
        }
        if (col) { if (crlf) b.push_back('\r'); b.push_back('\n'); }
        if (kind == 8) { b.push_back('\n'); b.push_back('\n'); }
        ++rec;
    }
    b.push_back('A'); b.push_back('C');     // unterminated last line
}

struct RefSpan { uint32_t n_nl = 0, n_hdr = 0, first = SP_NONE, last = SP_NONE; std::map<uint32_t, uint32_t> d; };

struct RefHdr { int64_t pos, line; uint32_t gran; };

int check(const std::vector<uint8_t> &hb) {
    const int64_t n = (int64_t)hb.size();
    const int64_t ngran = n / GRAN + 1;
    uint8_t *d;
    CK(hipMalloc((void **)&d, (size_t)n + 64));
    CK(hipMemset(d, 0x41, (size_t)n + 64));
    CK(hipMemcpy(d, hb.data(), (size_t)n, hipMemcpyHostToDevice));
    GranPk *go;
    CK(hipMalloc((void **)&go, (size_t)ngran * sizeof(GranPk)));
    GranList gl;
    CK(hipMalloc((void **)&gl.g, (size_t)ngran * 4));
    CK(hipMalloc((void **)&gl.count, 4));
    CK(hipMemset(gl.count, 0, 4));
    const int64_t nchunks = (ngran + CHUNK_GRANS - 1) / CHUNK_GRANS;
    ChunkTot *ct; Totals *tot; int64_t *dpn, *dph, *dpv;
    CK(hipMalloc((void **)&ct, nchunks * sizeof(ChunkTot))); CK(hipMalloc((void **)&tot, sizeof(Totals)));
    CK(hipMemset(tot, 0, sizeof(Totals)));
    CK(hipMalloc((void **)&dpn, (ngran + 1) * 8)); CK(hipMalloc((void **)&dph, (ngran + 1) * 8)); CK(hipMalloc((void **)&dpv, (ngran + 1) * 8));
    hipLaunchKernelGGL(k_span_scan<0>, dim3((unsigned)(((n / GRAN + SCAN_GPW - 1) / SCAN_GPW * 64 + 511) / 512)), dim3(512), 0, 0, d, n, (int)'\n', 1, n / GRAN, go, gl);
    hipLaunchKernelGGL((k_gran_reduce<0, CHUNK_GRANS>), dim3((unsigned)nchunks), dim3(CHUNK_GRANS), 0, 0, d, n, (int)'\n', 1, gl, go, ngran, (int64_t)1000, ct);
    hipLaunchKernelGGL(k_gran_prefix<CHUNK_GRANS>, dim3((unsigned)nchunks), dim3(CHUNK_GRANS), 0, 0, go, ngran, (int64_t)1000, ct, tot, dpn, dph, dpv);
    CK(hipDeviceSynchronize());
    std::vector<GranPk> hp(ngran);
    CK(hipMemcpy(hp.data(), go, (size_t)ngran * sizeof(GranPk), hipMemcpyDeviceToHost));
    std::vector<GranOut> hg(ngran);
    for (int64_t i = 0; i < ngran; ++i) {
        const GranPk &q = hp[i]; GranOut &o = hg[i];
        o.n = q.nh & 0xFFFF; o.h = q.nh >> 16; o.first = o.n ? (q.fl & 0xFFFF) : SP_NONE; o.last = o.n ? (q.fl >> 16) : SP_NONE;
        o.v1 = q.d1 & 0xFFFF; o.c1 = q.d1 >> 16; o.v2 = q.d2 & 0xFFFF; o.c2 = (q.d2 >> 16) & 0x7FFF; o.ovf = q.d2 >> 31;
    }
    // reference
    std::vector<RefSpan> ref(ngran);
    std::vector<RefHdr> rh;
    const bool virt = hb[n - 1] != '\n';
    int64_t prev_nl = -1, line = 0;
    for (int64_t i = 0; i <= n; ++i) {
        const bool nl = i < n ? hb[i] == '\n' : virt;
        const int64_t s = i / GRAN;
        if (i < n && hb[i] == '>' && (i == 0 || hb[i - 1] == '\n')) { rh.push_back({i, line, (uint32_t)s}); ref[s].n_hdr++; }
        if (nl) {
            RefSpan &r = ref[s];
            const uint32_t lp = (uint32_t)(i - s * GRAN);
            if (r.n_nl && prev_nl >= s * (int64_t)GRAN) r.d[(uint32_t)(i - prev_nl)]++;
            if (!r.n_nl) r.first = lp;
            r.last = lp; r.n_nl++;
            prev_nl = i; ++line;
        }
    }
    int bad = 0;
    for (int64_t s = 0; s < ngran && bad < 10; ++s) {
        const RefSpan &r = ref[s];
        const GranOut &o = hg[s];
        bool ok = o.n == r.n_nl && o.h == r.n_hdr && o.first == r.first && o.last == r.last;
        if (r.d.size() > 2) ok &= o.ovf == 1;
        else {
            std::map<uint32_t, uint32_t> m;
            if (o.c1) m[o.v1] += o.c1;
            if (o.c2) m[o.v2] += o.c2;
            ok &= o.ovf == 0 && m == r.d;
        }
        if (!ok) { ++bad; printf("granule %lld mismatch: gpu n=%u h=%u f=%u l=%u v1=%u c1=%u v2=%u c2=%u ovf=%u | ref n=%u h=%u f=%u l=%u nd=%zu\n",
                          (long long)s, o.n, o.h, o.first, o.last, o.v1, o.c1, o.v2, o.c2, o.ovf, r.n_nl, r.n_hdr, r.first, r.last, r.d.size()); }
    }
    // device prefixes vs host prefixes
    std::vector<int64_t> pn(ngran + 1, 0), ph(ngran + 1, 0), pv(ngran + 1, -1);
    for (int64_t s = 0; s < ngran; ++s) { pn[s + 1] = pn[s] + hg[s].n; ph[s + 1] = ph[s] + hg[s].h;
        pv[s + 1] = hg[s].n ? 1000 + s * GRAN + hg[s].last : pv[s]; }
    std::vector<int64_t> gn(ngran + 1), gh(ngran + 1), gv(ngran + 1);
    CK(hipMemcpy(gn.data(), dpn, (ngran + 1) * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gh.data(), dph, (ngran + 1) * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gv.data(), dpv, (ngran + 1) * 8, hipMemcpyDeviceToHost));
    for (int64_t s = 0; s <= ngran && bad < 10; ++s)
        if (gn[s] != pn[s] || gh[s] != ph[s] || gv[s] != pv[s]) { ++bad; printf("prefix %lld: gpu %lld %lld %lld | ref %lld %lld %lld\n", (long long)s,
            (long long)gn[s], (long long)gh[s], (long long)gv[s], (long long)pn[s], (long long)ph[s], (long long)pv[s]); }
    uint32_t ngl = 0;
    CK(hipMemcpy(&ngl, gl.count, 4, hipMemcpyDeviceToHost));
    if ((size_t)ph[ngran] != rh.size()) { printf("header count %lld vs %zu\n", (long long)ph[ngran], rh.size()); ++bad; }
    printf("check: %lld bytes, %lld granules, %zu headers in %u granules, %s\n", (long long)n, (long long)ngran, rh.size(), ngl, bad ? "MISMATCH" : "all equal");
    for (void *q : {(void *)d, (void *)go, (void *)gl.g, (void *)gl.count, (void *)dpn, (void *)dph, (void *)dpv, (void *)ct, (void *)tot}) (void)hipFree(q);
    return bad;
}

__global__ void fill(uint8_t *d, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t h = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 13);
        h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
        const char b[8] = {'A', 'C', 'G', 'T', 'a', 'c', 'g', 't'};
        const int64_t r = i % (61ll * 250000);            // a 61-byte header line every 15.25 MB
        d[i] = (i % 61 == 60) ? '\n' : (r == 0 ? '>' : (uint8_t)b[h & 7]);
    }
}

int main() {
    std::vector<uint8_t> hb;
    build_host(hb, 160u << 20, 9);
    if (check(hb)) return 1;
    hb.clear();
    build_host(hb, 24u << 20, 1);           // only small records: headers in every granule
    hb.push_back('\n');                      // ... and a terminated last line
    if (check(hb)) return 1;
    const int64_t n = 3050ll << 20;
    uint8_t *d;
    CK(hipMalloc((void **)&d, n + (1 << 20)));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, d, n + (1 << 20));
    const int64_t ngran = n / GRAN + 1;
    GranPk *go;
    CK(hipMalloc((void **)&go, (size_t)ngran * sizeof(GranPk)));
    GranList gl;
    CK(hipMalloc((void **)&gl.g, (size_t)ngran * 4));
    CK(hipMalloc((void **)&gl.count, 4));
    CK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int rep = 0; rep < 2; ++rep) for (int blk : {256, 512, 1024}) for (int grid : {(int)(((ngran + SCAN_GPW - 1) / SCAN_GPW * 64 + blk - 1) / blk)}) {
    float tot = 0, best = 1e9f;
    const int R = 40;
    for (int r = 0; r < R + 2; ++r) {
        CK(hipMemsetAsync(gl.count, 0, 4, 0));
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k_span_scan<0>, dim3(grid), dim3(blk), 0, 0, d, n, (int)'\n', 1, n / GRAN, go, gl);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (r >= 2) { tot += ms; best = ms < best ? ms : best; }
    }
    uint32_t nh = 0;
    CK(hipMemcpy(&nh, gl.count, 4, hipMemcpyDeviceToHost));
    printf("k_span_scan block %4d grid %6d  %.2f GB  avg %.4f ms  best %.4f ms  -> %.0f GB/s (best %.0f)  header granules %u\n", blk, grid, n / 1e9, tot / R, best,
           n / (tot / R * 1e-3) / 1e9, n / (best * 1e-3) / 1e9, nh);
  }
    return 0;
}
