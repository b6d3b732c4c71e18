// fx_fxi.hpp -- bulk loader for the big table of a `.fxi` (SQLite3) index file.  Host code.
//
// The reference fills `read` / `seq` with one sqlite3_step per record inside a transaction
// (index.c:239-251, fastq.c:136-149): 0.8 M rows/s, the ceiling of a 10^8-read FASTQ index
//  once the scan itself takes milliseconds.  The rows come out of the GPU in rowid
// order, so the table b-tree can be written directly: leaf pages are filled left to right, the
// interior levels are a few pages on top, and nothing is ever rebalanced.  SQLite itself still
// creates the file, the schema and the small tables, and builds the UNIQUE INDEX afterwards
// (pyfastx_amd/fxi.py); this code only replaces the empty root page of the table by the loaded tree
// and appends the new pages (file format: https://www.sqlite.org/fileformat2.html, sections 1.3,
// 1.6, 2.1).  Row shape: (rowid alias stored as NULL, name TEXT, k INTEGER columns) -- both `read`
// (fastq.c:29-36) and `seq` (index.c:178-189) have it.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/statvfs.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace fxi {

// worker threads of the page formatters: a quarter of the host's hardware threads, 4 .. 16 (FX_FXI_THREADS overrides).
// More does not help: 20 M rows / 2 GB on tmpfs take 1.0 s with 16 threads, 2.2 s with 64, 2.6 s with 128 -- the
// writers then queue on the file's page cache.
static inline int max_threads() {
    static const int n = []() {
        if (const char *e = getenv("FX_FXI_THREADS")) { const int v = atoi(e); if (v > 0) return v > 256 ? 256 : v; }
        const unsigned hw = std::thread::hardware_concurrency();
        const int v = (int)(hw / 4);
        return v < 4 ? 4 : v > 16 ? 16 : v;
    }();
    return n;
}


enum { OK = 0, E_IO = -3, E_INVAL = -7, E_ROW = -6 };      // same numbering as fx_status

static inline int put_varint(uint8_t *p, uint64_t v) {     // SQLite varint: big-endian base 128, 9th byte holds 8 bits
    if (v <= 0x7F) { p[0] = (uint8_t)v; return 1; }
    if (v >> 56) {
        p[8] = (uint8_t)v; v >>= 8;
        for (int i = 7; i >= 0; --i) { p[i] = (uint8_t)((v & 0x7F) | 0x80); v >>= 7; }
        return 9;
    }
    uint8_t tmp[9];
    int n = 0;
    while (v) { tmp[n++] = (uint8_t)(v & 0x7F); v >>= 7; }
    for (int i = 0; i < n; ++i) p[i] = tmp[n - 1 - i] | (i < n - 1 ? 0x80 : 0);
    return n;
}
static inline int varint_len(uint64_t v) {
    int n = 1;
    while (v > 0x7F && n < 9) { v >>= 7; ++n; }
    return n;
}
// serial type and byte length of an INTEGER value, as SQLite (schema format 4) stores it
static inline int int_serial(int64_t v, int *nbytes) {
    if (v == 0) { *nbytes = 0; return 8; }
    if (v == 1) { *nbytes = 0; return 9; }
    const uint64_t u = v < 0 ? ~(uint64_t)v : (uint64_t)v;
    if (u <= 127) { *nbytes = 1; return 1; }
    if (u <= 32767) { *nbytes = 2; return 2; }
    if (u <= 8388607) { *nbytes = 3; return 3; }
    if (u <= 2147483647ull) { *nbytes = 4; return 4; }
    if (u <= 140737488355327ull) { *nbytes = 6; return 5; }
    *nbytes = 8; return 6;
}
static inline void put_be(uint8_t *p, uint64_t v, int n) { for (int i = n - 1; i >= 0; --i) { p[i] = (uint8_t)v; v >>= 8; } }

struct Rows {
    int64_t n;
    const uint8_t *names;             // null (with name_off null): the table has no TEXT column after the key
    const int64_t *name_off;          // n + 1 offsets into names
    int ncols;
    const int64_t *const *cols;
};

// size of the leaf cell of row i (payload-size varint + rowid varint + record); 0 when the row cannot be
// stored without an overflow page
static inline int cell_size(const Rows &r, int64_t i, int usable) {
    const bool text = r.name_off != nullptr;
    const int64_t L = text ? r.name_off[i + 1] - r.name_off[i] : 0;
    const int tlen = text ? varint_len((uint64_t)(13 + 2 * L)) : 0;
    int hdr = 1 + 1 + tlen + r.ncols, body = (int)L;
    for (int c = 0; c < r.ncols; ++c) { int nb; int_serial(r.cols[c][i], &nb); body += nb; }
    const int64_t payload = hdr + body;
    if (hdr > 127 || payload > usable - 35 || payload > 8000) return 0;      // 8000: format_leaf's cell buffer
    return varint_len((uint64_t)payload) + varint_len((uint64_t)(i + 1)) + (int)payload;
}
static inline int put_cell(uint8_t *p, const Rows &r, int64_t i) {
    const bool text = r.name_off != nullptr;
    const int64_t L = text ? r.name_off[i + 1] - r.name_off[i] : 0;
    const int tlen = text ? varint_len((uint64_t)(13 + 2 * L)) : 0;
    const int hdr = 1 + 1 + tlen + r.ncols;
    int body = (int)L, nb[16], st[16];
    for (int c = 0; c < r.ncols; ++c) { st[c] = int_serial(r.cols[c][i], &nb[c]); body += nb[c]; }
    uint8_t *q = p;
    q += put_varint(q, (uint64_t)(hdr + body));
    q += put_varint(q, (uint64_t)(i + 1));
    *q++ = (uint8_t)hdr;
    *q++ = 0;                                              // INTEGER PRIMARY KEY column: NULL, the rowid is the value
    if (text) q += put_varint(q, (uint64_t)(13 + 2 * L));
    for (int c = 0; c < r.ncols; ++c) *q++ = (uint8_t)st[c];
    if (L) { memcpy(q, r.names + r.name_off[i], (size_t)L); q += L; }
    for (int c = 0; c < r.ncols; ++c) { put_be(q, (uint64_t)r.cols[c][i], nb[c]); q += nb[c]; }
    return (int)(q - p);
}

// one table-leaf page holding rows [a, b)
static void format_leaf(uint8_t *pg, int pagesize, int usable, const Rows &r, int64_t a, int64_t b) {
    memset(pg, 0, (size_t)pagesize);
    int top = usable;
    for (int64_t i = a; i < b; ++i) {
        uint8_t tmp[8192];
        const int len = put_cell(tmp, r, i);
        top -= len;
        memcpy(pg + top, tmp, (size_t)len);
        put_be(pg + 8 + 2 * (i - a), (uint64_t)top, 2);
    }
    pg[0] = 0x0D;
    put_be(pg + 3, (uint64_t)(b - a), 2);
    put_be(pg + 5, (uint64_t)(top == 65536 ? 0 : top), 2);
}
// one table-interior page: children [a, b) of `kids` (page numbers) with `maxkey` (largest rowid below each)
static void format_interior(uint8_t *pg, int pagesize, int usable, const std::vector<uint32_t> &kids,
                            const std::vector<int64_t> &maxkey, size_t a, size_t b, int hdr_off) {
    memset(pg + hdr_off, 0, (size_t)(pagesize - hdr_off));
    int top = usable;
    for (size_t i = a; i + 1 < b; ++i) {                   // the last child is the right-most pointer
        uint8_t tmp[16];
        put_be(tmp, kids[i], 4);
        const int len = 4 + put_varint(tmp + 4, (uint64_t)maxkey[i]);
        top -= len;
        memcpy(pg + top, tmp, (size_t)len);
        put_be(pg + hdr_off + 12 + 2 * (i - a), (uint64_t)top, 2);
    }
    pg[hdr_off] = 0x05;
    put_be(pg + hdr_off + 3, (uint64_t)(b - a - 1), 2);
    put_be(pg + hdr_off + 5, (uint64_t)(top == 65536 ? 0 : top), 2);
    put_be(pg + hdr_off + 8, kids[b - 1], 4);
}

static bool pwrite_all(int fd, const uint8_t *p, size_t n, off_t off) {
    while (n) { const ssize_t w = pwrite(fd, p, n, off); if (w <= 0) return false; p += w; n -= (size_t)w; off += w; }
    return true;
}

// The file grown to its final size and mapped: the writer threads format their pages in place.  pwrite() takes the
// inode lock for every call, so sixteen threads writing one file queue up behind each other; stores into a shared
// mapping do not.  (No mapping -- a file system that refuses it -- and the pages go through pwrite as before.)
// The file mapped for the writers -- in SEPARATE mappings of 8 MiB, a guard between them so that the kernel keeps them apart.
// Every store into a page of a mapping that has no page-table entry yet is a write fault (tmpfs here has no huge pages: one
// fault per 4 KiB), a fault takes the lock of ITS mapping, and with one mapping for the whole file that lock is one cache line
// for all copy threads.  tools/filewrite_probe2.c (profiles/r06_filewrite_populate.txt): 16 threads store 10 GiB into ONE
// mapping in 0.34 s (the copy alone, entries made beforehand: 0.05 s), into mappings of their own in 0.17 s.  In the product
// the copy lanes also wait for their device-to-host pieces, so 16 of them gain little (4.3 GB: 105 against 105-129 ms) -- but
// with one mapping more lanes made the copy-out SLOWER (24 / 32 lanes: 146 / 157 ms), with separate ones they wait for the
// device instead (99 / 110 ms).  at(off): where byte `off` of the file is; a span is cut at the mappings' borders (put()).
struct FileMap {
    static constexpr size_t MIN_CHUNK = 8u << 20, GUARD = 65536;         // (a page size of up to 64 KiB between the mappings)
    uint8_t *p = nullptr;
    size_t len = 0, area = 0;                              // bytes of the file covered; bytes of address space taken
    size_t chunk = MIN_CHUNK;                              // bytes per mapping: 8 MiB, more for files of more than 16 GiB (at most ~2048 mappings:
                                                           // a process may hold 65530 of them by default, and every guard is one more)
    bool one = false;                                      // ONE mapping (FX_FXI_ONE_MAPPING=1: the form before, for comparison)
    size_t stride() const { return chunk + GUARD; }
    uint8_t *at(size_t off) const { return one ? p + off : p + (off / chunk) * stride() + off % chunk; }
    void put(size_t off, const uint8_t *src, size_t n) const {
        while (n) {
            const size_t m = one ? n : std::min(n, chunk - off % chunk);
            memcpy(at(off), src, m);
            off += m; src += m; n -= m;
        }
    }
    // the first `newlen` bytes of a file that has them
    bool map_existing(int fd, size_t newlen) {
        close();
        if (getenv("FX_FXI_NO_MMAP") || newlen == 0) return false;
        static const bool single = [] { const char *e = getenv("FX_FXI_ONE_MAPPING"); return e && atoi(e) != 0; }();
        if (single) {
            void *m = mmap(nullptr, newlen, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (m == MAP_FAILED) return false;
            p = (uint8_t *)m; len = newlen; area = newlen; one = true;
            return true;
        }
        chunk = MIN_CHUNK;
        while ((newlen + chunk - 1) / chunk > 2048) chunk <<= 1;
        const size_t nch = (newlen + chunk - 1) / chunk;
        void *r = mmap(nullptr, nch * stride(), PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (r == MAP_FAILED) return false;
        for (size_t k = 0; k < nch; ++k) {
            const size_t off = k * chunk, n = std::min(chunk, newlen - off);
            if (mmap((uint8_t *)r + k * stride(), n, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, fd, (off_t)off) == MAP_FAILED) { munmap(r, nch * stride()); return false; }
        }
        p = (uint8_t *)r; len = newlen; area = nch * stride(); one = false;
        return true;
    }
    bool open(int fd, size_t newlen) {
        if (getenv("FX_FXI_NO_MMAP")) return false;
        // a store into a mapping of a full file system is a SIGBUS, a pwrite an error return: map only when the space is there
        struct stat st;
        struct statvfs vfs;
        if (fstat(fd, &st) != 0 || fstatvfs(fd, &vfs) != 0) return false;
        const uint64_t grow = newlen > (size_t)st.st_size ? newlen - (size_t)st.st_size : 0;
        if ((uint64_t)vfs.f_bavail * (uint64_t)vfs.f_frsize < grow + (64u << 20)) return false;
        if (ftruncate(fd, (off_t)newlen) != 0) return false;
        return map_existing(fd, newlen);
    }
    void close() { if (p) munmap(p, area); p = nullptr; len = area = 0; }
    ~FileMap() { close(); }
};
// one page into the mapping or through pwrite
static bool put_page(int fd, const FileMap &m, const uint8_t *pg, int pagesize, uint32_t pageno) {
    const size_t off = (size_t)(pageno - 1) * (size_t)pagesize;
    if (m.p) { memcpy(m.at(off), pg, (size_t)pagesize); return true; }
    return pwrite_all(fd, pg, (size_t)pagesize, (off_t)off);
}

// Load `rows` into the (empty) table whose b-tree root is page `rootpage` of the database file `path`.
// The file must be closed by every SQLite connection.  Returns OK, E_ROW when a row does not fit a page
// (the caller uses the INSERT path), E_IO / E_INVAL otherwise.
// New pages are appended to the file in order; the page that holds byte 2^30 (SQLite's "pending byte" page, used
// for file locking) must never carry data and is stepped over.
struct PageSeq {
    uint32_t base, lock;                                    // first new page; the page to skip
    PageSeq(uint32_t first, int pagesize) : base(first), lock((uint32_t)(0x40000000u / (uint32_t)pagesize) + 1) {}
    uint32_t at(uint64_t k) const {
        const uint64_t p = (uint64_t)base + k;
        return (uint32_t)((base <= lock && p >= lock) ? p + 1 : p);
    }
};


// ---- the interior levels of a table b-tree over leaves that are (being) written elsewhere: leaf k sits on page
// seq.at(k) and holds rows [leaf_first[k], leaf_first[k + 1]).  A cell is 4 + varint(rowid) <= 13 bytes, + 2 for its
// pointer.  Children are spread evenly over the pages of a level, so no page ends up with a single child; the levels
// follow the leaves in the page sequence, the top level lives in the root page.  (The host formatter below and the
// device formatter, fx_fxi_dev.hpp, both end here.)
static inline size_t table_fan(int usable) { return (size_t)((usable - 12) / 15) + 1; }             // children per interior page
static inline uint64_t table_new_pages(size_t nleaf, size_t fan) {
    uint64_t total = nleaf;
    for (size_t K = nleaf; K > fan; K = (K + fan - 1) / fan) total += (K + fan - 1) / fan;
    return total;
}
static bool table_interior(int fd, const FileMap &map, int pagesize, int usable, uint32_t rootpage, const PageSeq &seq,
                           const int64_t *leaf_first, size_t nleaf, uint64_t total) {
    std::vector<uint32_t> kids(nleaf);
    std::vector<int64_t> maxkey(nleaf);
    for (size_t k = 0; k < nleaf; ++k) { kids[k] = seq.at(k); maxkey[k] = leaf_first[k + 1]; }   // rowid = row + 1
    uint64_t next_k = nleaf;
    const size_t fan = table_fan(usable);
    std::vector<uint8_t> page((size_t)pagesize);
    bool ok = true;
    while (ok && kids.size() > fan) {
        const size_t K = kids.size(), groups = (K + fan - 1) / fan;
        std::vector<uint32_t> up(groups);
        std::vector<int64_t> upkey(groups);
        for (size_t g = 0; g < groups && ok; ++g) {
            const size_t a = K * g / groups, b = K * (g + 1) / groups;
            format_interior(page.data(), pagesize, usable, kids, maxkey, a, b, 0);
            up[g] = seq.at(next_k++);
            ok = put_page(fd, map, page.data(), pagesize, up[g]);
            upkey[g] = maxkey[b - 1];
        }
        kids.swap(up); maxkey.swap(upkey);
    }
    if (ok) {
        format_interior(page.data(), pagesize, usable, kids, maxkey, 0, kids.size(), 0);
        ok = put_page(fd, map, page.data(), pagesize, rootpage);
    }
    return ok && next_k == total;                            // the page count the file was sized for
}

static int bulk_load_table(const char *path, uint32_t rootpage, const Rows &r) {
    if (r.ncols < 0 || r.ncols > 16 || r.n < 0 || rootpage < 2) return E_INVAL;
    const int fd = open(path, O_RDWR);
    if (fd < 0) return E_IO;
    uint8_t hdr[100];
    if (pread(fd, hdr, 100, 0) != 100 || memcmp(hdr, "SQLite format 3", 16) != 0) { close(fd); return E_INVAL; }
    int pagesize = (hdr[16] << 8) | hdr[17];
    if (pagesize == 1) pagesize = 65536;
    const int usable = pagesize - hdr[20];
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return E_IO; }
    uint32_t npages = (uint32_t)(st.st_size / pagesize);
    if (rootpage > npages || hdr[18] > 1 || hdr[19] > 1) { close(fd); return E_INVAL; }     // journal mode must be the legacy one
    if (hdr[52] | hdr[53] | hdr[54] | hdr[55]) { close(fd); return E_INVAL; }                 // auto-vacuum files interleave pointer-map pages

    // ---- cell sizes (parallel), then leaves by greedy fill (sequential, cheap)
    const int T = (int)std::min<int64_t>(max_threads(), std::max<int64_t>(1, r.n / 65536));
    std::vector<uint16_t> csz((size_t)r.n);
    std::atomic<int> bad(0);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t]() {
                for (int64_t i = r.n * t / T; i < r.n * (t + 1) / T; ++i) {
                    const int c = cell_size(r, i, usable);
                    if (!c || c > 65535) { bad.store(1); return; }
                    csz[(size_t)i] = (uint16_t)c;
                }
            });
        for (auto &x : th) x.join();
    }
    if (bad.load()) { close(fd); return E_ROW; }
    std::vector<int64_t> leaf_first;                       // first row of every leaf, + n at the end
    {
        int freeb = 0;
        for (int64_t i = 0; i < r.n; ++i) {
            const int need = csz[(size_t)i] + 2;
            if (leaf_first.empty() || need > freeb) { leaf_first.push_back(i); freeb = usable - 8; }
            freeb -= need;
        }
        if (leaf_first.empty()) leaf_first.push_back(0);
        leaf_first.push_back(r.n);
    }
    const size_t nleaf = leaf_first.size() - 1;
    std::vector<uint8_t> page((size_t)pagesize);
    bool ok = true;
    if (nleaf == 1) {                                      // everything fits the root page itself
        format_leaf(page.data(), pagesize, usable, r, 0, r.n);
        ok = pwrite_all(fd, page.data(), (size_t)pagesize, (off_t)(rootpage - 1) * pagesize);
    } else {
        // ---- page numbers: leaves first, then the interior levels; the top level lives in the root page
        const PageSeq seq(npages + 1, pagesize);
        const uint64_t total = table_new_pages(nleaf, table_fan(usable));          // all new pages: the leaves + every interior level but the top
        FileMap map;
        map.open(fd, (size_t)seq.at(total - 1) * (size_t)pagesize);
        // leaves, in parallel, 256 pages per write
        {
            std::atomic<size_t> cursor(0);
            std::atomic<int> err(0);
            std::vector<std::thread> th;
            const int TW = (int)std::min<size_t>((size_t)max_threads(), std::max<size_t>(1, nleaf / 512));
            for (int t = 0; t < TW; ++t)
                th.emplace_back([&]() {
                    std::vector<uint8_t> buf(map.p ? 0 : (size_t)pagesize * 256);
                    for (;;) {
                        const size_t a = cursor.fetch_add(256);
                        if (a >= nleaf || err.load()) return;
                        const size_t b = std::min(nleaf, a + 256);
                        if (map.p) {                         // in place
                            for (size_t k = a; k < b; ++k)
                                format_leaf(map.at((size_t)(seq.at(k) - 1) * (size_t)pagesize), pagesize, usable, r, leaf_first[k], leaf_first[k + 1]);
                            continue;
                        }
                        for (size_t k = a; k < b; ++k)
                            format_leaf(buf.data() + (k - a) * (size_t)pagesize, pagesize, usable, r, leaf_first[k], leaf_first[k + 1]);
                        size_t cut = a + 1;                  // pages a .. cut-1 are adjacent in the file (the skipped page splits a batch)
                        while (cut < b && seq.at(cut) == seq.at(cut - 1) + 1) ++cut;
                        if (!pwrite_all(fd, buf.data(), (cut - a) * (size_t)pagesize, (off_t)(seq.at(a) - 1) * pagesize)) err.store(1);
                        if (cut < b && !pwrite_all(fd, buf.data() + (cut - a) * (size_t)pagesize, (b - cut) * (size_t)pagesize,
                                                   (off_t)(seq.at(cut) - 1) * pagesize)) err.store(1);
                    }
                });
            for (auto &x : th) x.join();
            ok = !err.load();
        }
        if (ok) ok = table_interior(fd, map, pagesize, usable, rootpage, seq, leaf_first.data(), nleaf, total);
        npages = seq.at(total - 1);
        map.close();
    }
    // ---- file header: size in pages, change counter, "version valid for"
    if (ok) {
        uint32_t change = ((uint32_t)hdr[24] << 24) | (hdr[25] << 16) | (hdr[26] << 8) | hdr[27];
        ++change;
        put_be(hdr + 24, change, 4);
        put_be(hdr + 28, npages, 4);
        put_be(hdr + 92, change, 4);
        ok = pwrite_all(fd, hdr, 100, 0);
    }
    // failure: the header still says the old page count -- give the pages appended so far back instead of leaving
    // them orphaned behind the database (the root page may have been rewritten: the caller discards or refills the table)
    if (!ok) (void)!ftruncate(fd, st.st_size);
    close(fd);
    return ok ? OK : E_IO;
}

// ------------------------------------------------------------------ index b-tree
// UNIQUE INDEX on the name column, loaded bottom-up from the SORTED order of the names (memcmp order, shorter first
// on a tie: SQLite's BINARY collation) -- the permutation fx_names_sort produces on the GPU.  An index b-tree keeps every
// (name, rowid) entry exactly once: leaves are filled left to right, the entry that follows a full leaf moves up
// as the divider of its parent, and the same happens between the pages of every upper level.
struct Entries {
    int64_t n;
    const uint8_t *names;             // TEXT key: packed in ROW order (the same buffer the table loader got) ...
    const int64_t *name_off;          // ... n + 1 offsets
    const int64_t *ikey;              // or INTEGER key (names / name_off null): ikey[row]
    const int64_t *order;             // order[i] = 0-based row of the i-th smallest key; its rowid is order[i] + 1
    const int64_t *rowid = nullptr;   // given (TEXT keys only): names are packed in ENTRY order and entry i carries rowid[i] (order unused)
};
static inline int entry_payload(const Entries &e, int64_t i) {
    const int64_t r = e.rowid ? i : e.order[i];
    const int64_t rid = e.rowid ? e.rowid[i] : r + 1;
    int nb, kb;
    int_serial(rid, &nb);
    if (e.ikey) { int_serial(e.ikey[r], &kb); return 1 + 1 + 1 + kb + nb; }
    const int64_t L = e.name_off[r + 1] - e.name_off[r];
    return 1 + varint_len((uint64_t)(13 + 2 * L)) + 1 + (int)L + nb;
}
static inline int put_entry(uint8_t *p, const Entries &e, int64_t i) {      // varint(payload) + record(key, rowid)
    const int64_t r = e.rowid ? i : e.order[i];
    const int64_t rid = e.rowid ? e.rowid[i] : r + 1;
    int nb;
    const int st = int_serial(rid, &nb);
    uint8_t *q = p;
    if (e.ikey) {
        int kb;
        const int kst = int_serial(e.ikey[r], &kb);
        q += put_varint(q, (uint64_t)(3 + kb + nb));
        *q++ = 3; *q++ = (uint8_t)kst; *q++ = (uint8_t)st;
        put_be(q, (uint64_t)e.ikey[r], kb); q += kb;
    } else {
        const int64_t L = e.name_off[r + 1] - e.name_off[r];
        const int tlen = varint_len((uint64_t)(13 + 2 * L));
        q += put_varint(q, (uint64_t)(1 + tlen + 1 + L + nb));
        *q++ = (uint8_t)(1 + tlen + 1);
        q += put_varint(q, (uint64_t)(13 + 2 * L));
        *q++ = (uint8_t)st;
        if (L) { memcpy(q, e.names + e.name_off[r], (size_t)L); q += L; }
    }
    put_be(q, (uint64_t)rid, nb); q += nb;
    return (int)(q - p);
}

// one level of the tree under construction: page k holds items [first[k], first[k+1]) of the level's item list;
// item j of level 0 is entry j, item j of level l > 0 is the divider entry up[l-1][j] with left child kid[l-1][j]
struct Level { std::vector<int64_t> first; std::vector<int64_t> divider; };   // divider[k]: item promoted after page k

// greedy fill of the items [0, m) with sizes sz(j) (cell bytes + 2) into pages of `room` bytes; the item after a
// full page is promoted.  Returns page boundaries + promoted items.
template <class SizeFn>
static Level fill_level(int64_t m, int room, SizeFn sz) {
    Level lv;
    int64_t j = 0;
    while (j < m) {
        lv.first.push_back(j);
        int freeb = room;
        int64_t cnt = 0;
        while (j < m && sz(j) <= freeb) { freeb -= sz(j); ++j; ++cnt; }
        if (cnt == 0) { lv.first.clear(); return lv; }      // an item larger than a page: caller treats as E_ROW
        if (j < m) {
            if (j == m - 1) { --j; }                         // never promote the very last item: give it a page of its own ...
            lv.divider.push_back(j);                         // ... by promoting the last item of this page instead
            ++j;
        }
    }
    lv.first.push_back(m);
    return lv;
}

static int bulk_load_index(const char *path, uint32_t rootpage, const Entries &e) {
    if (e.n < 0 || rootpage < 2) return E_INVAL;
    const int fd = open(path, O_RDWR);
    if (fd < 0) return E_IO;
    uint8_t hdr[100];
    if (pread(fd, hdr, 100, 0) != 100 || memcmp(hdr, "SQLite format 3", 16) != 0) { close(fd); return E_INVAL; }
    int pagesize = (hdr[16] << 8) | hdr[17];
    if (pagesize == 1) pagesize = 65536;
    const int usable = pagesize - hdr[20];
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return E_IO; }
    uint32_t npages = (uint32_t)(st.st_size / pagesize);
    if (rootpage > npages || hdr[18] > 1 || hdr[19] > 1 || (hdr[52] | hdr[53] | hdr[54] | hdr[55])) { close(fd); return E_INVAL; }
    const int max_local = ((usable - 12) * 64 / 255) - 23;  // larger index payloads would spill to overflow pages

    // payload sizes (parallel)
    std::vector<uint16_t> psz((size_t)e.n);
    {
        const int T = (int)std::min<int64_t>(max_threads(), std::max<int64_t>(1, e.n / 65536));
        std::atomic<int> bad(0);
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t]() {
                for (int64_t i = e.n * t / T; i < e.n * (t + 1) / T; ++i) {
                    const int p = entry_payload(e, i);
                    if (p > max_local || p > 4000) { bad.store(1); return; }
                    psz[(size_t)i] = (uint16_t)p;
                }
            });
        for (auto &x : th) x.join();
        if (bad.load()) { close(fd); return E_ROW; }
    }
    auto leaf_cell = [&](int64_t i) { return (int)psz[(size_t)i] + varint_len(psz[(size_t)i]) + 2; };            // + cell pointer
    auto int_cell = [&](int64_t i) { return leaf_cell(i) + 4; };

    // ---- shape of the tree, level by level (items of level l+1 = dividers of level l)
    std::vector<Level> levels;
    std::vector<std::vector<int64_t>> items;                 // items[l][j] = entry index of item j (level 0: identity, not stored)
    levels.push_back(fill_level(e.n, usable - 8, [&](int64_t j) { return leaf_cell(j); }));
    if (e.n && levels[0].first.empty()) { close(fd); return E_ROW; }
    if (e.n == 0) { levels[0].first = {0, 0}; }
    items.push_back({});
    while (levels.back().first.size() - 1 > 1) {             // more than one page on this level: build the level above
        const Level &lo = levels.back();
        std::vector<int64_t> up;                             // entry indices of the dividers
        const std::vector<int64_t> &src = items.back();
        for (int64_t d : lo.divider) up.push_back(levels.size() == 1 ? d : src[(size_t)d]);
        items.push_back(up);
        const std::vector<int64_t> &it = items.back();
        levels.push_back(fill_level((int64_t)it.size(), usable - 12, [&](int64_t j) { return int_cell(it[(size_t)j]); }));
        if (levels.back().first.empty()) { close(fd); return E_ROW; }
    }
    // ---- page numbers: level 0 pages first, ...; the single page of the top level is the root page
    const size_t nlev = levels.size();
    std::vector<std::vector<uint32_t>> pageno(nlev);
    const PageSeq seq(npages + 1, pagesize);
    uint64_t next_k = 0;
    for (size_t l = 0; l < nlev; ++l) {
        const size_t np = levels[l].first.size() - 1;
        pageno[l].resize(np);
        for (size_t k = 0; k < np; ++k) pageno[l][k] = (l + 1 == nlev) ? rootpage : seq.at(next_k++);
    }
    bool ok = true;
    FileMap map;
    if (next_k) map.open(fd, (size_t)seq.at(next_k - 1) * (size_t)pagesize);
    // ---- leaves (parallel)
    {
        const Level &lv = levels[0];
        const size_t np = lv.first.size() - 1;
        std::atomic<size_t> cursor(0);
        std::atomic<int> err(0);
        const int TW = (int)std::min<size_t>((size_t)max_threads(), std::max<size_t>(1, np / 512));
        std::vector<std::thread> th;
        for (int t = 0; t < TW; ++t)
            th.emplace_back([&]() {
                std::vector<uint8_t> own((size_t)pagesize);
                uint8_t tmp[8192];
                for (;;) {
                    const size_t k = cursor.fetch_add(1);
                    if (k >= np || err.load()) return;
                    uint8_t *const pg = map.p ? map.at((size_t)(pageno[0][k] - 1) * (size_t)pagesize) : own.data();   // in place when mapped
                    memset(pg, 0, (size_t)pagesize);
                    // items of page k: entries first[k] .. end, where the promoted one (if any) is excluded
                    int64_t a = lv.first[k], b = lv.first[k + 1];
                    if (k + 1 < np) --b;                     // the last item before the next page is this page's divider
                    int top = usable, c = 0;
                    for (int64_t i = a; i < b; ++i, ++c) {
                        const int len = put_entry(tmp, e, i);
                        top -= len;
                        memcpy(pg + top, tmp, (size_t)len);
                        put_be(pg + 8 + 2 * c, (uint64_t)top, 2);
                    }
                    pg[0] = 0x0A;
                    put_be(pg + 3, (uint64_t)c, 2);
                    put_be(pg + 5, (uint64_t)(top == 65536 ? 0 : top), 2);
                    if (!map.p && !pwrite_all(fd, pg, (size_t)pagesize, (off_t)(pageno[0][k] - 1) * pagesize)) err.store(1);
                }
            });
        for (auto &x : th) x.join();
        ok = !err.load();
    }
    // ---- interior levels
    std::vector<uint8_t> pg((size_t)pagesize);
    uint8_t tmp[8192];
    for (size_t l = 1; l < nlev && ok; ++l) {
        const Level &lv = levels[l];
        const std::vector<int64_t> &it = items[l];           // item j: divider entry it[j]; left child = page j of level l-1
        const size_t np = lv.first.size() - 1;
        for (size_t k = 0; k < np && ok; ++k) {
            memset(pg.data(), 0, (size_t)pagesize);
            int64_t a = lv.first[k], b = lv.first[k + 1];
            if (k + 1 < np) --b;                             // that item is promoted further up
            int top = usable, c = 0;
            for (int64_t j = a; j < b; ++j, ++c) {
                put_be(tmp, pageno[l - 1][(size_t)j], 4);    // left child: the page that ends just before divider j
                const int len = 4 + put_entry(tmp + 4, e, it[(size_t)j]);
                top -= len;
                memcpy(pg.data() + top, tmp, (size_t)len);
                put_be(pg.data() + 12 + 2 * c, (uint64_t)top, 2);
            }
            pg[0] = 0x02;
            put_be(pg.data() + 3, (uint64_t)c, 2);
            put_be(pg.data() + 5, (uint64_t)(top == 65536 ? 0 : top), 2);
            put_be(pg.data() + 8, pageno[l - 1][(size_t)b], 4);   // right-most child: the page after the last divider kept here
            ok = put_page(fd, map, pg.data(), pagesize, pageno[l][k]);
        }
    }
    if (ok && nlev == 1 && levels[0].first.size() - 1 == 1) { /* the single leaf was written to the root page above */ }
    if (ok) {
        uint32_t change = ((uint32_t)hdr[24] << 24) | (hdr[25] << 16) | (hdr[26] << 8) | hdr[27];
        ++change;
        put_be(hdr + 24, change, 4);
        put_be(hdr + 28, next_k ? seq.at(next_k - 1) : npages, 4);
        put_be(hdr + 92, change, 4);
        map.close();
        ok = pwrite_all(fd, hdr, 100, 0);
    }
    map.close();
    // failure: the header still says the old page count -- give the pages appended so far back instead of leaving
    // them orphaned behind the database (the root page may have been rewritten: the caller discards or refills the table)
    if (!ok) (void)!ftruncate(fd, st.st_size);
    close(fd);
    return ok ? OK : E_IO;
}

// ------------------------------------------------------------------ for the device formatter (fx_fxi_dev.hpp)
// An index-file database opened for appending pages: header checked, size known.
struct DbFile {
    int fd = -1;
    uint8_t hdr[100];
    int pagesize = 0, usable = 0;
    uint32_t npages = 0;
    off_t size0 = 0;
    int open_rw(const char *path, uint32_t rootpage) {
        fd = open(path, O_RDWR);
        if (fd < 0) return E_IO;
        if (pread(fd, hdr, 100, 0) != 100 || memcmp(hdr, "SQLite format 3", 16) != 0) return E_INVAL;
        pagesize = (hdr[16] << 8) | hdr[17];
        if (pagesize == 1) pagesize = 65536;
        usable = pagesize - hdr[20];
        struct stat st;
        if (fstat(fd, &st) != 0) return E_IO;
        size0 = st.st_size;
        npages = (uint32_t)(st.st_size / pagesize);
        // a file that was made longer than its database (room set aside for the pages to come): the header's own count,
        // valid when "version valid for" equals the change counter (fileformat2 1.3.8), says where the database ends
        const uint32_t in_hdr = ((uint32_t)hdr[28] << 24) | (hdr[29] << 16) | (hdr[30] << 8) | hdr[31];
        if (in_hdr && in_hdr < npages && memcmp(hdr + 24, hdr + 92, 4) == 0) npages = in_hdr;
        if (rootpage < 2 || rootpage > npages || hdr[18] > 1 || hdr[19] > 1 || (hdr[52] | hdr[53] | hdr[54] | hdr[55])) return E_INVAL;
        return OK;
    }
    bool finish(uint32_t new_npages) {                       // file header: size in pages, change counter, "version valid for"
        uint32_t change = ((uint32_t)hdr[24] << 24) | (hdr[25] << 16) | (hdr[26] << 8) | hdr[27];
        ++change;
        put_be(hdr + 24, change, 4);
        put_be(hdr + 28, new_npages, 4);
        put_be(hdr + 92, change, 4);
        return pwrite_all(fd, hdr, 100, 0);
    }
    void give_back() { if (fd >= 0) (void)!ftruncate(fd, size0); }     // failure: no orphaned pages behind the database
    ~DbFile() { if (fd >= 0) close(fd); }
};

// The levels of an index b-tree ABOVE leaves that are written elsewhere: leaf k sits on page seq.at(k); divider d -- the
// entry between leaf d and leaf d + 1 -- is entry d of `dv` (names packed in that order, dv.rowid set).  plan() works out
// the shape (so that the file can be sized before anything is written), write() formats the pages: level 1 follows the
// leaves in the page sequence, ..., the single page of the top level is the root page.
struct IndexUpper {
    std::vector<Level> levels;                               // levels[0] = the level above the leaves
    std::vector<std::vector<int64_t>> items;                 // items[l][j] = divider (entry of dv) that is item j of that level
    uint64_t pages = 0;                                      // new pages: every level here but the top one
    bool plan(size_t nleaf, const Entries &dv, int usable) {
        levels.clear(); items.clear(); pages = 0;
        if (nleaf < 2) return true;
        auto cell = [&](int64_t i) { const int p = entry_payload(dv, i); return p + varint_len((uint64_t)p) + 2 + 4; };
        std::vector<int64_t> it((size_t)nleaf - 1);
        for (size_t j = 0; j < it.size(); ++j) it[j] = (int64_t)j;
        for (;;) {
            items.push_back(it);
            const std::vector<int64_t> &cur = items.back();
            levels.push_back(fill_level((int64_t)cur.size(), usable - 12, [&](int64_t j) { return cell(cur[(size_t)j]); }));
            const Level &lv = levels.back();
            if (lv.first.empty()) return false;
            if (lv.first.size() - 1 <= 1) break;
            pages += lv.first.size() - 1;
            std::vector<int64_t> up;
            for (int64_t d : lv.divider) up.push_back(cur[(size_t)d]);
            it.swap(up);
        }
        return true;
    }
    bool write(int fd, const FileMap &map, int pagesize, int usable, uint32_t rootpage, const PageSeq &seq, size_t nleaf, const Entries &dv) const {
        std::vector<uint8_t> pg((size_t)pagesize);
        std::vector<uint8_t> tmp(8192);
        uint64_t next_k = nleaf;
        std::vector<uint32_t> below(nleaf), here;
        for (size_t k = 0; k < nleaf; ++k) below[k] = seq.at(k);
        for (size_t l = 0; l < levels.size(); ++l) {
            const Level &lv = levels[l];
            const std::vector<int64_t> &it = items[l];       // item j: divider it[j]; left child = page j of the level below
            const size_t np = lv.first.size() - 1;
            here.assign(np, 0);
            for (size_t k = 0; k < np; ++k) here[k] = (l + 1 == levels.size()) ? rootpage : seq.at(next_k++);
            for (size_t k = 0; k < np; ++k) {
                memset(pg.data(), 0, (size_t)pagesize);
                int64_t a = lv.first[k], b = lv.first[k + 1];
                if (k + 1 < np) --b;                         // that item is promoted further up
                int top = usable, c = 0;
                for (int64_t j = a; j < b; ++j, ++c) {
                    put_be(tmp.data(), below[(size_t)j], 4);
                    const int len = 4 + put_entry(tmp.data() + 4, dv, it[(size_t)j]);
                    top -= len;
                    memcpy(pg.data() + top, tmp.data(), (size_t)len);
                    put_be(pg.data() + 12 + 2 * c, (uint64_t)top, 2);
                }
                pg[0] = 0x02;
                put_be(pg.data() + 3, (uint64_t)c, 2);
                put_be(pg.data() + 5, (uint64_t)(top == 65536 ? 0 : top), 2);
                put_be(pg.data() + 8, below[(size_t)b], 4);  // right-most child: the page after the last divider kept here
                if (!put_page(fd, map, pg.data(), pagesize, here[k])) return false;
            }
            below.swap(here);
        }
        return next_k == nleaf + pages;
    }
};

}  // namespace fxi
