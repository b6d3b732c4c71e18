"""fxi::FileMap (csrc/fx_fxi.hpp, host code): the index file mapped in separate pieces.  A small C++ program built with g++
(no device) stores a pattern through put() / at() across the borders of the pieces, takes the mappings down, and the file is
read back here byte for byte -- both forms (separate mappings; FX_FXI_ONE_MAPPING=1) and a length that is no multiple of a page."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include "fx_fxi.hpp"
#include <cstdio>
int main(int argc, char **argv) {
    const char *path = argv[1];
    const size_t len = (size_t)atoll(argv[2]);
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return 2;
    fxi::FileMap m;
    if (!m.open(fd, len)) { printf("no mapping\n"); return 3; }
    std::vector<uint8_t> src(len);
    for (size_t i = 0; i < len; ++i) src[i] = (uint8_t)((i * 2654435761u) >> 13);
    // spans of odd sizes, so that many of them straddle a border between two mappings
    for (size_t off = 0; off < len;) { const size_t n = std::min(len - off, (size_t)(1 + (off * 7919) % 3000000)); m.put(off, src.data() + off, n); off += n; }
    // single pages through at(), as the host page writers store them
    for (size_t off = 0; off + 4096 <= len; off += 4096 * 257) memcpy(m.at(off), src.data() + off, 4096);
    printf("pieces of %zu bytes, one mapping: %d, area %zu\n", m.chunk, (int)m.one, m.area);
    m.close();
    close(fd);
    return 0;
}
'''


@pytest.mark.parametrize("one", ["0", "1"])
def test_filemap_pieces(tmp_path, one):
    exe = tmp_path / "filemap_probe"
    (tmp_path / "p.cpp").write_text(SRC)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "pyfastx_amd", "csrc"), "-o", str(exe), str(tmp_path / "p.cpp"), "-lpthread"])
    for length in (5, 4096, (8 << 20) - 1, (8 << 20) + 4097, 3 * (8 << 20) + 12345):
        f = tmp_path / ("f_%d.bin" % length)
        env = dict(os.environ, FX_FXI_ONE_MAPPING=one)
        env.pop("FX_FXI_NO_MMAP", None)
        out = subprocess.run([str(exe), str(f), str(length)], env=env, capture_output=True, text=True)
        assert out.returncode == 0, (out.stdout, out.stderr)
        assert ("one mapping: %s" % one) in out.stdout
        i = np.arange(length, dtype=np.uint64)
        want = (((i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)) >> np.uint64(13)).astype(np.uint8)
        got = np.fromfile(str(f), dtype=np.uint8)
        assert got.size == length and (got == want).all(), (length, one)
