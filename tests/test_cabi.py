"""CPU: libfxgpu.so loads, exports every function include/fxgpu.h declares, and
refuses to compute without a GPU (no silent CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared():
    hdr = open(os.path.join(ROOT, "include", "fxgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(fx_[a-z0-9_]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported():
    from pyfastx_amd import _lib
    L = _lib.lib()
    names = declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "%s declared in include/fxgpu.h but not exported" % n
    assert set(names) == set(_lib.SYMBOLS), set(names) ^ set(_lib.SYMBOLS)
    assert b"gfx950" in L.fx_version()


def test_summary_struct_layout_matches_header():
    from pyfastx_amd import _lib, shard
    assert C.sizeof(_lib.ShardSummary) == 8 * shard.NWORDS == 224
    assert [f[0] for f in _lib.ShardSummary._fields_] == shard.FIELDS


def test_no_cpu_fallback_without_gpu():
    from pyfastx_amd import _lib
    L = _lib.lib()
    if L.fx_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.FxError) as e:
        _lib.Blob.from_bytes(b">a\nACGT\n")
    assert e.value.code == _lib.FX_EDEVICE and "no CPU fallback" in str(e.value)
    with pytest.raises(_lib.FxError):
        _lib.revcomp_bytes(b"ACGT")
    import pyfastx_amd
    with pytest.raises(RuntimeError):                     # the object API surfaces it, it does not route around it
        pyfastx_amd.Fasta(os.path.join(ROOT, "tests", "data", "test.fa"), memory_index=True)


def test_missing_file_status():
    from pyfastx_amd import _lib
    h = C.c_void_p()
    rc = _lib.lib().fx_open_file(b"/nonexistent/file.fa", 0, C.byref(h))
    assert rc == _lib.FX_ENOENT
