"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/zebra_hip.h declares, the ctypes table matches the header, and -- with no GPU -- the
product path fails loudly instead of falling back to anything."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "zebra_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    return sorted(set(re.findall(r"^ZH_API[^;(]*?\b(zh_\w+)\s*\(", text, flags=re.M)))


def test_header_declares_the_documented_entry_points():
    syms = declared_symbols()
    for must in ("zh_index_create", "zh_index_add", "zh_index_build", "zh_search_batch", "zh_search_batch_device",
                 "zh_hash_signs", "zh_distance_batch", "zh_merge_topk_device", "zh_stats", "zh_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from zebra_amd import _ffi
    assert os.path.exists(_ffi.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in zebra_hip.h but not exported"
    # and nothing undeclared leaks out of the library with a zh_ prefix
    out = subprocess.check_output(["nm", "-D", "--defined-only", _ffi.LIB_PATH], text=True)
    exported = sorted(set(re.findall(r"\bT (zh_\w+)", out)))
    assert exported == declared_symbols()


def test_ctypes_table_matches_header():
    from zebra_amd import _ffi
    assert sorted(n for n, _, _ in _ffi.SYMBOLS) == declared_symbols()
    _ffi.lib()  # binds every symbol


def test_struct_layouts_match_header():
    """compile a tiny C program against the header and compare sizeof/offsetof with the ctypes mirrors"""
    from zebra_amd import _ffi
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "zebra_hip.h"
int main(void){
  printf("%zu %zu %zu %zu\n", sizeof(zh_options), sizeof(zh_forest_view), sizeof(zh_forest_sizes), sizeof(zh_stats_t));
  printf("%zu %zu %zu\n", offsetof(zh_options, seed), offsetof(zh_options, id_base), offsetof(zh_stats_t, ms_hash));
  return 0; }'''
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        a, b = subprocess.check_output([exe], text=True).strip().split("\n")
    sizes = [int(x) for x in a.split()]
    offs = [int(x) for x in b.split()]
    assert sizes == [ctypes.sizeof(_ffi.Options), ctypes.sizeof(_ffi.ForestView), ctypes.sizeof(_ffi.ForestSizes),
                     ctypes.sizeof(_ffi.Stats)]
    assert offs == [_ffi.Options.seed.offset, _ffi.Options.id_base.offset, _ffi.Stats.ms_hash.offset]


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import zebra_amd
    with pytest.raises(zebra_amd.ZhError) as e:
        zebra_amd.LSHIndex(8)
    assert "no CPU fallback" in str(e.value)
    with pytest.raises(zebra_amd.ZhError):
        zebra_amd.L2SquaredDistance().distance([1, 2, 3, 4], [1, 2, 3, 5])


def test_product_does_not_import_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    pkg = os.path.join(ROOT, "zebra_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or f == "Makefile":
                assert "oracle" not in open(os.path.join(dp, f), errors="replace").read().replace("the oracle", "").replace("oracle/zebra_oracle.c", "").replace("oracle zo_", ""), f
    for f in os.listdir(os.path.join(ROOT, "include")):
        assert "zebra_oracle" not in open(os.path.join(ROOT, "include", f)).read()


def test_integration_doc_binds_every_entry_point():
    """INTEGRATION.md shows the reference-side (Rust) binding: its extern "C" blocks name every function of the header"""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    bound = set(re.findall(r"pub fn (zh_\w+)", text))
    assert sorted(set(declared_symbols()) - bound) == []
    assert sorted(bound - set(declared_symbols())) == []
