"""The C-ABI shared library builds, loads without a GPU, and exports every symbol include/dsvg.h declares
(no compute calls here)."""
import ctypes
import os
import re

import pytest

from deepsvg_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "dsvg.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dsvg_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(L, name), f"{name} is declared in include/dsvg.h but not exported"
    # and the Python binding table covers exactly the declared surface
    assert sorted(lib.SIGNATURES) == declared


def test_gemm_desc_layout_matches_header():
    """field order of the ctypes mirror == field order of struct dsvg_gemm_desc"""
    txt = open(os.path.join(ROOT, "include", "dsvg.h")).read()
    body = re.search(r"typedef struct dsvg_gemm_desc \{(.*?)\} dsvg_gemm_desc;", txt, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        parts = [p.strip() for p in decl.split(",")]
        fields.append(re.split(r"[\s\*]+", parts[0])[-1])
        fields.extend(p.lstrip("* ") for p in parts[1:])
    assert fields == [f[0] for f in lib.GemmDesc._fields_]


def test_error_reporting_without_gpu():
    L = lib.load()
    hdr = open(os.path.join(ROOT, "include", "dsvg.h")).read()
    assert L.dsvg_version() == lib.ABI_VERSION == int(re.search(r"#define DSVG_ABI_VERSION (\d+)", hdr).group(1))
    rc = L.dsvg_gemm(None, None)
    assert rc != 0 and b"null desc" in L.dsvg_last_error()
    with pytest.raises(lib.DsvgError):
        lib.check(rc, "dsvg_gemm")
