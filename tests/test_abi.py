"""CPU-side checks of the boundary: the shared library loads, exports every
symbol that include/openrec_hip.h declares, and fails loudly (no fallback)
without a device.  No compute call is made here."""
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "openrec_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(orx_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from openrec_amd import _ffi
    lib = _ffi.load()
    syms = _declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in openrec_hip.h but not exported"
        assert s in _ffi.SIGNATURES, f"{s} has no ctypes signature in _ffi.SIGNATURES"
    assert set(_ffi.SIGNATURES) == set(syms)
    assert lib.orx_version() >= 100


def test_no_cpu_fallback():
    """Without a GPU every entry point must fail with an error, never compute."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from openrec_amd import runtime as rt
    with pytest.raises(Exception):
        rt.Context(0)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "openrec_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{f} mentions the oracle"


def test_the_library_in_the_tree_is_built_from_the_sources_in_the_tree():
    """__graft_entry__.build() rebuilds by content, not by mtime (a shipped snapshot has arbitrary mtimes): after a build the hash
    file beside the .so names the sources it was built from, and a changed source asks for a rebuild."""
    from openrec_amd import build
    build.build()
    assert os.path.exists(build.LIB) and not build.needs_build()
    assert open(build.HASH_FILE).read().strip() == build.source_hash()
    saved = open(build.HASH_FILE).read()
    try:
        open(build.HASH_FILE, "w").write("0" * 16)          # as if a source had changed since the build
        assert build.needs_build()
    finally:
        open(build.HASH_FILE, "w").write(saved)
    assert not build.needs_build()
