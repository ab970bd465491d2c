"""The C-ABI library loads and exports every symbol include/dm2_hip.h declares (no compute calls)."""
import os
import re

from util import ROOT

from dmesh2_renderer_amd import _C


def declared_functions():
    hdr = open(os.path.join(ROOT, "include", "dm2_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(dm2_[a-z_0-9]+)\s*\(", hdr)))


def test_header_symbols_are_bound_and_exported():
    names = declared_functions()
    assert len(names) >= 9
    assert set(names) == set(_C.EXPORTS), (names, sorted(_C.EXPORTS))
    lib = _C.load_library()                       # raises if a declared symbol is missing
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.dm2_abi_version() == 5


def test_structs_match_header_layout():
    # 10 / 8 x 4-byte scalars followed by 18 / 11 pointers
    import ctypes
    assert ctypes.sizeof(_C.RenderDesc) == 10 * 4 + 18 * 8
    assert ctypes.sizeof(_C.LayersDesc) == 8 * 4 + 11 * 8


def test_cpu_tensors_are_refused():
    import pytest
    import torch
    from util import soup_args
    args, _ = soup_args(32, 32, 10, 1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        _C.render_forward_cuda(*args)


def test_missing_library_fails_loudly(tmp_path):
    import pytest
    with pytest.raises(RuntimeError, match="native library not found"):
        _C.load_library(str(tmp_path / "nope.so"))
