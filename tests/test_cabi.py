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
    assert lib.dm2_abi_version() == _C.ABI_VERSION == int(re.search(r"#define DM2_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "dm2_hip.h")).read()).group(1))


def header_struct(name):
    """[(field, size, is_pointer)] of `typedef struct name {...} name;` in include/dm2_hip.h, in declaration order."""
    hdr = open(os.path.join(ROOT, "include", "dm2_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), hdr, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        m = re.match(r"(const\s+)?(\w+)\s*(\*?)\s*(.*)$", decl, flags=re.S)
        ctype, ptr, names = m.group(2), m.group(3) == "*", m.group(4)
        for nm in names.split(","):
            nm = nm.strip()
            is_ptr = ptr or nm.startswith("*")
            size = 8 if is_ptr else {"int32_t": 4, "float": 4, "int64_t": 8, "uint32_t": 4}[ctype]
            fields.append((nm.lstrip("* "), size, is_ptr))
    return fields


def test_structs_match_header_layout():
    """Field ORDER, sizes and offsets of the three descriptor structs: the ctypes mirrors in _C.py against the declarations
    of include/dm2_hip.h (natural alignment, as the C compiler lays them out)."""
    import ctypes
    for name, cls in (("dm2_render_desc", _C.RenderDesc), ("dm2_layers_desc", _C.LayersDesc), ("dm2_prep_desc", _C.PrepDesc)):
        decl = header_struct(name)
        assert [f[0] for f in decl] == [f[0] for f in cls._fields_], name
        off = 0
        for (nm, size, is_ptr), (cnm, ctype) in zip(decl, cls._fields_):
            off = (off + size - 1) // size * size
            cf = getattr(cls, cnm)
            assert (cf.offset, cf.size) == (off, size), (name, nm, cf.offset, off)
            assert (ctype is ctypes.c_void_p) == is_ptr, (name, nm)
            off += size
        assert ctypes.sizeof(cls) == (off + 7) // 8 * 8, name


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
