"""The C-ABI shared library loads on a GPU-less box and exports every symbol that
include/wedm_hip.h declares (no compute calls here)."""
from __future__ import annotations

import ctypes as C
import re
from pathlib import Path

import pytest

from sparc_amd import _abi, _lib

ROOT = Path(__file__).resolve().parents[1]


def declared_functions():
    text = (ROOT / "include" / "wedm_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wedm_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ("wedm_create", "wedm_destroy", "wedm_bind_state", "wedm_bind_geometry", "wedm_reset", "wedm_step",
                 "wedm_set_kernel", "wedm_set_lanes", "wedm_last_error", "wedm_last_kernel", "wedm_abi_version"):
        assert must in names


def test_library_exports_every_declared_symbol():
    if not _lib.LIB_PATH.exists():
        pytest.fail(f"{_lib.LIB_PATH} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = _lib.load()
    for name in declared_functions():
        assert hasattr(L, name), f"{name} declared in include/wedm_hip.h but not exported"
    assert L.wedm_abi_version() == _abi.ABI_VERSION
    assert L.wedm_sizeof_params() == C.sizeof(_abi.Params)


def test_build_id_names_the_sources_the_library_was_built_from():
    """`wedm_build_id()` = sha256 over the kernel sources + compiler flags at build time: the in-tree library is the one
    `__graft_entry__.build()` makes from the sources as they are (an edited kernel with a stale library fails here), and
    the id is what bench.py compares the recorded counter rows of profiles/*.json with."""
    import __graft_entry__ as g

    assert _lib.build_id() == g.kernel_build_id() and len(_lib.build_id()) == 16


def test_bad_arguments_return_status_codes_not_crashes():
    L = _lib.load()
    assert L.wedm_create(None, 4, 4, None) == _abi.ERR_BAD_ARG
    assert L.wedm_destroy(None) == _abi.ERR_BAD_ARG
    assert L.wedm_set_kernel(None, 0) == _abi.ERR_BAD_ARG
    assert b"null pointer" in L.wedm_last_error(None)


def test_enum_mirror_matches_header():
    text = (ROOT / "include" / "wedm_hip.h").read_text()
    for enum_name, py_enum, count in (("wedm_f64_field", _abi.F64, _abi.F64_COUNT), ("wedm_i32_field", _abi.I32, _abi.I32_COUNT),
                                      ("wedm_i8_field", _abi.I8, _abi.I8_COUNT), ("wedm_geom_f64_field", _abi.GF64, _abi.GEOM_F64_COUNT),
                                      ("wedm_geom_i32_field", _abi.GI32, _abi.GEOM_I32_COUNT)):
        body = re.search(r"enum %s \{(.*?)\};" % enum_name, text, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = [n.strip().split("=")[0].strip() for n in body.split(",") if n.strip()]
        assert len(names) == count + 1, enum_name  # + the _COUNT sentinel
        prefix = {"wedm_f64_field": "WEDM_F_", "wedm_i32_field": "WEDM_I_", "wedm_i8_field": "WEDM_B_",
                  "wedm_geom_f64_field": "WEDM_G_", "wedm_geom_i32_field": "WEDM_GI_"}[enum_name]
        for idx, n in enumerate(names[:-1]):
            assert py_enum[n[len(prefix):]].value == idx, (enum_name, n)


def test_header_is_plain_c_and_struct_layouts_match_the_ctypes_mirrors(tmp_path):
    """include/wedm_hip.h compiles as strict C99 (what a cgo / JNI / ctypes binder consumes) and
    every structure crossing the boundary has the size and field offsets of its Python mirror."""
    import subprocess

    mirrors = {"wedm_params": _abi.Params, "wedm_state_ptrs": _abi.StatePtrs, "wedm_geom_ptrs": _abi.GeomPtrs,
               "wedm_action_ptrs": _abi.ActionPtrs, "wedm_trace_desc": _abi.TraceDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "wedm_hip.h"', "int main(void) {"]
    for cname, mirror in mirrors.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for field, _ in mirror._fields_:
            lines.append(f'  printf("{cname} {field} %zu\\n", offsetof({cname}, {field}));')
    lines += ['  printf("abi %d stat_count %d\\n", WEDM_ABI_VERSION, (int)WEDM_STAT_COUNT);', "  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(src), "-o", str(exe)],
                   check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    seen = 0
    for row in out:
        parts = row.split()
        if len(parts) == 3 and parts[0] in mirrors:
            mirror = mirrors[parts[0]]
            if parts[1] == "size":
                assert C.sizeof(mirror) == int(parts[2]), row
            else:
                assert getattr(mirror, parts[1]).offset == int(parts[2]), row
            seen += 1
    assert seen > 100
    assert f"abi {_abi.ABI_VERSION} stat_count {_abi.STAT_COUNT}" in out
