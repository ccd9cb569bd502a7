"""The C-ABI library builds, loads and exports every symbol include/cmf_amd.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "cmf_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cmf_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built_lib():
    from cmf_amd.build import build
    return build(verbose=False)


def test_header_declares_what_python_binds():
    from cmf_amd import _lib
    assert header_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    for name in header_symbols():
        assert hasattr(lib, name), f"{name} declared in include/cmf_amd.h but not exported"
    lib.cmf_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.cmf_version()


def test_loader_binds_and_fails_loudly_when_missing(built_lib, monkeypatch, tmp_path):
    from cmf_amd import _lib
    assert _lib.load() is not None
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_code_object_targets_gfx950_only(built_lib):
    import subprocess
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", built_lib], capture_output=True, text=True)
    blob = open(built_lib, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90", b"gfx1100"):
        assert other not in blob


def test_struct_layouts_match_header():
    """ctypes mirrors of the two argument structs have the field order of the header."""
    from cmf_amd import _lib
    text = open(os.path.join(ROOT, "include", "cmf_amd.h")).read()

    def fields(struct):
        end = text.index("} " + struct + ";")
        body = text[text.rindex("typedef struct {", 0, end) + len("typedef struct {"):end]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            ids = re.sub(r"\b(const|float|int|long|void|unsigned|char)\b", "", decl).replace("*", "")
            names += [t.strip() for t in ids.split(",") if t.strip()]
        return names

    assert fields("cmf_conv_tangent_args") == [f[0] for f in _lib.ConvTangentArgs._fields_]
    assert fields("cmf_conv_primal_args") == [f[0] for f in _lib.ConvPrimalArgs._fields_]
