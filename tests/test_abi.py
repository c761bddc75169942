"""CPU suite: the C-ABI library loads and exports every symbol include/gauss_hip.h declares.
No compute calls here (there is no GPU in the CPU test environment)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "gauss_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gauss_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_all_bound_and_exported():
    from gauss_amd import _lib, build
    build.build_hip()
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 14
    assert sorted(_lib.SYMBOLS) == declared
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.gauss_hip_version()


def _header_symbols(name):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    return sorted(set(re.findall(r"\b(gauss_[A-Za-z_0-9]+)\s*\(", src)))


@pytest.mark.parametrize("lib,header", [("libgauss_hip.so", "gauss_hip.h"), ("libgauss_host.so", "gauss_host.h")])
def test_libraries_export_the_c_abi_and_nothing_else(lib, header):
    """`nm -D --defined-only` of both libraries lists exactly the functions their header declares: no C++ internals (round 5's
    libgauss_hip.so exported `fail`, `trace_on`, `queues_exclusive`, ...), no libstdc++ instances, no kernel handles
    (-fvisibility=hidden + gauss_amd/csrc/exports.map; the reference exports one routine table, RcppExports.cpp:333-358)."""
    import shutil
    import subprocess
    from gauss_amd import build
    build.build_hip()
    build.build_host()
    nm = shutil.which("nm")
    if nm is None:
        pytest.skip("nm not available")
    out = subprocess.check_output([nm, "-D", "--defined-only", os.path.join(ROOT, "gauss_amd", "lib", lib)]).decode()
    exported = sorted(l.split()[-1] for l in out.splitlines() if l.strip())
    assert exported == _header_symbols(header)


def test_missing_library_fails_loudly(monkeypatch):
    from gauss_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(ROOT, "gauss_amd", "lib", "nope.so"))
    with pytest.raises(_lib.GaussHipError):
        _lib.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gauss_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "liboracle" not in text, f


def test_integration_snippets_compile_against_the_headers():
    """The driver-side code of INTEGRATION.md (with a mock of the few reference types it touches) must compile
    against include/gauss_hip.h: keeps the documented Rcpp bindings in step with the C ABI."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    subprocess.check_call([gxx, "-std=c++17", "-fsyntax-only", "-Wall", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "abi", "integration_snippets.cpp")])
    # and a plain C compiler must accept both headers (the boundary is a C ABI)
    gcc = shutil.which("gcc")
    src = os.path.join(root, "tests", "abi", "c_abi.c")
    if gcc and os.path.exists(src):
        subprocess.check_call([gcc, "-std=c99", "-fsyntax-only", "-Wall", "-I" + os.path.join(root, "include"), src])
