"""CPU: the C-ABI library loads here (no GPU, no compute calls) and exports every symbol that
include/mi355x_rec.h declares, with the argument counts the ctypes binding assumes."""
import os
import re
import subprocess

from mi355x_rec import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    src = open(os.path.join(ROOT, "include", "mi355x_rec.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(mi_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        decls[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return decls


def test_header_binding_and_library_agree(lib):
    decls = _header_decls()
    assert len(decls) >= 30
    assert set(decls) == set(_lib.SIGNATURES), set(decls) ^ set(_lib.SIGNATURES)
    for name, nargs in decls.items():
        assert hasattr(lib, name), name
        assert len(_lib.SIGNATURES[name][1]) == nargs, name
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (mi_\w+)", out))
    assert set(decls) <= exported
    # nothing but the C ABI leaks out of the library
    assert all(s.startswith("mi_") for s in exported), exported


def test_library_identity(lib):
    assert lib.mi_abi_version() == _lib.ABI_VERSION
    assert b"gfx950" in lib.mi_build_info()
    assert lib.mi_last_error() is not None


def test_only_gfx950_code_objects():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          "--input=" + _lib.LIB_PATH], capture_output=True, text=True)
    if out.returncode == 0 and out.stdout.strip():
        targets = [t for t in out.stdout.split() if "amdgcn" in t]
        assert targets and all("gfx950" in t for t in targets), targets


def test_shipped_library_reads_no_environment_variables():
    """tuning switches live in the tools' build (make -C csrc tuning, -DMI_TUNING) only — the product
    library does not even import getenv."""
    out = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in out
