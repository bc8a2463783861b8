"""The Rust binding a maintainer of the reference adds (ffi/mrt-sys, source only: no Rust toolchain here) must declare
every function include/mrt.h declares, with the same number of arguments."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def c_functions():
    text = open(os.path.join(ROOT, "include", "mrt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(mrt_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return out


def rust_functions():
    text = open(os.path.join(ROOT, "ffi", "mrt-sys", "src", "lib.rs")).read()
    text = re.sub(r"//.*", "", text)
    out = {}
    for m in re.finditer(r"pub fn (mrt_[a-z0-9_]+)\s*\(([^;]*?)\)\s*(?:->[^;]*)?;", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if not args else len([a for a in args.split(",") if a.strip()])
    return out


def test_rust_binding_declares_the_whole_header():
    from ray_tracing_amd import _ffi
    c, r = c_functions(), rust_functions()
    assert set(_ffi.EXPORTS) <= set(c), set(_ffi.EXPORTS) - set(c)
    assert set(c) == set(r), (set(c) - set(r), set(r) - set(c))
    assert {k: c[k] for k in c} == {k: r[k] for k in c}
