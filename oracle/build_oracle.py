#!/usr/bin/env python3
"""Build the compiled parts of the oracle (test infrastructure): oracle/knn_ref.c ->
oracle/_build/libknn_ref.so.  Called by __graft_entry__.build() and by the tests."""
import pathlib
import subprocess
import sys

HERE = pathlib.Path(__file__).resolve().parent
OUT = HERE / "_build"


def build() -> pathlib.Path:
    OUT.mkdir(exist_ok=True)
    lib = OUT / "libknn_ref.so"
    src = HERE / "knn_ref.c"
    if lib.exists() and lib.stat().st_mtime >= src.stat().st_mtime:
        return lib
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-fopenmp",
           str(src), "-o", str(lib), "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("building oracle/knn_ref.c failed")
    return lib


if __name__ == "__main__":
    print(build())
