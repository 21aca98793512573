"""Compile the product kernel sources with g++ against the wave64 emulator shim.

TEST INFRASTRUCTURE ONLY (see tests/emul/shim/hip/hip_runtime.h).  Output:
tests/emul/_build/libgnntrk_emul[_asan].so with the same C ABI as libgnntrk.so,
operating on host pointers.
"""

from __future__ import annotations

import hashlib
import pathlib
import subprocess
import sys

HERE = pathlib.Path(__file__).resolve().parent
REPO = HERE.parent.parent
CSRC = REPO / "gnn_tracking_amd" / "csrc"
OUT = HERE / "_build"
SKIP = {"sort_pairs.hip"}  # rocPRIM unit; emul_runtime.cpp provides the host sort


def build(asan: bool = False, verbose: bool = False) -> pathlib.Path:
    OUT.mkdir(exist_ok=True)
    name = "libgnntrk_emul_asan.so" if asan else "libgnntrk_emul.so"
    lib = OUT / name
    srcs = [s for s in sorted(CSRC.glob("*.hip")) if s.name not in SKIP]
    deps = srcs + sorted(CSRC.glob("*.h")) + sorted(CSRC.glob("*.inc")) + sorted((REPO / "include").glob("*.h")) + [
        HERE / "shim/hip/hip_runtime.h", HERE / "emul_runtime.cpp"]
    h = hashlib.sha256()
    for d in deps:
        h.update(d.read_bytes())
    stamp = OUT / (name + ".stamp")
    if lib.exists() and stamp.exists() and stamp.read_text() == h.hexdigest():
        return lib
    flags = ["-std=c++17", "-O2", "-fPIC", "-shared", "-pthread", "-ffp-contract=off",
             "-Wno-attributes", "-Wno-unknown-pragmas", "-Wno-ignored-attributes",
             "-DGNNTRK_GI_HUB_BITMAP_MIN=8192",   # (the graph index's large-hub path from 8 192 edges on: reachable at emulator sizes)
             f"-I{HERE / 'shim'}", f"-I{REPO / 'include'}", f"-I{CSRC}"]
    if asan:
        flags += ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"]
    # one object per translation unit, compiled in parallel (the bf16 kernel headers make a
    # single g++ invocation the longest step of a fresh CPU test run)
    import concurrent.futures as cf
    import os

    cflags = [f for f in flags if f != "-shared"]
    units = [(s, OUT / (s.stem + (".asan" if asan else "") + ".o")) for s in srcs]
    units.append((HERE / "emul_runtime.cpp", OUT / ("emul_runtime" + (".asan" if asan else "") + ".o")))

    def compile_one(unit):
        src, obj = unit
        cmd = ["g++", *cflags, "-c", "-x", "c++", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        return subprocess.run(cmd, capture_output=True, text=True)

    with cf.ThreadPoolExecutor(max_workers=max(1, min(8, (os.cpu_count() or 2)))) as ex:
        results = list(ex.map(compile_one, units))
    for r in results:
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("emulator build failed")
    link = ["g++", "-shared", "-pthread", *(["-fsanitize=address,undefined"] if asan else []),
            *[str(o) for _, o in units], "-o", str(lib)]
    r = subprocess.run(link, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("emulator build failed")
    stamp.write_text(h.hexdigest())
    return lib


if __name__ == "__main__":
    print(build(asan="--asan" in sys.argv, verbose=True))
