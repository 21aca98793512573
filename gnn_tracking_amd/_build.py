"""Build libgnntrk.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc.

In-tree build: objects under ``gnn_tracking_amd/csrc/_obj``, library at
``gnn_tracking_amd/libgnntrk.so`` (git-ignored; travels to the GPU box with the
snapshot).  hipcc cross-compiles without a GPU.
"""

from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import pathlib
import shutil
import subprocess
import sys

PKG = pathlib.Path(__file__).resolve().parent
CSRC = PKG / "csrc"
INCLUDE = PKG.parent / "include"
OBJ = CSRC / "_obj"
LIB = PKG / "libgnntrk.so"
ARCH = "gfx950"

FLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
    "-Wall", "-Wno-unused-function", "-Wno-pass-failed", f"-I{INCLUDE}", f"-I{CSRC}",
]


# per-file additions: the bf16 kernels convert every MFMA result straight away (pack to
# bf16), so results should land in VGPRs instead of AGPRs + v_accvgpr_read copies
EXTRA_FLAGS = {"mlp_bf16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "mlp_bf16_fwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-mllvm",
                                    "-amdgpu-sched-strategy=max-ilp"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and pathlib.Path(c).exists():
            return c
    raise RuntimeError("hipcc not found: cannot build libgnntrk.so (ROCm toolchain required)")


def have_hipcc() -> bool:
    try:
        _hipcc()
        return True
    except RuntimeError:
        return False


def _stamp(src: pathlib.Path) -> str:
    h = hashlib.sha256()
    h.update(" ".join(FLAGS + EXTRA_FLAGS.get(src.name, [])).encode())
    h.update(src.read_bytes())
    for hdr in sorted(list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc")) + list(INCLUDE.glob("*.h"))):
        h.update(hdr.read_bytes())
    return h.hexdigest()


def _compile(src: pathlib.Path, verbose: bool) -> pathlib.Path:
    obj = OBJ / (src.stem + ".o")
    stamp = OBJ / (src.stem + ".stamp")
    want = _stamp(src)
    if obj.exists() and stamp.exists() and stamp.read_text() == want:
        return obj
    tmp = obj.with_suffix(f".o.tmp{os.getpid()}")
    cmd = [_hipcc(), *FLAGS, *EXTRA_FLAGS.get(src.name, []), "-c", str(src), "-o", str(tmp)]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        tmp.unlink(missing_ok=True)
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError(f"hipcc failed on {src.name}")
    if verbose and r.stderr.strip():
        sys.stderr.write(r.stderr)
    os.replace(tmp, obj)   # (a reader never sees a half-written object)
    stamp.write_text(want)
    return obj


def build_lib(verbose: bool = False, jobs: int | None = None) -> pathlib.Path:
    """Compile what changed and relink.  Safe under concurrent callers (ranks started by torchrun or
    Lightning DDP after a source edit): one process builds under an exclusive file lock, the others
    wait and then find everything up to date; outputs are moved into place atomically."""
    import fcntl

    OBJ.mkdir(parents=True, exist_ok=True)
    with open(OBJ / ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(verbose, jobs)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose: bool, jobs: int | None) -> pathlib.Path:
    srcs = sorted(CSRC.glob("*.hip"))
    jobs = jobs or min(len(srcs), max(1, (os.cpu_count() or 2) - 1))
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, verbose), srcs))
    newest = max(o.stat().st_mtime for o in objs)
    if not LIB.exists() or LIB.stat().st_mtime < newest:
        tmp = LIB.with_suffix(f".so.tmp{os.getpid()}")
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(tmp),
               *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            tmp.unlink(missing_ok=True)
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link of libgnntrk.so failed")
        os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build_lib(verbose=True))
