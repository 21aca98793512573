"""Build a VARIANT of libgnntrk.so for A/B measurements: the named sources are recompiled with extra
compiler arguments, the other objects are the ones of the regular build.

    python tools/build_variant.py prio mlp_bf16.hip,mlp_bf16_g32.hip -DGNNTRK_BWD_PRIO=1
    GNNTRK_LIB=tools/_bin/variants/prio/libgnntrk.so python tools/bench_bwd_io.py --only 0

The library lands in tools/_bin/variants/<name>/ (git-ignored, travels with gpurun).
"""
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from gnn_tracking_amd import _build  # noqa: E402


def main():
    name, srcs, extra = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    _build.build_lib()
    out = ROOT / "tools" / "_bin" / "variants" / name
    out.mkdir(parents=True, exist_ok=True)
    objs = []
    procs = []
    for src in sorted(_build.CSRC.glob("*.hip")):
        if src.name in srcs:
            obj = out / (src.stem + ".o")
            cmd = [_build._hipcc(), *_build.FLAGS, *_build.EXTRA_FLAGS.get(src.name, []), *extra, "-c", str(src),
                   "-o", str(obj)]
            procs.append((subprocess.Popen(cmd), src.name))
            objs.append(obj)
        else:
            objs.append(_build.OBJ / (src.stem + ".o"))
    for p, n in procs:
        if p.wait() != 0:
            raise SystemExit(f"hipcc failed on {n}")
    lib = out / "libgnntrk.so"
    subprocess.check_call([_build._hipcc(), f"--offload-arch={_build.ARCH}", "-shared", "-fPIC", "-o", str(lib),
                           *map(str, objs)])
    print(lib)


if __name__ == "__main__":
    main()
