"""VGPR / SGPR / LDS / spill figures of the kernels in a HIP object file (from the code object's metadata notes).

    python tools/kernel_resources.py gnn_tracking_amd/csrc/_obj/mlp_bf16.o [name-substring]
"""
import re
import subprocess
import sys
import tempfile

BIN = "/opt/rocm/lib/llvm/bin/"


def main():
    obj, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    with tempfile.TemporaryDirectory() as td:
        co, fat = td + "/dev.co", td + "/fat.bin"
        subprocess.check_call([BIN + "llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj])
        subprocess.check_call([BIN + "clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"],
                              stderr=subprocess.DEVNULL)
        notes = subprocess.run([BIN + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for blk in notes.split("- .agpr_count:")[1:]:
        get = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
        name = get("name")
        if pat not in name:
            continue
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(anonymous namespace\)::|gnntrk::", "", dem).split("(")[0]
        print(f"vgpr {get('vgpr_count'):>4} agpr {blk.split()[0]:>3} sgpr {get('sgpr_count'):>4} lds {get('group_segment_fixed_size'):>6} "
              f"spill {get('vgpr_spill_count'):>3} scratch {get('private_segment_fixed_size'):>4}  {dem}")


if __name__ == "__main__":
    main()
