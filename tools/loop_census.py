"""Instruction census of the loops of one kernel in a HIP object file (static counts from llvm-objdump).

    python tools/loop_census.py gnn_tracking_amd/csrc/_obj/mlp_bf16.o 'mlp16_bwd_kernel<1, 3, 2, true, false, 2, IoRelational<3> >' [--dump out.s]

A loop = the address range of a backward branch; printed from the largest down, with the number of
instructions per class (MFMA, other VALU split into moves / packs / packed-16 ops / rest, LDS, VMEM, SALU,
s_nop, s_waitcnt) so that a change of the tile loop can be read off the build, before it goes to the GPU.
"""
import collections
import re
import subprocess
import sys
import tempfile

BIN = "/opt/rocm/lib/llvm/bin/"


def disassemble(obj):
    with tempfile.TemporaryDirectory() as td:
        co, fat = td + "/dev.co", td + "/fat.bin"
        if obj.endswith(".co"):   # (an offload bundle as such: hipcc --cuda-device-only -c)
            fat = obj
        else:
            subprocess.check_call([BIN + "llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj])
        subprocess.check_call([BIN + "clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"],
                              stderr=subprocess.DEVNULL)
        return subprocess.run([BIN + "llvm-objdump", "-d", "-C", co], capture_output=True, text=True).stdout


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "v_acc_mov"
    if op in ("v_mov_b32_e32", "v_mov_b32_e64", "v_mov_b64_e32", "v_mov_b64_e64", "v_mov_b32_dpp", "v_mov_b64"):
        return "v_mov"
    if op.startswith("v_cvt_pk_bf16"):
        return "v_pack"
    if op.startswith("v_pk_"):
        return "v_pk16"
    if op.startswith("v_cndmask"):
        return "v_select"
    if op.startswith("v_cmp"):
        return "v_cmp"
    if op.startswith("v_"):
        return "v_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op == "s_nop":
        return "s_nop"
    if op == "s_waitcnt":
        return "s_waitcnt"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    obj, name = sys.argv[1], sys.argv[2]
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    text = disassemble(obj)
    blocks = re.split(r"\n(?=[0-9a-f]{16} <)", text)
    blk = [b for b in blocks if name in b.split("\n", 1)[0]]
    if not blk:
        sys.exit(f"no kernel matching {name!r}")
    lines = blk[0].split("\n")[1:]
    insts = []
    for ln in lines:
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", ln)
        if m:
            insts.append((int(m.group(3), 16), m.group(1), m.group(2)))
    if dump:
        with open(dump, "w") as f:
            f.write("\n".join(f"{a:08x} {o} {r}" for a, o, r in insts))
    addr_index = {a: i for i, (a, _, _) in enumerate(insts)}
    loops = []
    for i, (a, op, rest) in enumerate(insts):
        if op.startswith("s_cbranch") or op == "s_branch":
            off = int(rest.split()[0])
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + 4 * off
            if tgt <= a and tgt in addr_index:
                loops.append((addr_index[tgt], i))
    has_mfma = lambda t: any(op.startswith("v_mfma") for _, op, _ in insts[t[0]:t[1] + 1])
    loops = sorted((t for t in loops if has_mfma(t)), key=lambda t: t[0] - t[1])
    print(f"{len(insts)} instructions, {len(loops)} loops with MFMAs (largest first)")
    verbose = "-v" in sys.argv
    for lo, hi in loops:
        c = collections.Counter(classify(op) for _, op, _ in insts[lo:hi + 1])
        n = hi - lo + 1
        nops = sum(int(r.split()[0]) + 1 for _, op, r in insts[lo:hi + 1] if op == "s_nop")
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        print(f"loop [{lo}..{hi}] {n} instructions: mfma {c['mfma']}, other VALU {valu} "
              f"({', '.join(f'{k[2:]} {v}' for k, v in sorted(c.items()) if k.startswith('v_'))}), "
              f"lds {c['lds']}, vmem {c['vmem']}, salu {c['salu']}, s_waitcnt {c['s_waitcnt']}, "
              f"s_nop {c['s_nop']} ({nops} idle states)")
        ops = collections.Counter(op for _, op, _ in insts[lo:hi + 1])
        if verbose:
            print("   ", ", ".join(f"{k} {v}" for k, v in ops.most_common(45)))


if __name__ == "__main__":
    main()
