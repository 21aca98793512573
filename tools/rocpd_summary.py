#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel stats table:
calls, total / average / min / max duration, share of GPU time, VGPR/LDS per dispatch.

    python tools/rocpd_summary.py gpurun_out/prof/xxx_results.db > profiles/xxx.md
"""

import re
import sqlite3
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("gnntrk::", "")
    return name if len(name) <= 90 else name[:87] + "..."


def main(path: str) -> None:
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    q = ("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start)"
         + (", max(vgpr_count), max(accum_vgpr_count), max(lds_size)"
            if {"vgpr_count", "lds_size"} <= set(cols) else ", 0, 0, 0")
         + " from kernels group by name order by 3 desc")
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace summary of `{path}`\n")
    print(f"total GPU kernel time: {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds B |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for name, n, tot, avg, mn, mx, vg, ag, lds in rows:
        print(f"| `{short(name)}` | {n} | {tot/1e6:.3f} | {avg/1e3:.1f} | {mn/1e3:.1f} | "
              f"{mx/1e3:.1f} | {100*tot/total:.1f} | {vg} | {ag} | {lds} |")


if __name__ == "__main__":
    main(sys.argv[1])
