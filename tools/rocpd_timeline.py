#!/usr/bin/env python3
"""The kernels of the LAST step of a rocprofv3 kernel trace (rocpd sqlite) in launch order: start offset,
duration, idle gap before the kernel.  A step starts at a launch of `--mark` (default: gi_count_kernel<KeysCoo>,
the first kernel of the graph-index build).

    python tools/rocpd_timeline.py /tmp/p_k/k_results.db [--mark NAME] > profiles/rNN_step_timeline.md
"""
import argparse
import sqlite3

from rocpd_summary import short

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--mark", default="gi_count_kernel<KeysCoo>")
ap.add_argument("--end-mark", default="multi_tensor_apply_kernel", help="the step ends with the last launch of this kernel after the mark")
ap.add_argument("--nth-last", type=int, default=1, help="start at the n-th last launch of the mark (a step may launch it more than once)")
ap.add_argument("--gaps-only", action="store_true", help="print only launches that follow an idle gap of more than 3 us")
args = ap.parse_args()
rows = list(sqlite3.connect(args.db).execute("select name, start, end from kernels order by start"))
marks = [i for i, r in enumerate(rows) if args.mark in short(r[0])]
if not marks:
    raise SystemExit(f"no kernel matching {args.mark!r}")
first = marks[-args.nth_last]
last = first
for i in range(first, len(rows)):
    if args.end_mark in rows[i][0]:
        last = i
if last == first:
    last = len(rows) - 1
t0, prev_end = rows[first][1], rows[first][1]
busy = 0
print(f"# last step of `{args.db}`: kernels in launch order\n")
print("| # | start ms | dur us | gap us | kernel |")
print("|---:|---:|---:|---:|---|")
for k, (name, s, e) in enumerate(rows[first:last + 1]):
    if not args.gaps_only or s - prev_end > 3000:
        print(f"| {k} | {(s - t0) / 1e6:.3f} | {(e - s) / 1e3:.1f} | {max(0, s - prev_end) / 1e3:.1f} | `{short(name)}` |")
    busy += e - s
    prev_end = max(prev_end, e)
span = rows[last][2] - t0
print(f"\nspan {span / 1e6:.3f} ms, kernels {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms, {last - first + 1} launches")
