#!/usr/bin/env python3
"""HBM bytes per launch of the fused-MLP kernels from two rocprofv3 --pmc passes of the SAME
bench.py command (FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md, TCC row)
-> the JSON bench.py's ``roofline.traffic`` reads.

    python tools/make_traffic_json.py <fetch_results.db> <write_results.db> <rows_per_launch> "<about>" > profiles/rNN_hbm_traffic_bf16.json

Correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE reports half of the bytes of
wide coalesced reads -> bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (both counters are in KiB).
For narrow random accesses the factor 2 is an upper bound (the guide calibrates it on 16-byte
streaming loads only); ratios between variants of one kernel are unaffected.
"""
import json
import re
import sqlite3
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("gnntrk::", "")
    return name


def per_kernel(path: str, counter: str):
    cur = sqlite3.connect(path).cursor()
    q = ("select name, dispatch_id, sum(counter_value), max(duration) from pmc_events "
         "where counter_name = ? group by name, dispatch_id")
    vals, durs = defaultdict(list), defaultdict(list)
    for name, _did, v, d in cur.execute(q, (counter,)):
        vals[short(name)].append(v)
        durs[short(name)].append(d)
    return ({k: sum(v) / len(v) for k, v in vals.items()}, {k: sum(v) / len(v) / 1e3 for k, v in durs.items()},
            {k: len(v) for k, v in vals.items()})


def main():
    fetch_db, write_db, rows, about = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    f, fdur, fn = per_kernel(fetch_db, "FETCH_SIZE")
    w, _, _ = per_kernel(write_db, "WRITE_SIZE")
    out = {"_about": about + "  bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md: FETCH_SIZE counts "
           "half of wide coalesced reads on gfx950; an upper bound for narrow random accesses).", "kernels": {}}
    for k in sorted(f, key=lambda k: -fdur[k] * fn[k]):
        if not (k.startswith("mlp16_") or k.startswith("mlp_") or k.startswith("segment_sum")) or k not in w:
            continue
        b = (2 * f[k] + w[k]) * 1024
        out["kernels"][k] = {"rows_per_launch": rows, "FETCH_SIZE_KiB": f[k], "WRITE_SIZE_KiB": w[k],
                             "hbm_bytes_per_launch": b, "hbm_bytes_per_row": round(b / rows, 2),
                             "avg_us_under_pmc": round(fdur[k], 1), "launches_profiled": fn[k]}
    # which build the counters describe: bench.py marks a block read from a profile of another build as stale
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out["_csrc_sha256"] = bench.library_id()["csrc_sha256"]
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
