#!/usr/bin/env python3
"""Measured pipe utilisation per kernel from one rocprofv3 --pmc pass (SQ counters) -> the JSON that
bench.py's cfg5 / DBSCAN stage blocks quote instead of a flop-count estimate (the pruned searches
visit a data-dependent few per cent of the pairs: there is no fixed flop count to price them with).

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY \
              SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d out -o s -- python bench.py --workload cfg5 ...
    python tools/make_pipe_json.py out/s_results.db "<about>" > profiles/rNN_pipe_util_cfg5.json

valu_busy = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE): the share of the chip's vector
issue slots the kernel's wave-instructions occupy (a wave64 VALU instruction holds its SIMD for 4
cycles; MFMA instructions are counted in SQ_INSTS_VALU and subtracted).  issue / wait / stall =
SQ_ACTIVE_INST_ANY / SQ_WAIT_ANY / SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES (MI355X_MICROARCH.md, PMC
section: the three are disjoint shares of the waves' cycles).
"""
import json
import re
import sqlite3
import sys
from collections import defaultdict

N_SIMD = 1024


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("gnntrk::", "")
    return name


def main():
    db, about = sys.argv[1], sys.argv[2]
    cur = sqlite3.connect(db).cursor()
    # (SQ counters: summed over their instances = the whole chip; GRBM_GUI_ACTIVE: one value per XCD,
    #  all equal to the kernel's duration in cycles -> the mean)
    q = ("select name, counter_name, dispatch_id, sum(counter_value), avg(counter_value), max(duration) from pmc_events "
         "group by name, counter_name, dispatch_id")
    vals = defaultdict(lambda: defaultdict(list))
    durs = defaultdict(dict)
    for name, cn, did, v, av, d in cur.execute(q):
        vals[short(name)][cn].append(av if cn.startswith("GRBM") else v)
        durs[short(name)][did] = d
    total = sum(sum(d.values()) for d in durs.values()) or 1
    out = {"_about": about, "kernels": {}}
    for k in sorted(durs, key=lambda k: -sum(durs[k].values())):
        if sum(durs[k].values()) < 0.002 * total or not any(c in vals[k] for c in ("SQ_INSTS_VALU",)):
            continue
        m = {c: sum(v) / len(v) for c, v in vals[k].items()}
        n = len(durs[k])
        gui = m.get("GRBM_GUI_ACTIVE", 0.0)
        wave = m.get("SQ_WAVE_CYCLES", 0.0)
        valu = m.get("SQ_INSTS_VALU", 0.0) - m.get("SQ_INSTS_MFMA", 0.0)
        rec = {"launches_profiled": n, "avg_us_under_pmc": round(sum(durs[k].values()) / n / 1e3, 1),
               "valu_insts": valu, "mfma_insts": m.get("SQ_INSTS_MFMA", 0.0), "salu_insts": m.get("SQ_INSTS_SALU", 0.0),
               "lds_insts": m.get("SQ_INSTS_LDS", 0.0)}
        if gui:
            rec["valu_busy"] = round(valu * 4.0 / (N_SIMD * gui), 4)
        if wave:
            rec["issue_share"] = round(m.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 4)
            rec["wait_share"] = round(m.get("SQ_WAIT_ANY", 0.0) / wave, 4)
            rec["stall_share"] = round(m.get("SQ_WAIT_INST_ANY", 0.0) / wave, 4)
        out["kernels"][k] = rec
    # which build the counters describe: bench.py marks a block read from a profile of another build as stale
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out["_csrc_sha256"] = bench.library_id()["csrc_sha256"]
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
