#!/usr/bin/env python3
"""Per-kernel PMC summary of a rocprofv3 --pmc run (rocpd sqlite): for every kernel
name whose total duration is >= 1 % of the run, the mean per-dispatch value of every
collected counter (values are summed over XCDs/SEs by rocprofv3's dimension rows).

    python tools/rocpd_pmc_summary.py <results.db> [name-filter]
"""

import re
import sqlite3
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("gnntrk::", "")
    return name if len(name) <= 80 else name[:77] + "..."


def main(path, flt=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    # sum over counter dimension instances per dispatch, then average per kernel name
    q = ("select name, counter_name, dispatch_id, sum(counter_value), max(duration) "
         "from pmc_events group by name, counter_name, dispatch_id")
    per = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    for name, cn, did, val, d in cur.execute(q):
        per[name][cn].append(val)
        dur[name][did] = d
    tot = {n: sum(v.values()) for n, v in dur.items()}
    total = sum(tot.values()) or 1
    print(f"# rocprofv3 --pmc per-kernel means of `{path}`\n")
    for name in sorted(tot, key=lambda n: -tot[n]):
        if tot[name] < 0.01 * total or (flt and flt not in name):
            continue
        n = len(dur[name])
        print(f"## `{short(name)}`  dispatches={n}  avg_duration_us={tot[name]/n/1e3:.1f}")
        for cn in sorted(per[name]):
            v = per[name][cn]
            print(f"  {cn:32s} {sum(v)/len(v):.6g}")
        print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
