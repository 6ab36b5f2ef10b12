#!/usr/bin/env python3
"""Per-kernel sums of the counters in rocprofv3 --pmc result databases (rocpd sqlite): python tools/pmc_db.py <dir-or-db> [name filter]
Prints, per kernel and dispatch geometry, the counter values averaged per dispatch."""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def main():
    paths = []
    for a in sys.argv[1:]:
        if os.path.isdir(a):
            paths += glob.glob(os.path.join(a, "**", "*.db"), recursive=True)
        elif a.endswith(".db"):
            paths.append(a)
    filt = [a for a in sys.argv[1:] if not os.path.isdir(a) and not a.endswith(".db")]
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for p in paths:
        c = sqlite3.connect(p)
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
        sfx = [t for t in tabs if t.startswith("rocpd_pmc_event")][0][len("rocpd_pmc_event"):]
        q = f"""select k.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, p.name, e.value, d.id
                from rocpd_pmc_event{sfx} e join rocpd_info_pmc{sfx} p on e.pmc_id = p.id
                join rocpd_kernel_dispatch{sfx} d on d.event_id = e.event_id
                join rocpd_info_kernel_symbol{sfx} k on k.id = d.kernel_id"""
        seen = defaultdict(set)
        for name, gx, gy, gz, pmc, val, did in c.execute(q):
            if filt and not any(f in name for f in filt):
                continue
            key = (name.split("(")[0][:70], gx, gy, gz)
            acc[key][pmc] += val
            seen[(key, pmc)].add(did)
        for (key, pmc), s in seen.items():
            cnt[key][pmc] += len(s)
    for key in sorted(acc):
        print(f"{key[0]}  grid {key[1]}x{key[2]}x{key[3]}")
        for pmc in sorted(acc[key]):
            print(f"    {pmc:32s} {acc[key][pmc] / max(cnt[key][pmc], 1):16.0f}  (per dispatch, {cnt[key][pmc]} dispatches)")


if __name__ == "__main__":
    main()
