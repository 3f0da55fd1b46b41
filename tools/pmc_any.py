#!/usr/bin/env python3
"""Per-kernel sums of whatever counters one rocprofv3 --pmc pass collected (rocpd database), per launch:
     python tools/pmc_any.py pass.db [name-substring]
SQ_* cycle counters of gfx950 count quad-cycles summed over the chip's waves / SIMDs (MI355X_MICROARCH.md, PMC section)."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
symc = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "display_name" if "display_name" in symc else "kernel_name"
q = f"""select s.{name_col}, p.name, d.id, sum(e.value) from rocpd_pmc_event e
        join rocpd_info_pmc p on e.pmc_id = p.id
        join rocpd_kernel_dispatch d on e.event_id = d.event_id
        join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by d.id, p.name"""
agg = collections.defaultdict(lambda: collections.defaultdict(float)); ids = collections.defaultdict(set)
for name, ctr, did, val in cur.execute(q):
    if pat in name:
        agg[name][ctr] += val; ids[name].add(did)
for k, v in agg.items():
    n = len(ids[k])
    print(f"{k[:100]}  launches {n}")
    for c, x in sorted(v.items()):
        print(f"    {c:32s} {x / n:16.0f} / launch")
