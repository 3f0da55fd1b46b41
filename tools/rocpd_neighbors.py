#!/usr/bin/env python3
"""what runs right before / after each dispatch of a kernel (steady-state tail of a rocprofv3 rocpd database):
   python tools/rocpd_neighbors.py db kernel_substring [tail_fraction=0.5]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = sys.argv[2]
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
symc = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "display_name" if "display_name" in symc else ("kernel_name" if "kernel_name" in symc else "name")
dcols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
qcol = "queue_id" if "queue_id" in dcols else ("stream_id" if "stream_id" in dcols else None)
rows = list(cur.execute(f"select d.start, d.end, s.{name_col}{', d.' + qcol if qcol else ''} from rocpd_kernel_dispatch d "
                        f"join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
rows = rows[int(len(rows) * (1 - frac)):]
byq = collections.defaultdict(list)
for r in rows:
    byq[r[3] if qcol else 0].append(r)
prev, nxt, n = collections.Counter(), collections.Counter(), 0
for q, rs in byq.items():
    for i, r in enumerate(rs):
        if pat in r[2]:
            n += 1
            prev[rs[i - 1][2][:70] if i else "-"] += 1
            nxt[rs[i + 1][2][:70] if i + 1 < len(rs) else "-"] += 1
print(f"{n} dispatches of *{pat}* in the last {frac:.0%} of the trace ({len(rows)} dispatches, {len(byq)} queues)")
print("-- preceded by"); [print(f"{c:6d}  {k}") for k, c in prev.most_common(14)]
print("-- followed by"); [print(f"{c:6d}  {k}") for k, c in nxt.most_common(14)]
