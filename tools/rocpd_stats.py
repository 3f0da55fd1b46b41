#!/usr/bin/env python3
"""kernel stats (calls, total / average duration) from a rocprofv3 rocpd database: python tools/rocpd_stats.py db [steps]"""
import sqlite3, sys, csv
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
out = sys.argv[3] if len(sys.argv) > 3 else None
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
symc = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "display_name" if "display_name" in symc else ("kernel_name" if "kernel_name" in symc else "name")
q = f"""select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
        from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.{name_col} order by 3 desc"""
rows = list(cur.execute(q))
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot/1e6:.3f} ms over {steps:g} steps = {tot/1e6/steps:.3f} ms/step; {sum(r[1] for r in rows)/steps:.0f} launches/step")
w = csv.writer(open(out, "w")) if out else None
if w: w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for n, c, t, mn, mx in rows:
    if w: w.writerow([n, c, t, t / c, 100.0 * t / tot, mn, mx])
for n, c, t, mn, mx in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 45]:
    print(f"{n[:100]:100s} {c/steps:7.1f}/step {t/1e6/steps:8.3f} ms/step {t/c/1e3:8.1f} us avg {100.0*t/tot:5.1f}%")
