#!/usr/bin/env python3
"""Per-kernel time inside the steady-state replays at the END of a rocprofv3 kernel trace (the timed steps of bench.py /
tools/overlap_probe.py), with the union busy time and the idle time between dispatches:
   python tools/replay_window_stats.py kt_results.db [window_ms = 100] [step_ms] [rows = 40]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
W = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 100e6
step_ms = float(sys.argv[3]) if len(sys.argv) > 3 else None
nrows = int(sys.argv[4]) if len(sys.argv) > 4 else 40
symc = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "display_name" if "display_name" in symc else ("kernel_name" if "kernel_name" in symc else "name")
rows = list(cur.execute(f"select d.start, d.end, s.{name_col} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
t_end = max(r[1] for r in rows)
lo, hi = t_end - W - 5e6, t_end - 5e6
rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
span = hi - lo
ev = sorted([(s, 1) for s, e, _ in rows] + [(e, -1) for s, e, _ in rows])
busy, depth, last = 0, 0, lo
for t, d in ev:
    if depth > 0: busy += t - last
    depth += d; last = t
ksum = sum(e - s for s, e, _ in rows)
steps = span / (step_ms * 1e6) if step_ms else 1.0
print(f"window {span/1e6:.1f} ms = {steps:.2f} steps, {len(rows)} dispatches ({len(rows)/steps:.0f}/step); busy (>= 1 kernel) "
      f"{busy/span*100:.1f} %, idle {100 - busy/span*100:.1f} % = {(span - busy)/1e6/steps:.3f} ms/step; sum of kernel durations "
      f"{ksum/1e6/steps:.3f} ms/step")
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    agg[n][0] += 1; agg[n][1] += e - s
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:nrows]:
    print(f"{n[:96]:96s} {c/steps:7.1f}/step {t/1e6/steps:8.3f} ms/step {t/c/1e3:8.1f} us avg {100.0*t/ksum:5.1f}%")
