#!/usr/bin/env python3
"""Per-queue busy / idle time of the replayed GD step from a rocprofv3 kernel-trace database:
   python tools/stream_timeline.py kt_results.db [n_steps]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
print("columns:", cols)
symc = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "display_name" if "display_name" in symc else ("kernel_name" if "kernel_name" in symc else "name")
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(cur.execute(f"select d.{qcol}, d.start, d.end, s.{name_col} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
# the steady-state replays: the 120 ms window holding the most dispatches
import bisect
starts = [r[1] for r in rows]
W = 120_000_000
best, bi = 0, 0
for i in range(0, len(rows), 50):
    j = bisect.bisect_right(starts, starts[i] + W)
    if j - i > best:
        best, bi = j - i, i
rows = rows[bi:bisect.bisect_right(starts, starts[bi] + W)]
span = rows[-1][2] - rows[0][1]
byq = collections.defaultdict(list)
for q, s, e, n in rows:
    byq[q].append((s, e, n))
print(f"window {span/1e6:.2f} ms, {len(rows)} dispatches")
# union busy time over all queues
ev = sorted([(s, 1) for _, s, e, _ in rows] + [(e, -1) for _, s, e, _ in rows])
busy, depth, last = 0, 0, None
conc = collections.Counter()
for t, d in ev:
    if depth > 0: busy += t - last; conc[min(depth, 3)] += t - last
    depth += d; last = t
print(f"GPU busy (>= 1 kernel running): {busy/span*100:.1f} % of the window; time with 1 / 2 / 3+ kernels in flight: "
      f"{conc[1]/span*100:.1f} / {conc[2]/span*100:.1f} / {conc[3]/span*100:.1f} %")
for q, lst in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    b = sum(e - s for s, e, _ in lst)
    gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
    small = [g for g in gaps if 0 < g < 20000]
    print(f"queue {q}: {len(lst)} dispatches, busy {b/1e6:.2f} ms ({b/span*100:.1f} %), "
          f"sum of gaps < 20 us: {sum(small)/1e6:.2f} ms ({len(small)} gaps, median {sorted(small)[len(small)//2]/1e3 if small else 0:.2f} us)")
    hist = collections.Counter(min(int(g // 1000), 20) for g in gaps if g > 0)
    print("   gap histogram (us: count): " + " ".join(f"{k}:{hist[k]}" for k in sorted(hist)))
    neg = [g for g in gaps if g <= 0]
    print(f"   back-to-back overlapped (gap <= 0): {len(neg)}")
