#!/usr/bin/env python3
"""dp_path_probe's simulated wire under rocprofv3 --kernel-trace: does anything run BESIDE the spin kernels that stand in for
the all-reduces?   python tools/sleep_overlap.py results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
symc = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "display_name" if "display_name" in symc else ("kernel_name" if "kernel_name" in symc else "name")
dcols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
qcol = "queue_id" if "queue_id" in dcols else "stream_id"
rows = list(cur.execute(f"select d.start, d.end, s.{name_col}, d.{qcol} from rocpd_kernel_dispatch d "
                        f"join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
sl = [r for r in rows if "spin" in r[2].lower() or "sleep" in r[2].lower()]
print(len(rows), "dispatches,", len(sl), "spin kernels; queues:", sorted({r[3] for r in rows}))
sl = sl[-18:]
t0 = sl[0][0]
import bisect
starts = [r[0] for r in rows]
for s, e, n, q in sl:
    i = bisect.bisect_left(starts, s - 3_000_000)
    other = [(max(a, s), min(b, e), nm, qq) for a, b, nm, qq in rows[i:] if a < e and b > s and not (a == s and b == e)]
    busy = sum(b - a for a, b, _, _ in other)
    print(f"spin on queue {q}: {(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us ({(e - s) / 1e3:7.1f} us); beside it {len(other):4d} dispatches, "
          f"{busy / 1e3:8.1f} us of kernel time on queues {sorted({qq for _, _, _, qq in other})}")
# which queues carry what (last 40 % of the trace)
import collections
tail = rows[int(len(rows) * 0.6):]
byq = collections.defaultdict(collections.Counter)
for s, e, n, q in tail:
    byq[q][n.split("(")[0][:48]] += 1
for q in sorted(byq):
    print(f"queue {q}: {sum(byq[q].values())} dispatches; top:", byq[q].most_common(4))
# what follows each spin run: the first non-spin dispatch after it, and on which queue
k = 0
for idx, (s, e, n, q) in enumerate(rows):
    if ("spin" in n.lower() or "sleep" in n.lower()) and idx + 1 < len(rows):
        nxt = rows[idx + 1]
        if not ("spin" in nxt[2].lower() or "sleep" in nxt[2].lower()):
            k += 1
            if k > 270:
                print(f"after spin (queue {q}, ended {e}): next dispatch starts {(nxt[0] - e) / 1e3:7.1f} us later on queue {nxt[3]}: {nxt[2][:60]}")
