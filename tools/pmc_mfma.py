#!/usr/bin/env python3
"""Per-kernel MFMA-pipe occupancy from one rocprofv3 --pmc pass (rocpd database):
     python tools/pmc_mfma.py pass.db out.json
Counters: SQ_VALU_MFMA_BUSY_CYCLES (cycles a SIMD's matrix pipe is busy, summed over the chip's 1024 SIMDs) and
GRBM_GUI_ACTIVE (cycles the graphics pipe is active, summed over the 8 XCDs).  busy fraction = MFMA_BUSY / (GUI_ACTIVE / 8 x
1024): the share of SIMD-cycles of the kernel's lifetime in which the matrix pipe was occupied."""
import sqlite3, sys, json, collections

db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
symc = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "display_name" if "display_name" in symc else "kernel_name"
q = f"""select s.{name_col}, p.name, d.id, sum(e.value) from rocpd_pmc_event e
        join rocpd_info_pmc p on e.pmc_id = p.id
        join rocpd_kernel_dispatch d on e.event_id = d.event_id
        join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by d.id, p.name"""
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for name, ctr, did, val in cur.execute(q):
    agg[name][ctr] += val
    if ctr == "GRBM_GUI_ACTIVE":
        cnt[name] += 1
out = {}
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
    gui, busy = v.get("GRBM_GUI_ACTIVE", 0.0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if gui <= 0:
        continue
    out[k] = {"launches": cnt[k], "mfma_busy_cycles_per_launch": round(busy / max(cnt[k], 1)),
              "gui_active_cycles_per_launch_per_xcd": round(gui / 8 / max(cnt[k], 1)),
              "mfma_busy_fraction": round(busy / (gui / 8 * 1024), 4)}
json.dump({"note": __doc__, "kernels": out}, open(sys.argv[2], "w"), indent=1)
for k, v in list(out.items())[:10]:
    print(f"{k[:64]:64s} n={v['launches']:5d} busy/launch {v['mfma_busy_cycles_per_launch']:12d} gui/xcd {v['gui_active_cycles_per_launch_per_xcd']:9d} frac {v['mfma_busy_fraction']:.3f}")
