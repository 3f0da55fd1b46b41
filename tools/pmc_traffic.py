#!/usr/bin/env python3
"""Per-kernel HBM traffic per launch from two rocprofv3 --pmc passes (rocpd databases):
     python tools/pmc_traffic.py fetch.db write.db out.json
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests as 64 (MI355X_MICROARCH.md, HBM), so
read bytes = 2 x FETCH_SIZE x 1024."""
import sqlite3, sys, json, collections

def per_kernel(dbpath, counter):
    db = sqlite3.connect(dbpath); cur = db.cursor()
    symc = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "display_name" if "display_name" in symc else "kernel_name"
    pmcc = [r[1] for r in cur.execute("pragma table_info(rocpd_info_pmc)")]
    q = f"""select s.{name_col}, d.id, sum(e.value) from rocpd_pmc_event e
            join rocpd_info_pmc p on e.pmc_id = p.id
            join rocpd_kernel_dispatch d on e.event_id = d.event_id
            join rocpd_info_kernel_symbol s on d.kernel_id = s.id
            where p.name = ? group by d.id"""
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, _, val in cur.execute(q, (counter,)):
        a = agg[name]; a[0] += 1; a[1] += val
    return agg

fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
import os
out = {"commit": os.environ.get("EVLM_COMMIT"), "note": "bytes per launch; fetch = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB; separate --pmc passes "
               "of `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-oracle-check --no-graph`", "kernels": {}}
for k in sorted(fetch, key=lambda k: -fetch[k][1]):
    n, f = fetch[k]; nw, w = write.get(k, (0, 0.0))
    fb = 2.0 * f * 1024 / max(n, 1); wb = w * 1024 / max(nw, 1)
    out["kernels"][k] = {"launches": n, "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                         "hbm_bytes_per_launch": round(fb + wb)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in list(out["kernels"].items())[:12]:
    print(f"{k[:70]:70s} n={v['launches']:5d} fetch {v['fetch_bytes_per_launch']/1e6:8.1f} MB write {v['write_bytes_per_launch']/1e6:8.1f} MB")
