#!/usr/bin/env python3
"""VGPR / AGPR / scratch / occupancy of every kernel of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage),
one line per kernel:  python tools/kernel_resources.py efficientvlm_amd/csrc/attention_mfma.hip [filter]"""
import re, subprocess, sys, tempfile, os
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as td:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
                        "-c", src, "-o", os.path.join(td, "x.o")] + sys.argv[3:], capture_output=True, text=True)
cur = None; rows = []
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur); continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for c in rows:
    if flt in c["name"]:
        print(f'{c["name"][:70]:70s} VGPR {c.get("VGPRs", -1):4d} AGPR {c.get("AGPRs", -1):4d} scratch {c.get("ScratchSize", -1):5d} '
              f'spillV {c.get("VGPRs Spill", -1):4d} occ {c.get("Occupancy", -1):2d} LDS {c.get("LDS Size", -1)}')
