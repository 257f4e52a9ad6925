#!/usr/bin/env python3
"""Timeline of ONE Newton step from a rocprofv3 --kernel-trace CSV: every launch between two
consecutive k_assemble_kkt launches with its start, duration and the idle gap before it; the
factorisation's own kernels are summarised in one line each."""
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""))
        for r in csv.DictReader(open(f))]
rows.sort()
asm = [i for i, r in enumerate(rows) if r[2].startswith("k_assemble_kkt")]
a, b = asm[-2], asm[-1]
t0 = rows[a][0]
prev_end = rows[a - 1][1] if a else t0
fact = ("k_chain_update", "k_diag_chain", "k_trsm_ud", "k_trsm_block", "k_update_diag", "k_ldlt_update", "k_update_jobs")
agg = {}
gap_total = 0
print(f"step = {(rows[b][0] - t0) / 1e3:.1f} us, {b - a} launches")
for s, e, n in rows[a:b]:
    gap = s - prev_end
    gap_total += max(gap, 0)
    prev_end = max(prev_end, e)
    if n.startswith(fact):
        k = n.split("<")[0]
        c = agg.setdefault(k, [0, 0, 0])
        c[0] += 1
        c[1] += e - s
        c[2] += max(gap, 0)
    else:
        print(f"{(s - t0) / 1e3:9.1f}  {(e - s) / 1e3:7.1f} us  gap {gap / 1e3:6.1f}  {n[:60]}")
for k, (c, t, g) in agg.items():
    print(f"   {k:20s} x{c:3d}  {t / 1e3:8.1f} us  gaps {g / 1e3:6.1f}")
print(f"idle gaps in the step: {gap_total / 1e3:.1f} us")
