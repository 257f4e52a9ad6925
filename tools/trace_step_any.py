#!/usr/bin/env python3
"""Timeline of the LAST step of any bench workload from a rocprofv3 --kernel-trace CSV: the
launches between the last two occurrences of a marker kernel (argv[2], e.g. k_residual)."""
import csv
import glob
import sys

d, marker = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""))
        for r in csv.DictReader(open(f))]
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].startswith(marker)]
a, b = idx[-2], idx[-1]
t0 = rows[a][0]
prev_end = rows[a - 1][1] if a else t0
gaps = 0
print(f"step = {(rows[b][0] - t0) / 1e3:.1f} us, {b - a} launches")
for s, e, n in rows[a:b]:
    gap = s - prev_end
    gaps += max(gap, 0)
    prev_end = max(prev_end, e)
    print(f"{(s - t0) / 1e3:9.1f}  {(e - s) / 1e3:7.1f} us  gap {gap / 1e3:6.1f}  {n[:70]}")
print(f"idle gaps in the step: {gaps / 1e3:.1f} us")
