#!/usr/bin/env python3
"""Durations of one kernel's launches in launch order (rocprofv3 --kernel-trace CSV)."""
import csv, glob, sys
d, name = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)) if name in r["Kernel_Name"]]
rows.sort()
print(name, len(rows), "launches; last 40 durations (us):")
print(" ".join(f"{(e - s) / 1e3:.0f}" for s, e in rows[-40:]))
