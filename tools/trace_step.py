"""Summarise the last Newton step of a rocprofv3 kernel trace (per-kernel totals, timeline)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
n_show = int(sys.argv[2]) if len(sys.argv) > 2 else 0
f = glob.glob(d + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:30], r["Stream_Id"],
             int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"])) for r in rows)
idx = [i for i, k in enumerate(ks) if "assemble" in k[2]][-2]
seg = ks[idx:]
end = [i for i, k in enumerate(seg) if "step_update" in k[2]][0]
seg = seg[: end + 1]
t0 = seg[0][0]
print("step span us", (seg[-1][1] - t0) / 1e3)
tot = collections.Counter()
cnt = collections.Counter()
for s, e, n, q, gx, gy in seg:
    tot[n] += e - s
    cnt[n] += 1
for n, v in tot.most_common(8):
    print(f"{n:32s} {cnt[n]:4d} {v/1e3:9.1f} us")
for s, e, n, q, gx, gy in seg[:n_show]:
    print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:7.1f} q={q} grid={gx}x{gy} {n}")
