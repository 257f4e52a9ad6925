#!/usr/bin/env python3
"""Read a rocprofv3 --kernel-trace CSV: per-kernel statistics and, for the dense look-ahead
schedule, how much of each k_diag_chain launch ran beside a k_ldlt_update launch."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
st = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    st[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
tot = sum(e - s for v in st.values() for s, e in v)
print(f"{'kernel':50s} {'calls':>6s} {'avg us':>9s} {'total ms':>9s} {'%':>6s}")
for k, v in sorted(st.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    t = sum(e - s for s, e in v)
    print(f"{k[:50]:50s} {len(v):6d} {t / len(v) / 1e3:9.2f} {t / 1e6:9.3f} {100 * t / tot:6.1f}")
ch = [x for k, v in st.items() if "k_diag_chain" in k for x in v]
up = sorted(x for k, v in st.items() if k.startswith("k_ldlt_update") for x in v)
if ch and up:
    ov = 0
    for s, e in ch:
        for us, ue in up:
            lo, hi = max(s, us), min(e, ue)
            if hi > lo:
                ov += hi - lo
    print(f"k_diag_chain time overlapped by k_ldlt_update: {ov / 1e6:.3f} ms of {sum(e - s for s, e in ch) / 1e6:.3f} ms")
    # timeline of one factorisation (last 20 chain launches)
    ev = sorted([(s, e, k) for k, v in st.items() for s, e in v])
    last = sorted(ch)[-20][0] if len(ch) >= 20 else ch[0][0]
    t0 = last
    n = 0
    for s, e, k in ev:
        if s >= t0 and n < 90:
            print(f"  +{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  {k[:40]}")
            n += 1
