"""Per-kernel MFMA utilisation from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`
pass (counter_collection.csv).  SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs of the chip;
GRBM_GUI_ACTIVE is reported as the sum over the 8 XCDs (MI355X_MICROARCH.md, DVFS section), so
  utilisation = MFMA_BUSY / ((GUI_ACTIVE / 8) * 1024 SIMDs)
and effective clock = GUI_ACTIVE / 8 / kernel time (reads high for dispatches under ~0.3 ms)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[name].add(r["Dispatch_Id"])
rows = []
for k, c in acc.items():
    busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
    rows.append((busy, k, len(disp[k]), act))
print(f"{'kernel':44s} {'launches':>8s} {'MFMA_BUSY_CYCLES':>18s} {'GRBM_GUI_ACTIVE':>16s} {'mfma util':>10s}")
for busy, k, n, act in sorted(rows, reverse=True):
    util = busy / (act / 8.0 * 1024.0) if act else 0.0
    print(f"{k[:44]:44s} {n:8d} {busy:18.4e} {act:16.4e} {util:10.3f}")
