"""Summarise rocprofv3 --pmc passes per kernel (average counter value per launch) and write
profiles/pmc_traffic.json for the dominant kernel.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE reads half
of the bytes of wide coalesced streaming reads (MI355X_MICROARCH.md, HBM section), so the
read side is doubled; other access widths are uncalibrated, which makes the figure an
estimate of the memory-side traffic, not an exact byte count.
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(dirname, counter):
    f = glob.glob(dirname + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    seen = set()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        key = (r["Dispatch_Id"], r["Counter_Name"])
        name = r["Kernel_Name"].split("(")[0]
        acc[name][0] += float(r["Counter_Value"])
        if key not in seen:
            acc[name][1] += 1
            seen.add(key)
    return {k: (v[0] / max(v[1], 1), v[1]) for k, v in acc.items()}


if __name__ == "__main__":
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
        fk, n = fetch[k]
        wk = write.get(k, (0.0, 0))[0]
        hbm = (2.0 * fk + wk) * 1024.0
        out[k] = dict(launches=n, fetch_kib_raw=fk, write_kib=wk, hbm_bytes_per_launch=hbm)
        print(f"{k[:48]:48s} launches={n:5d} FETCH_SIZE={fk:12.1f} KiB (x2) WRITE_SIZE={wk:12.1f} KiB  -> {hbm/1e6:10.2f} MB/launch")
    if len(sys.argv) > 3:
        rec = {}
        for k in out:
            if "ldlt_update" not in k and "k_update_jobs" not in k and "chain_update" not in k:
                continue
            key = ("kb_chain_update" if "kb_chain_update" in k else
                   "k_chain_update" if "k_chain_update" in k else
                   "kb_ldlt_update" if k.startswith("kb_") else
                   "k_update_jobs" if "k_update_jobs" in k else "k_ldlt_update")
            if key in rec:  # (template instances of one kernel: keep the one with more launches)
                if rec[key]["launches"] >= out[k]["launches"]:
                    continue
            rec[key] = dict(out[k], kernel=k,
                            note="(2*FETCH_SIZE + WRITE_SIZE) KiB averaged over all launches; "
                                 "gfx950 FETCH_SIZE x2 correction; estimate")
        json.dump(rec, open(sys.argv[3], "w"), indent=1)
