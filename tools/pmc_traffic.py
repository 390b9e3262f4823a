"""Sum rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs per kernel -> JSON with corrected HBM bytes per launch.
usage: pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv steps_in_run out.json
Correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane)
coalesced reads -> doubled for the kernels whose loads are b128 (the igemm / Winograd / element-wise float4 kernels);
WRITE_SIZE is taken as reported. Counter unit: KB."""
import csv
import json
import sys


def load(path, counter):
    per = {}
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter:
                continue
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            a = per.setdefault(n, [0.0, 0])
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return per


f = load(sys.argv[1], "FETCH_SIZE")
w = load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(f) | set(w)):
    fk, fc = f.get(k, [0.0, 0])
    wk, wc = w.get(k, [0.0, 0])
    calls = max(fc, wc)
    if not calls:
        continue
    out[k] = {"launches": calls, "fetch_kb_reported": fk, "write_kb_reported": wk,
              "hbm_bytes_per_launch_corrected": (2.0 * fk / max(fc, 1) + wk / max(wc, 1)) * 1024.0}
json.dump(out, open(sys.argv[4], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda x: -x[1]["hbm_bytes_per_launch_corrected"] * x[1]["launches"])[:12]:
    print("%-50s launches %5d  %.1f MB/launch (corrected)" % (k, v["launches"], v["hbm_bytes_per_launch_corrected"] / 1e6))
