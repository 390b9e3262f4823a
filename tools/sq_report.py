"""sum the SQ counters of the igemm launches in a rocprofv3 --pmc counter_collection.csv"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
agg = {}
for r in rows:
    agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
wc = agg.get("SQ_WAVE_CYCLES", 1)
bc = agg.get("SQ_BUSY_CYCLES", 1)
for k, v in sorted(agg.items()):
    print("  %-28s %16.0f  %.3f of wave cycles  %.3f of busy cycles" % (k, v, v / wc, v / bc))
