"""The largest intervals of a step during which NO contraction kernel runs (rocprofv3 --kernel-trace CSV): when, how long,
which light kernels run inside, which contraction ended before and which starts after. usage: exposed_intervals.py trace.csv
[marker] [min_us]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
marker = sys.argv[2] if len(sys.argv) > 2 else "sgd_momentum"
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 25.0
ends = [i for i, r in enumerate(rows) if marker in r[2]]
step_ends = [i for k, i in enumerate(ends) if k + 1 == len(ends) or ends[k + 1] - i > 5]
a, b = step_ends[-2] + 1, step_ends[-1] + 1
seg = rows[a:b]
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][:28]  # noqa: E731
heavy = lambda n: ("igemm_" in n) or ("wgrad_split" in n)  # noqa: E731
t0 = seg[0][0]
hv = sorted((s, e, n) for s, e, n, q in seg if heavy(n))
# union of heavy intervals
merged = []
for s, e, n in hv:
    if merged and s <= merged[-1][1]:
        if e > merged[-1][1]:
            merged[-1][1] = e
            merged[-1][2] = n
    else:
        merged.append([s, e, n, n])
gaps = []
prev_end, prev_name = t0, "(step start)"
for s, e, last_n, first_n in merged:
    if s - prev_end > min_us * 1e3:
        gaps.append((prev_end, s, prev_name, first_n))
    prev_end, prev_name = e, last_n
tend = max(r[1] for r in seg)
if tend - prev_end > min_us * 1e3:
    gaps.append((prev_end, tend, prev_name, "(step end)"))
tot = 0
print("step span %.3f ms; intervals without a contraction longer than %.0f us:" % ((tend - t0) / 1e6, min_us))
for g0, g1, pn, nn in gaps:
    inside = {}
    for s, e, n, q in seg:
        if not heavy(n) and e > g0 and s < g1:
            k = short(n) + "@q" + str(q)
            inside[k] = inside.get(k, 0) + (min(e, g1) - max(s, g0))
    top = sorted(inside.items(), key=lambda kv: -kv[1])[:6]
    tot += g1 - g0
    print("  t=%7.3f ms  %6.1f us  after %-26s before %-26s | %s" % ((g0 - t0) / 1e6, (g1 - g0) / 1e3, short(pn), short(nn),
                                                                      ", ".join("%s %.0f" % (k, v / 1e3) for k, v in top)))
print("sum of listed intervals: %.3f ms" % (tot / 1e6))
