"""How much of a step is NOT covered by a contraction kernel? From a rocprofv3 --kernel-trace CSV: per step (split at a
marker kernel) the span, the time at least one HEAVY kernel (igemm_* / wgrad_split_* : the MFMA contractions) is running,
the time only LIGHT kernels run (with the kernels that account for it), and the time nothing runs; plus busy time per
hardware queue.  usage: exposed_time.py kernel_trace.csv [marker] [step index from the end, default 1]"""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
marker = sys.argv[2] if len(sys.argv) > 2 else "sgd_momentum"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ends = [i for i, r in enumerate(rows) if marker in r[2]]
step_ends = [i for k, i in enumerate(ends) if k + 1 == len(ends) or ends[k + 1] - i > 5]
a, b = step_ends[-1 - back] + 1, step_ends[-back] + 1
seg = rows[a:b]
t0, t1 = seg[0][0], max(r[1] for r in seg)
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][:40]  # noqa: E731
heavy = lambda n: ("igemm_" in n) or ("wgrad_split" in n)  # noqa: E731
ev = []
for s, e, n, q in seg:
    ev.append((s, 1, heavy(n), n))
    ev.append((e, -1, heavy(n), n))
ev.sort(key=lambda x: (x[0], x[1]))
nh = nl = 0
last = t0
cover = dict(heavy=0, light_only=0, idle=0)
light_names = collections.Counter()
live_light = collections.Counter()
for t, d, h, n in ev:
    dt = t - last
    if dt > 0:
        if nh > 0:
            cover["heavy"] += dt
        elif nl > 0:
            cover["light_only"] += dt
            for k_, c_ in live_light.items():
                if c_ > 0:
                    light_names[short(k_)] += dt / sum(1 for v in live_light.values() if v > 0)
        else:
            cover["idle"] += dt
    last = t
    if h:
        nh += d
    else:
        nl += d
        live_light[n] += d
span = t1 - t0
print("step: %d kernels, span %.3f ms: a contraction running %.3f ms (%.0f %%), only light kernels %.3f ms (%.0f %%), nothing %.3f ms (%.0f %%)" % (
    len(seg), span / 1e6, cover["heavy"] / 1e6, 100 * cover["heavy"] / span, cover["light_only"] / 1e6,
    100 * cover["light_only"] / span, cover["idle"] / 1e6, 100 * cover["idle"] / span))
print("light-only time by kernel (ms):")
for k_, v in light_names.most_common(18):
    print("  %-42s %.3f" % (k_, v / 1e6))
qb = collections.defaultdict(int)
qn = collections.Counter()
for s, e, n, q in seg:
    qb[q] += e - s
    qn[q] += 1
print("per hardware queue: " + ", ".join("q%s %d kernels %.2f ms" % (q, qn[q], qb[q] / 1e6) for q in sorted(qb)))
hk = sum(e - s for s, e, n, q in seg if heavy(n))
lk = sum(e - s for s, e, n, q in seg if not heavy(n))
print("kernel time: contractions %.3f ms, light %.3f ms (%d launches)" % (hk / 1e6, lk / 1e6, sum(1 for r in seg if not heavy(r[2]))))
