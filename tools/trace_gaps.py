"""GPU idle gaps in a rocprofv3 --kernel-trace CSV: per step (split at sgd_momentum / given marker kernel), the busy
time, the span, and the largest gaps with the kernels on either side.  usage: trace_gaps.py kernel_trace.csv [marker]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marker = sys.argv[2] if len(sys.argv) > 2 else "sgd_momentum"
ends = [i for i, r in enumerate(rows) if marker in r[2]]
# a step ends at the last marker kernel of a burst
step_ends = [i for k, i in enumerate(ends) if k + 1 == len(ends) or ends[k + 1] - i > 5]
if len(step_ends) < 3:
    sys.exit("not enough steps")
spans = [(rows[step_ends[k + 1]][1] - rows[step_ends[k]][1]) / 1e6 for k in range(len(step_ends) - 1)]
print("step spans (ms):", " ".join("%.1f" % x for x in spans))
which = len(spans) - 1
if len(sys.argv) > 3 and sys.argv[3] == "slowest":
    which = max(range(3, len(spans)), key=lambda k: spans[k])
a, b = step_ends[which] + 1, step_ends[which + 1] + 1
seg = rows[a:b]
span = seg[-1][1] - seg[0][0]
busy, cur_end, gaps = 0, seg[0][0], []
for s, e, n in seg:
    if s > cur_end:
        gaps.append((s - cur_end, n))
        busy += e - s
        cur_end = e
    else:
        if e > cur_end:
            busy += e - cur_end
            cur_end = e
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]  # noqa
print("step %d: %d kernels, span %.2f ms, busy %.2f ms, idle %.2f ms" % (which, len(seg), span / 1e6, busy / 1e6, (span - busy) / 1e6))
hist = {}
for g, n in gaps:
    k = "<5us" if g < 5000 else "<20us" if g < 20000 else "<100us" if g < 100000 else ">=100us"
    h = hist.setdefault(k, [0, 0])
    h[0] += 1
    h[1] += g
for k, (c, t) in sorted(hist.items()):
    print("  gaps %-8s n=%4d total %.2f ms" % (k, c, t / 1e6))
prev = {id(x): None for x in seg}
names = [n for _, _, n in seg]
idx = {i: seg[i] for i in range(len(seg))}
big = sorted(((seg[i][0] - max(x[1] for x in seg[:i]), i) for i in range(1, len(seg))), reverse=True)[:15]
for g, i in big:
    print("  gap %.3f ms before #%d %s (after %s)" % (g / 1e6, i, short(seg[i][2]), short(seg[i - 1][2])))
