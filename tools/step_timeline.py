"""One step's kernel timeline from a rocprofv3 --kernel-trace CSV (default multi-stream forward): the rows of the last
complete step (steps end at `marker`), time-ordered with start offset, duration, queue, and a per-phase summary.
usage: step_timeline.py kernel_trace.csv out.csv [marker]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"),
                     r.get("Workgroup_Size", ""), r.get("Grid_Size", "")))
rows.sort()
marker = sys.argv[3] if len(sys.argv) > 3 else "rcnn_loss_b_kernel"
ends = [i for i, r in enumerate(rows) if marker in r[2]]
if len(ends) < 3:
    sys.exit("not enough steps")
a, b = ends[-2] + 1, ends[-1] + 1
seg = rows[a:b]
t0 = seg[0][0]
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]  # noqa
with open(sys.argv[2], "w") as fh:
    fh.write("start_us,dur_us,queue,grid,wg,kernel\n")
    for s, e, n, q, wg, g in seg:
        fh.write("%.1f,%.1f,%s,%s,%s,%s\n" % ((s - t0) / 1e3, (e - s) / 1e3, q, g, wg, short(n)))
span = seg[-1][1] - t0
busy, cur = 0, t0
for s, e, *_ in seg:
    if s > cur:
        busy += e - s
        cur = e
    elif e > cur:
        busy += e - cur
        cur = e
print("step: %d kernels, span %.3f ms, busy %.3f ms, sum of durations %.3f ms" % (
    len(seg), span / 1e6, busy / 1e6, sum(e - s for s, e, *_ in seg) / 1e6))
