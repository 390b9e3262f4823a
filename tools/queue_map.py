"""Which hardware queue did each role of the forward land on in THIS process, and how long were its steps?
From a rocprofv3 --kernel-trace CSV of `bench.py --launch eager`: steps end at `rcnn_loss_b_kernel`; per step the span from
its first kernel's start to its last kernel's end (median over the steps), the kernels per queue, and the queue of a few
landmark launches (the two stem convs = query / support trunk, RoIAlign = caller's stream, the NMS scan = proposal layer,
anchor targets = the targets stream). Used by tools/populations_probe.sh (profiles/r5_step_time_populations.txt).
usage: queue_map.py kernel_trace.csv"""
import collections
import csv
import statistics
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
ends = [i for i, r in enumerate(rows) if "rcnn_loss_b_kernel" in r[2]]
if len(ends) < 4:
    sys.exit("not enough steps in the trace")
spans, per_q, marks = [], collections.Counter(), collections.defaultdict(collections.Counter)
LAND = {"stem": "igemm_split_kernel<128, 64, 1", "roi_align": "roi_align_fwd", "nms_scan": "nms_scan", "anchor": "anchor_target",
        "softmax": "attn_softmax_unary_kernel", "maxpool": "maxpool3x3s2"}
for a, b in zip(ends[1:-1], ends[2:]):  # (skip the first steps: allocator warm-up)
    seg = rows[a + 1:b + 1]
    spans.append((max(e for _, e, _, _ in seg) - seg[0][0]) / 1e6)
    for s, e, n, q in seg:
        per_q[q] += 1
        for k, pat in LAND.items():
            if pat in n:
                marks[k][q] += 1
n = len(spans)
print("steps %d  span median %.3f ms (min %.3f max %.3f)  kernels per queue per step: %s" % (
    n, statistics.median(spans), min(spans), max(spans), " ".join("q%s:%.0f" % (q, c / n) for q, c in sorted(per_q.items()))))
print("   landmarks: " + "  ".join("%s=%s" % (k, "+".join("q%s(%.0f)" % (q, c / n) for q, c in sorted(v.items()))) for k, v in sorted(marks.items())))
