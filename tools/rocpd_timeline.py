#!/usr/bin/env python
"""Print the kernel timeline of the LAST step of a rocprofv3 trace: busy/idle intervals and a
per-phase listing. Usage: tools/rocpd_timeline.py <results.db> <steps> [min_gap_us]"""
import re
import sqlite3
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void ", "", n)
    if "igemm" in n:
        return re.sub(r".*igemm_f32_kernel<(\d+), (\d+), (\d+)>.*", r"igemm<\1,\2,\3>", n)
    n = re.sub(r"\(.*$", "", n)
    return n[:48]


def main():
    db, steps = sys.argv[1], int(sys.argv[2])
    min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    ix = {n: i for i, n in enumerate(cols)}
    rows = c.execute("select * from kernels order by start").fetchall()
    name_col = "name" if "name" in ix else "kernel_name"
    qcol = "queue_id" if "queue_id" in ix else ("stream_id" if "stream_id" in ix else None)
    # find the last step: split on nchw_to_nhwc launches with the query grid (first kernel of a step on main)
    starts = [i for i, r in enumerate(rows) if "nchw_to_nhwc" in r[ix[name_col]]]
    per = len(starts) // max(steps, 1)
    first = starts[-per] if per else 0
    rows = rows[first:]
    t0 = rows[0][ix["start"]]
    busy_end = t0
    idle = 0.0
    print("last step: %d dispatches, span %.1f us" % (len(rows), (max(r[ix["end"]] for r in rows) - t0) / 1e3))
    for r in rows:
        st, en = r[ix["start"]], r[ix["end"]]
        if st > busy_end:
            gap = (st - busy_end) / 1e3
            idle += gap
            if gap >= min_gap:
                print("   ---- idle %.1f us ----" % gap)
        busy_end = max(busy_end, en)
        d = (en - st) / 1e3
        if d >= min_gap:
            print("%9.1f  %8.1f us  q%-3s %s" % ((st - t0) / 1e3, d, r[ix[qcol]] if qcol else "", short(r[ix[name_col]])))
    print("total idle inside the step: %.1f us" % idle)


if __name__ == "__main__":
    main()
