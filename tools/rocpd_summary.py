#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / avg / share, and for
igemm launches the per-grid breakdown. Usage: tools/rocpd_summary.py <results.db> [skip_dispatches]"""
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select * from kernels order by start").fetchall()
    ix = {n: i for i, n in enumerate(cols)}
    name_col = "name" if "name" in ix else "kernel_name"
    agg = {}
    total = 0.0
    for r in rows:
        n = r[ix[name_col]]
        n = n.replace("(anonymous namespace)::", "")
        n = re.sub(r"\(.*$", "", n)
        n = re.sub(r"^void ", "", n)
        if "igemm" in n:
            n = re.sub(r".*igemm_f32_kernel<(\d+), (\d+), (\d+)>.*", r"igemm_f32_kernel<\1,\2,stem=\3>", n)
        d = (r[ix["end"]] - r[ix["start"]]) / 1e3
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += d
        total += d
    span = (rows[-1][ix["end"]] - rows[0][ix["start"]]) / 1e3
    print("dispatches %d, kernel time %.1f us, first-to-last span %.1f us (GPU busy %.1f%%)" % (
        len(rows), total, span, 100 * total / span))
    print("%-72s %7s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for n, (cnt, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-72s %7d %12.1f %10.2f %6.2f" % (n[:72], cnt, t, t / cnt, 100 * t / total))


if __name__ == "__main__":
    main()
