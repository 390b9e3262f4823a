"""Where does the step time go? sum launches by group from a --dump-launches table."""
import sys, re, collections
g = collections.OrderedDict()
for line in open(sys.argv[1]):
    p = line.split()
    idx, kind = int(p[0]), p[1]
    us = float(p[p.index("us") - 1]); gf = float(p[p.index("GF") - 1])
    M = int(re.search(r"M=(\d+)", line).group(1))
    key = kind + (" big-M" if M >= 30000 else (" mid-M" if M >= 9000 else " small-M"))
    a = g.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += us; a[2] += gf
tot = sum(v[1] for v in g.values())
for k, (n, us, gf) in sorted(g.items(), key=lambda kv: -kv[1][1]):
    print("%-22s %3d launches %8.1f us %5.1f%% %8.1f GF %6.1f TF/s" % (k, n, us, 100 * us / tot, gf, gf / us * 1e-3 * 1e3 / 1e3 * 1e3 if False else gf / (us * 1e-6) / 1e3))
print("total %.1f us" % tot)
