"""Condense a rocprofv3 --kernel-trace --stats run into a per-kernel table (calls, total, average, share).

    python tools/kernel_stats_summary.py <dir with *_kernel_stats.csv> <iterations in the run> [out.csv]
"""
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+)(<[^(]*>)?", name)
    base = m.group(1) if m else name
    tmpl = m.group(2) or "" if m else ""
    if len(tmpl) > 40:
        tmpl = tmpl[:37] + "...>"
    return base + tmpl


def main():
    d, iters = sys.argv[1], float(sys.argv[2])
    files = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    rows = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            c = rows.setdefault(k, [0, 0.0])
            c[0] += int(r["Calls"])
            c[1] += float(r["TotalDurationNs"])
    tot = sum(v[1] for v in rows.values())
    out = [("kernel", "calls_per_iteration", "avg_us", "ms_per_iteration", "percent")]
    for k, (n, ns) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        out.append((k, "%.2f" % (n / iters), "%.2f" % (ns / n / 1e3), "%.4f" % (ns / iters / 1e6), "%.2f" % (100 * ns / tot)))
    out.append(("TOTAL", "%.1f" % (sum(v[0] for v in rows.values()) / iters), "", "%.4f" % (tot / iters / 1e6), "100"))
    w = csv.writer(open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout)
    w.writerows(out)


if __name__ == "__main__":
    main()
