"""Where a multi-stream step spends its wall time, from a rocprofv3 kernel trace.

    python tools/concurrency.py <kernel_trace.csv> <skip first fraction, e.g. 0.5> <iterations in the kept part>

Over the kept part of the trace: wall time, time with no kernel running, and per kernel family the time during which
it was the only kernel on the device ("alone") and its share of device time when kernels overlap (each instant split
evenly between the kernels running)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("void ", "")
    m = re.match(r"([A-Za-z0-9_:]+)(<[^(]*>)?", name)
    base = m.group(1)
    t = m.group(2) or ""
    return base + (t if len(t) < 40 else t[:37] + "...>")


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = float(sys.argv[2])
iters = float(sys.argv[3])
t_lo = int(rows[0]["Start_Timestamp"])
t_hi = max(int(r["End_Timestamp"]) for r in rows)
cut = t_lo + (t_hi - t_lo) * skip
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
ev = []
for i, r in enumerate(rows):
    ev.append((int(r["Start_Timestamp"]), 1, i))
    ev.append((int(r["End_Timestamp"]), 0, i))
ev.sort()
running = set()
alone = defaultdict(float)
share = defaultdict(float)
idle = 0.0
hist = defaultdict(float)
prev = ev[0][0]
for t, kind, i in ev:
    dt = t - prev
    if dt > 0:
        hist[min(len(running), 6)] += dt
        if not running:
            idle += dt
        else:
            for j in running:
                share[short(rows[j]["Kernel_Name"])] += dt / len(running)
            if len(running) == 1:
                alone[short(rows[next(iter(running))]["Kernel_Name"])] += dt
    prev = t
    if kind:
        running.add(i)
    else:
        running.discard(i)
wall = ev[-1][0] - ev[0][0]
print("wall %.3f ms/iteration, idle %.3f, kernels %.1f/iteration" % (wall / iters / 1e6, idle / iters / 1e6, len(rows) / iters))
print("time by number of kernels running (ms/iteration): " + "  ".join("%d:%.3f" % (k, v / iters / 1e6) for k, v in sorted(hist.items())))
print("%-70s %10s %10s" % ("kernel", "share ms", "alone ms"))
for k, v in sorted(share.items(), key=lambda kv: -kv[1])[:30]:
    print("%-70s %10.4f %10.4f" % (k, v / iters / 1e6, alone[k] / iters / 1e6))
