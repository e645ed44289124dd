"""How full is the GPU during the replayed steps?  From a rocprofv3 kernel trace: over the last `frac` of the trace,
the time with 0, 1, 2, 3+ kernels in flight, and per kernel name the time it ran with nothing else in flight
("alone": a proxy for the critical path) next to its total time.

    python tools/busy.py <kernel_trace.csv> [frac=0.5]
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("void ", "")
    return name[:70]


rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * (1 - frac)):]
ev = []
for i, r in enumerate(rows):
    ev.append((int(r["Start_Timestamp"]), 1, i))
    ev.append((int(r["End_Timestamp"]), -1, i))
ev.sort()
active = set()
hist = defaultdict(int)
alone = defaultdict(int)
total = defaultdict(int)
for r in rows:
    total[short(r["Kernel_Name"])] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
prev = ev[0][0]
for t, d, i in ev:
    dt = t - prev
    if dt > 0:
        hist[min(len(active), 3)] += dt
        if len(active) == 1:
            alone[short(rows[next(iter(active))]["Kernel_Name"])] += dt
    prev = t
    if d == 1:
        active.add(i)
    else:
        active.discard(i)
span = ev[-1][0] - ev[0][0]
print("span %.2f ms, kernels %d" % (span / 1e6, len(rows)))
for k in sorted(hist):
    print("  %s kernels in flight: %6.2f %%" % ("3+" if k == 3 else str(k), 100.0 * hist[k] / span))
print("kernel: alone %% of span | total %% of span")
for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:25]:
    print("  %5.2f | %5.2f  %s" % (100.0 * v / span, 100.0 * total[k] / span, k))
