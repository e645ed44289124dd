"""Text timeline of the last replayed step in a rocprofv3 kernel trace: start offset, duration, queue, kernel.

    python tools/timeline.py <kernel_trace.csv> <kernels per step> [max rows]
"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("void ", "")
    return name[:60]


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
end = max(int(r["End_Timestamp"]) for r in last)
print("step span %.1f us, kernels %d, busy sum %.1f us" % ((end - t0) / 1e3, n, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / 1e3))
qs = sorted({r["Queue_Id"] for r in last})
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 10 ** 9
prev_end = {q: t0 for q in qs}
for r in last[:lim]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = qs.index(r["Queue_Id"])
    gap = (s - prev_end[r["Queue_Id"]]) / 1e3
    prev_end[r["Queue_Id"]] = e
    print("%8.1f %7.1f  q%d gap %6.1f  %s  grid %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, gap, short(r["Kernel_Name"]), r.get("Grid_Size_X", "")))
