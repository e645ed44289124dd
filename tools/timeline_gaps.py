"""What runs while no MFMA product is running?  From a rocprofv3 kernel trace of bench.py in its default mode (HIP
graphs, branches on side streams, lookahead): over the last `frac` of the trace (the timed steps), the span, the time
with at least one kernel active, the time with at least one gemm_* kernel active, and - for the rest - which kernels
were active and for how long (a kernel's share of GEMM-idle time is what it would return if it were hidden).

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 bench.py --steps 40 --no-roofline --no-bf16-line --no-cpu-baseline --no-api-loop
    python tools/timeline_gaps.py OUT [iterations in the window = 20]
"""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("void ", "")
    name = re.sub(r"<.*", "", name)
    return name[:48]


path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in csv.DictReader(open(path))]
rows.sort()
# the timed loop is the dense tail of the trace: find the last Adam launches and count back `iters` iterations by the
# critics' optimizer launches (2 per iteration)
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
per_iter = 3.5
first = adam[-int(per_iter * iters) - 1] + 1
win = rows[first:adam[-1] + 1]
t0, t1 = win[0][0], max(r[1] for r in win)
events = []
for s, e, n, q in win:
    g = 1 if ("gemm_" in n or "splitk" in n) else 0
    events.append((s, 1, g, n))
    events.append((e, -1, g, n))
events.sort(key=lambda x: (x[0], x[1]))
active, gactive = 0, 0
names = collections.Counter()
busy = gbusy = 0
idle_by = collections.Counter()
idle_none = 0
prev = t0
for t, d, g, n in events:
    dt = t - prev
    if dt > 0:
        if active > 0:
            busy += dt
        if gactive > 0:
            gbusy += dt
        elif active > 0:
            for k, c in names.items():
                if c > 0:
                    idle_by[k] += dt / sum(1 for c2 in names.values() if c2 > 0)
        else:
            idle_none += dt
    prev = t
    active += d
    gactive += d * g
    names[short(n)] += d
span = t1 - t0
print("window: %d kernels, span %.3f ms = %.3f ms per iteration" % (len(win), span / 1e6, span / 1e6 / iters))
print("some kernel active %.3f ms/it, a GEMM-family kernel active %.3f ms/it, nothing active %.3f ms/it" % (
    busy / 1e6 / iters, gbusy / 1e6 / iters, idle_none / 1e6 / iters))
print("time with no GEMM-family kernel active, by the kernels that were running (ms per iteration):")
for k, v in idle_by.most_common(25):
    print("  %-50s %.4f" % (k, v / 1e6 / iters))
ksum = collections.Counter()
for s, e, n, q in win:
    ksum[short(n)] += e - s
print("kernel time by family (ms per iteration):")
for k, v in ksum.most_common(12):
    print("  %-50s %.4f" % (k, v / 1e6 / iters))
