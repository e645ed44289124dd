"""From a rocprofv3 --kernel-trace of bench.py: how busy is the GPU during the timed iterations?  Prints, for the last
`n` milliseconds of the trace, the share of wall time with 0, 1, 2, 3+ kernels in flight, and the kernels that run
alone most (the serial stretches of the step).

    python tools/trace_occupancy.py <dir with *_kernel_trace.csv> [window_ms]
"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
rows = []
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t_end = max(e for _, e, _ in rows)
t0 = t_end - int(win * 1e6)
rows = [r for r in rows if r[0] >= t0]
ev = []
for s, e, n in rows:
    ev.append((s, 1, n))
    ev.append((e, -1, n))
ev.sort()
level = 0
active = collections.Counter()
share = collections.Counter()
alone = collections.Counter()
prev = ev[0][0]
for t, dlt, n in ev:
    share[min(level, 3)] += t - prev
    if level == 1:
        alone[next(iter(k for k, v in active.items() if v > 0))] += t - prev
    prev = t
    level += dlt
    active[n] += dlt
tot = sum(share.values())
print(f"window {tot / 1e6:.2f} ms, {len(rows)} kernels; in flight: " +
      ", ".join(f"{k if k < 3 else '3+'}: {v / tot:.3f}" for k, v in sorted(share.items())))
print("time with exactly one kernel in flight, by kernel (ms):")
for n, v in alone.most_common(14):
    print(f"  {v / 1e6:7.3f}  {n[:100]}")
