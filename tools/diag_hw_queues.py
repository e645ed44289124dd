"""Which hardware queue does a HIP stream land on, and which queues do the branches of a forked HIP graph use?
Run under `rocprofv3 --kernel-trace` (the trace names the queue of every dispatch); every launch here is followed
by a device synchronisation, so the dispatches appear in the trace in program order, and each phase is announced by
a fill of a recognisable size (grid sizes are listed on stdout).

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/diag_hw_queues.py [pattern]

pattern: comma-separated indices of the raw streams to destroy before the graph is instantiated (default "0,1"), or
"auto": the two raw streams that share the null stream's hardware queue, which leaves that queue two streams short.

Environment: DIAG_GRAPH=1 capture a graph with four branches, instantiate it after the destruction and replay it from
the null stream; DIAG_NO_AFTER=1 make no further streams between the destruction and the instantiation.  "auto" with DIAG_GRAPH=1
DIAG_NO_AFTER=1 ends in a segmentation fault inside hipGraphLaunch: two of the graph's four internal streams land on
the null stream's queue, and the launch reads past the end of its stream list.
"""
import ctypes
import os
import sys

import torch

hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
x = torch.zeros(1 << 23, device=dev)
UNIT = 16384        # every launch fills a different multiple of this, so its grid size names it
marks = []


def mark(tag, n=None):
    """A fill of a unique size on the current stream, then a device synchronisation."""
    n = (len(marks) + 1) * UNIT
    x[:n].fill_(1.0)
    torch.cuda.synchronize()
    marks.append((tag, n))


def raw_stream():
    h = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(h), 1) == 0      # hipStreamNonBlocking
    return h


arg = sys.argv[1] if len(sys.argv) > 1 else "0,1"
mark("null stream", 1000)
pool = [torch.cuda.Stream(device=dev) for _ in range(6)]
if os.environ.get("DIAG_GRAPH"):        # the capture's streams exist before anything is destroyed
    cap = torch.cuda.Stream(device=dev)
    sides = [torch.cuda.Stream(device=dev) for _ in range(3)]
    pool += [cap] + sides
# model of the runtime's choice, as observed: the first four streams of a process open the four hardware queues in
# order; after that a new stream joins the queue with the fewest streams, the highest-numbered one on a tie
count = [1, 0, 0, 0]


def model_new():
    q = count.index(0) if 0 in count else max(i for i in range(4) if count[i] == min(count))
    count[q] += 1
    return q


pool_q = [model_new() for _ in pool]
raw_q = [model_new() for _ in range(8)]
if arg == "auto":      # destroy the two raw streams that share the null stream's queue: it is then two streams short
    pattern = [i for i, q in enumerate(raw_q) if q == 0][:2]
else:
    pattern = [int(t) for t in arg.split(",") if t != ""]
print("model: pool streams on queues", [q + 1 for q in pool_q], "raw streams on", [q + 1 for q in raw_q],
      "destroying raw", pattern, flush=True)
for i, s in enumerate(pool):
    with torch.cuda.stream(s):
        mark(f"torch pool stream {i}", 2000 + i)
raws = [raw_stream() for _ in range(8)]
for i, h in enumerate(raws):
    with torch.cuda.stream(torch.cuda.ExternalStream(h.value, device=dev)):
        mark(f"raw stream {i}", 3000 + i)
for i in pattern:
    assert hip.hipStreamDestroy(raws[i]) == 0
torch.cuda.synchronize()
after = [raw_stream() for _ in range(0 if os.environ.get("DIAG_NO_AFTER") else 4)]
for i, h in enumerate(after):
    with torch.cuda.stream(torch.cuda.ExternalStream(h.value, device=dev)):
        mark(f"raw stream made after destroying {pattern}: {i}", 4000 + i)
if os.environ.get("DIAG_GRAPH"):
    for i in [int(t) for t in os.environ.get("DIAG_DESTROY2", "").split(",") if t != ""]:
        assert hip.hipStreamDestroy(after[i]) == 0
    # a graph with four branches, forked from and joined to the capture's origin stream
    ev = [torch.cuda.Event() for _ in range(6)]
    g = torch.cuda.CUDAGraph(keep_graph=True)          # instantiated by g.instantiate() below
    with torch.cuda.graph(g, stream=cap):
        x[:200 * UNIT].fill_(2.0)
        for k, s in enumerate(sides):
            ev[k].record(cap)
            s.wait_event(ev[k])
        for r in range(3):
            x[:(210 + r) * UNIT].fill_(3.0)
        for k, s in enumerate(sides):
            with torch.cuda.stream(s):
                for r in range(3):
                    x[:(220 + 10 * k + r) * UNIT].fill_(3.0)
                ev[3 + k].record(s)
            cap.wait_event(ev[3 + k])
        x[:260 * UNIT].fill_(4.0)
    g.instantiate()
    torch.cuda.synchronize()
    mark("before replay on the null stream", 6000)
    for _ in range(3):
        x.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert bool((x[:260 * UNIT] == 4.0).all()) and bool((x[260 * UNIT:] == 0.0).all()), "replay wrote wrong values"
    mark("after replay", 6001)
for tag, n in marks:
    print(n // UNIT, tag)
print("graph: 200 root, 210-212 origin branch, 220+10k+r side branch k, 260 join (units of %d elements)" % UNIT)
print("done", flush=True)
