"""What does an edge between two streams of a captured HIP graph cost at replay?  A chain of small kernels on stream A
as a graph; then the same chain with a leaf kernel handed to stream B after every k-th link (B waits for an event of
A; one join at the end).  The leaf kernels alone, appended to the chain, give the cost of their launches.

    python tools/diag_graph_edge_cost.py
"""
import time

import torch

dev = torch.device("cuda", 0)
A, B = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.ones(4096, device=dev)
y = torch.ones(4096, device=dev)
N = 120


def body(every, inline):
    ev = []
    if every and not inline:
        e = torch.cuda.Event(); e.record(A); B.wait_event(e); ev.append(e)
    for i in range(N):
        x.mul_(1.0001)
        if every and i % every == 0:
            if inline:
                y.add_(1.0)
            else:
                e = torch.cuda.Event(); e.record(A); B.wait_event(e); ev.append(e)
                with torch.cuda.stream(B):
                    y.add_(1.0)
    if every and not inline:
        e = torch.cuda.Event(); e.record(B); A.wait_event(e); ev.append(e)
    return ev


def measure(every, inline=False):
    with torch.cuda.stream(A):
        body(every, inline)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g, stream=A, capture_error_mode="thread_local"):
        keep = body(every, inline)
    for _ in range(5):
        with torch.cuda.stream(A):
            g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        with torch.cuda.stream(A):
            g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 20, g, keep


held = []
base, g, k = measure(0); held.append((g, k))
print("chain of %d kernels, one stream            %.3f ms per replay" % (N, base))
for every in (30, 10, 5, 2):
    n = len(range(0, N, every))
    t_in, g, k = measure(every, True); held.append((g, k))
    t_x, g, k = measure(every, False); held.append((g, k))
    print("%3d leaves: in line %.3f ms, on a second stream %.3f ms  ->  %.1f us per edge more than the launch" % (
        n, t_in, t_x, (t_x - t_in) * 1e3 / n))
