"""Does work launched eagerly on one stream run beside a HIP graph replayed on another?  (The several-rank schedule
relies on it: the all-reduce kernels and the optimizer steps behind them are launched eagerly on the communication
stream while the calling stream replays the next device part.)  A chain of small kernels as a graph on stream A, a
few HBM-bound kernels launched eagerly on stream B; times each alone and both together.

    python tools/diag_graph_beside_eager.py
"""
import time

import torch

dev = torch.device("cuda", 0)
a, b = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.ones(4096, device=dev)
y = torch.ones(64 << 20, device=dev)          # 256 MB


def chain():
    for _ in range(300):
        x.mul_(1.0001)


NBIG = 12


def big():
    for _ in range(NBIG):
        y.mul_(1.0001)


with torch.cuda.stream(a):
    chain()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(keep_graph=True)
with torch.cuda.graph(g, stream=a):
    chain()
torch.cuda.synchronize()


def timed(fn, n=5):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def replay():
    with torch.cuda.stream(a):
        g.replay()


def eager_b():
    with torch.cuda.stream(b):
        big()


def eager_chain():
    with torch.cuda.stream(a):
        chain()


def both():
    replay()
    eager_b()


def both_rev():
    eager_b()
    replay()


def both_eager():
    eager_b()
    eager_chain()


for NBIG in (12, 48):
  print("-- %d big launches" % NBIG)
  for _ in range(2):
    both()
  print("graph alone        %.3f ms" % timed(replay))
  print("eager chain alone  %.3f ms" % timed(eager_chain))
  print("eager big alone    %.3f ms" % timed(eager_b))
  print("graph, then eager  %.3f ms" % timed(both))
  print("eager, then graph  %.3f ms" % timed(both_rev))
  print("eager + eager      %.3f ms" % timed(both_eager))
