"""Fit the tile/split-K cost model of cg_gemm (csrc/gemm.hip) to a sweep made by tools/sweep_gemm.py.

    python tools/fit_gemm_model.py gpurun_out/sweep_all.csv [bf16]

Model (microseconds), per tile kind c and split count sp:
    slabs * max(L_c, n * S_c) + P_c * (1.5 if accumulate else 1) + (sp > 1) * (F + (sp + 1) * bytes(M*N) / BW)
with n = ceil(tiles * sp / 256) workgroups on the busiest CU and slabs = ceil(ceil(K / BK) / sp).
Prints the fitted constants and, per shape, the time of the model's pick against the best swept time."""
import csv
import sys
from collections import defaultdict

import numpy as np
from scipy.optimize import minimize

TILES = [(128, 128), (128, 64), (64, 64)]


def predict(theta, M, N, K, acc, t, sp, bk, mm=0):
    L, S, P = theta[3 * t:3 * t + 3]
    L, S = L * (1 + theta[11] * mm), S * (1 + theta[11] * mm)
    F, BW = theta[9], theta[10]
    bm, bn = TILES[t]
    tiles = -(-M // bm) * -(-N // bn)
    nk = -(-K // bk)
    n = -(-(tiles * sp) // 256)
    slabs = -(-nk // sp)
    cost = slabs * max(L, n * S) + P * (1.5 if acc else 1.0)
    if sp > 1:
        cost += F + (sp + 1) * (M * N * 4.0) / (BW * 1e6)
    return cost


def main():
    rows = []
    bk = 64 if len(sys.argv) > 2 and sys.argv[2] == "bf16" else 32
    for r in csv.reader(open(sys.argv[1])):
        name, M, N, K, ta, tb, acc, t, sp, us = r
        M, N, K, acc, t, sp, us = int(M), int(N), int(K), int(acc), int(t), int(sp), float(us)
        if sp > 1 and (-(-K // bk)) // sp < 4:
            continue
        rows.append((name, M, N, K, acc, t, sp, us, int(ta) + (1 - int(tb))))

    def loss(theta):
        e = [np.log(predict(theta, M, N, K, acc, t, sp, bk, mm) / us) for _, M, N, K, acc, t, sp, us, mm in rows]
        return float(np.mean(np.square(e)))

    x0 = np.array([2.3, 2.0, 6.0, 1.2, 1.0, 4.0, 0.6, 0.52, 3.0, 5.0, 4.0, 0.1])
    if bk == 64:
        x0 = np.array([1.2, 0.65, 6.0, 0.8, 0.45, 4.0, 0.5, 0.25, 2.5, 5.0, 4.0, 0.5])
    best = minimize(loss, x0, method="Nelder-Mead", options=dict(maxiter=40000, xatol=1e-4, fatol=1e-8))
    best = minimize(loss, best.x, method="Nelder-Mead", options=dict(maxiter=40000, xatol=1e-4, fatol=1e-8))
    th = best.x
    print("rms log error %.3f" % np.sqrt(best.fun))
    for t, (bm, bn) in enumerate(TILES):
        print("  {%d, %d, %.3f, %.3f, %.2f}," % (bm, bn, th[3 * t], th[3 * t + 1], th[3 * t + 2]))
    print("  split fixed %.2f us, partial-sum bandwidth %.2f MB/us, per m-major operand +%.0f%%" % (th[9], th[10], 100 * th[11]))
    by = defaultdict(list)
    for name, M, N, K, acc, t, sp, us, mm in rows:
        by[(name, M, N, K, acc)].append((predict(th, M, N, K, acc, t, sp, bk, mm), us, t, sp))
    tot_pick = tot_best = 0.0
    for key, lst in by.items():
        pick = min(lst)
        bst = min(lst, key=lambda v: v[1])
        tot_pick += pick[1]
        tot_best += bst[1]
        print("%-16s %5dx%5dx%5d pick t%d s%-2d %7.1fus (model %7.1f) | best t%d s%-2d %7.1fus  %+5.1f%%" %
              (key[0], key[1], key[2], key[3], pick[2], pick[3], pick[1], pick[0], bst[2], bst[3], bst[1],
               100 * (pick[1] / bst[1] - 1)))
    print("sum of picks %.1f us, sum of bests %.1f us (%.1f%% regret)" % (tot_pick, tot_best, 100 * (tot_pick / tot_best - 1)))


if __name__ == "__main__":
    main()
