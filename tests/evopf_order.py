#!/usr/bin/env python
"""How the static elimination order of the EVOPF solver (rpo_amd/csrc/evopf_dev.h: kBusOrder) was chosen and checked.
Analysis script, not collected by pytest; it lives under tests/ because it uses the oracle (test infrastructure).

    python tests/evopf_order.py            # ~1 min: order search, then accuracy of the chosen order on sampled Jacobians

1. Cost of an order = number of (broadcast, fma) pairs of the row-per-lane Gauss-Jordan on [J_other | J_partial] (28 x 43,
   the six unit pivots excluded) when only structurally non-zero columns of each pivot row are touched: symbolic elimination on
   case14's branch pattern.  Simulated annealing over the 13! bus orders (pairs: bus i's P equation <-> its angle, load buses'
   Q equation <-> their magnitude).  Minimum degree: 233; best found: 154 (several orders).
2. Accuracy: float32 Gauss-Jordan in the static order vs partial pivoting vs numpy float64, on Jacobians at solved states,
   at GRG-like and gross perturbations of them, and at Newton's start; and how the pivot-ratio acceptance test
   (min |pivot| / max |pivot| > tau) separates the cases where the static order loses accuracy.
"""
import math
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import evopf as oe  # noqa: E402

G = oe.GRID
NB = 14
ADJ = (np.abs(G.Yr) + np.abs(G.Yi)) > 0
PQ = set(int(b) for b in G.pq)
NONSLACK = [i for i in range(NB) if i not in G.slack]
TRIV_COLS = [G.pg0] + [G.qg0 + g for g in range(G.ng)]
TRIV_ROWS = [0] + [NB + int(b) for b in G.spv]
PARTIAL = [int(v) for v in G.partial_vars]
CHOSEN = [2, 7, 11, 10, 13, 4, 12, 9, 8, 6, 3, 1, 5]             # == kBusOrder


def pattern():
    s = np.zeros((28, 43), bool)
    for i in range(NB):
        for k in range(NB):
            if ADJ[i, k]:
                for r in (i, NB + i):
                    s[r, G.vm0 + k] = s[r, G.va0 + k] = True
    for g, b in enumerate(G.spv):
        s[b, G.pg0 + g] = s[NB + b, G.qg0 + g] = s[b, G.pe0 + g] = True
    return s


S0 = pattern()


def pairs_of(order):
    out = []
    for i in order:
        out.append((i, G.va0 + i))
        if i in PQ:
            out.append((NB + i, G.vm0 + i))
    return out


def cost(order):
    p, elim, total = S0.copy(), set(TRIV_COLS), 0
    for r, c in pairs_of(order):
        live = [cc for cc in range(43) if p[r, cc] and cc != c and cc not in elim]
        for rr in np.where(p[:, c])[0]:
            if rr != r:
                p[rr, live] = True
        elim.add(c)
        total += len(live)
    return total


def search(seeds=4, iters=4000):
    best, bc = None, 10 ** 9
    for seed in range(seeds):
        random.seed(seed)
        cur = NONSLACK[:]
        random.shuffle(cur)
        cc, temp = cost(cur), 3.0
        for _ in range(iters):
            n = cur[:]
            a, b = random.sample(range(len(n)), 2)
            if random.random() < 0.5:
                n[a], n[b] = n[b], n[a]
            else:
                n.insert(b, n.pop(a))
            c = cost(n)
            if c < cc or random.random() < math.exp((cc - c) / temp):
                cur, cc = n, c
                if c < bc:
                    best, bc = n[:], c
            temp = max(0.15, temp * 0.999)
        print("seed", seed, "best so far", bc, best)
    return best, bc


def gj32(m, static):
    m = m.astype(np.float32).copy()
    n = 28
    used = np.zeros(n, bool)
    used[:6] = True
    piv, col_of = np.ones(n, np.float32), np.arange(n)
    for k in range(6, n):
        p = k if static else int(np.argmax(np.where(used, -1, np.abs(m[:, k]))))
        used[p], col_of[p], piv[p] = True, k, m[p, k]
        f = (-m[:, k] * (np.float32(1) / m[p, k])).astype(np.float32)
        f[p] = 0
        m[:, k + 1:] = (m[:, k + 1:] + f[:, None] * m[p, k + 1:][None]).astype(np.float32)
    d = np.zeros((n, 15), np.float32)
    d[col_of] = m[:, 28:] / piv[:, None]
    return d, float(np.abs(piv[6:]).min() / np.abs(piv[6:]).max())


def accuracy(order, n=300):
    rng = np.random.RandomState(0)
    ids = np.arange(n)
    s = np.concatenate([oe.episode_demand(5, ids, 3, rng.randint(0, 24, n)), rng.uniform(0.1, 0.8, (n, 5)),
                        oe.episode_price(5, ids, 3, rng.randint(0, 24, n))], 1)
    lo, hi = oe.partial_box(s)
    a = oe.complete_partial(s, lo + rng.uniform(0, 1, lo.shape) * (hi - lo))
    groups = [a]
    for sv, sa in ((0.03, 0.1), (0.1, 0.3), (0.2, 0.6)):
        b = a.copy()
        b[:, G.vm0:G.vm0 + NB] += rng.normal(0, sv, (n, NB))
        b[:, G.va0:G.va0 + NB] += rng.normal(0, sa, (n, NB))
        groups.append(b)
    flat = a.copy()
    flat[:, G.vm0 + G.pq] = G.vm_init[G.pq]
    flat[:, G.va0:G.va0 + NB] = G.va_init
    groups.append(flat)
    jac = oe.eq_jac(np.concatenate(groups))
    pr = pairs_of(order)
    rows, cols = TRIV_ROWS + [r for r, _ in pr], TRIV_COLS + [c for _, c in pr]
    es, ep, ratio = [], [], []
    for j in jac:
        m = j[rows][:, cols + PARTIAL]
        ref = np.linalg.solve(m[:, :28], m[:, 28:])
        sc = np.abs(ref).max()
        ds, rt = gj32(m, True)
        dp, _ = gj32(m, False)
        es.append(np.abs(ds - ref).max() / sc)
        ep.append(np.abs(dp - ref).max() / sc)
        ratio.append(rt)
    es, ep, ratio, grp = np.array(es), np.array(ep), np.array(ratio), np.repeat(np.arange(5), n)
    names = ("solved", "perturbed 0.03/0.1", "perturbed 0.1/0.3", "perturbed 0.2/0.6", "Newton start")
    for g, name in enumerate(names):
        k = grp == g
        print("%-20s static: median %.2e max %.2e | pivoted: median %.2e max %.2e | min pivot ratio %.1e" % (
            name, np.median(es[k]), es[k].max(), np.median(ep[k]), ep[k].max(), ratio[k].min()))
    for tau in (2.0 ** -6, 3e-3, 1e-3):
        ok = ratio > tau
        print("tau %.4f: fallback fraction per group %s; worst static error among accepted %.2e (x%.1f of pivoted)" % (
            tau, [round(float((~ok)[grp == g].mean()), 3) for g in range(5)], es[ok].max(), (es[ok] / ep[ok]).max()))


if __name__ == "__main__":
    print("chosen order", CHOSEN, "cost", cost(CHOSEN), "| minimum-degree order cost",
          cost([7, 2, 1, 4, 6, 3, 9, 10, 8, 11, 5, 12, 13]))
    if "--search" in sys.argv:
        print("search:", search())
    accuracy(CHOSEN)
