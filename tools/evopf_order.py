import sys, numpy as np, random
sys.path.insert(0, '/root/repo')
from oracle import evopf as oe
G = oe.GRID; nb = 14
adj = (np.abs(G.Yr) + np.abs(G.Yi)) > 0
S0 = np.zeros((28, 43), bool)
for i in range(nb):
    for k in range(nb):
        if adj[i, k]:
            for r in (i, nb + i): S0[r, G.vm0 + k] = True; S0[r, G.va0 + k] = True
for g, b in enumerate(G.spv): S0[b, G.pg0 + g] = True; S0[nb + b, G.qg0 + g] = True; S0[b, G.pe0 + g] = True
trivc = [G.pg0] + [G.qg0 + g for g in range(5)]
pq = set(G.pq); nonslack = [i for i in range(nb) if i != 0]
def cost(border, qfirst=False):
    P = S0.copy(); elim = set(trivc); tot = 0
    pairs = []
    for i in border:
        pr = [(i, G.va0 + i)] + ([(nb + i, G.vm0 + i)] if i in pq else [])
        if qfirst: pr = pr[::-1]
        pairs += pr
    for (r, c) in pairs:
        live = [cc for cc in range(43) if P[r, cc] and cc != c and cc not in elim]
        rows = np.where(P[:, c])[0]
        for rr in rows:
            if rr != r: P[rr, live] = True
        elim.add(c); tot += len(live)
    return tot
best = [7, 2, 1, 4, 6, 3, 9, 10, 8, 11, 5, 12, 13]; bc = cost(best); print("start", bc, cost(best, True))
random.seed(1)
cur, cc = best[:], bc
import math
T = 3.0
for it in range(6000):
    a, b = random.sample(range(13), 2)
    n = cur[:]; n[a], n[b] = n[b], n[a]
    c = cost(n)
    if c < cc or random.random() < math.exp((cc - c) / T):
        cur, cc = n, c
        if c < bc: best, bc = n[:], c; print(it, bc, best)
    T = max(0.2, T * 0.9993)
print("best", bc, best, cost(best, True))
