"""Mean day of the reference's demand / price series (data/demand.py:13-17, data/price.py:13-17) -> the 2 x 24 floats
kept as constants in oracle/evopf.py and rpo_amd/env/electrical_grid/data.py.  Build container only.

    python tests/golden/make_evopf_curves.py
"""
import pickle

import numpy as np

ROOT = "/root/reference/rpo/env/electrical_grid/data/crawlers/"


def curve(series, T=24):
    x = np.array(series)
    total = len(x) // T
    c = np.zeros(T)
    for i in range(total):
        c += x[i * T:i * T + T]
    return c / total, total


if __name__ == "__main__":
    np.set_printoptions(precision=12, linewidth=120)
    with open(ROOT + "demand.pickle", "rb") as f:
        c, n = curve(pickle.load(f)["value"])
    print("CURVE_DEMAND (%d days)" % n, repr(c))
    with open(ROOT + "price.pickle", "rb") as f:
        c, n = curve(pickle.load(f)["SYS"])
    print("CURVE_PRICE (%d days)" % n, repr(c))
