"""Import-compatibility placeholders for the reference's comparison baselines (rpo/algo/ddpg_lag.py, sac_lag.py).

`rpo/algo/__init__.py:1-4` of the reference exports `DDPG_LA` and `SAC_LA` next to `RPODDPG` / `RPOSAC`; no script uses
them and they are outside the hot path this build covers (plain Lagrangian DDPG / SAC on the full action: no equation
solver, no projection).  The names exist so that `from rpo.algo import *` keeps
working; constructing one says what is missing instead of failing somewhere deep inside.
"""


class _NotBuilt(object):
    _what = "baseline"

    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "%s is a comparison baseline of the reference (rpo/algo/%s) that the MI355X hot-path build does not cover; "
            "use RPODDPG / RPOSAC, or run the baseline from the reference tree." % (type(self).__name__, self._what))


class DDPG_LA(_NotBuilt):
    _what = "ddpg_lag.py"


class SAC_LA(_NotBuilt):
    _what = "sac_lag.py"
