#!/usr/bin/env python
"""Which kinds of nodes do the trainer's captured windows hold?  (a memset NODE replayed with a stale pattern.)
RPO_GRAPH_AUDIT=1 makes the trainer keep each captured hipGraph_t and count its nodes by kind (hipGraphGetNodes /
hipGraphNodeGetType; trainer._graph_node_kinds).  Usage:  python tools/probe/graph_nodes.py [workload ...] [--large-batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("RPO_VERBOSE", "0")
os.environ["RPO_GRAPH_AUDIT"] = "1"
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")] or ["cart_ddpg"]
    large = "--large-batch" in sys.argv
    dev = torch.device("cuda")
    rc = 0
    for w in args:
        n = 4096 if "evopf" not in w else 1024
        kw = dict(batch_size=256 * n, capacity=64) if large else {}
        tr = bench.make_trainer(n, dev, 3000, workload=w, torch_seed=123, seed=7000, use_graph=True, **kw)
        tr.vec.reset()
        tr.run_steps(96)
        torch.cuda.synchronize()
        seen = 0
        for key, e in tr._graphs.entries.items():
            if e.get("graph") is None:
                continue
            seen += 1
            kinds = e.get("node_kinds")
            print("%s%s window %r: %s" % (w, " (large batch)" if large else "", key, kinds), flush=True)
            if set(kinds) - {"kernel"}:
                rc = 1
        if not seen:
            print(w, "no graph was captured")
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
