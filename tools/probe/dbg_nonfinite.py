"""Debugging aid (round 6): where do non-finite values first appear in a run?  Runs the bench trainer in chunks and reports the
first chunk after which a parameter / Adam moment / ring row / the failure word is non-finite, and which tensors are affected.
    python tools/probe/dbg_nonfinite.py [large|ref] [torch_seed] [philox_seed] [chunk]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("RPO_VERBOSE", "0")
import torch  # noqa: E402
import bench  # noqa: E402
from rpo_amd import _lib  # noqa: E402
from rpo_amd.algo import NonFiniteError  # noqa: E402

NF = _lib.CONST["RPO_CTRL_NONFINITE"]
mode = sys.argv[1] if len(sys.argv) > 1 else "large"
tseed = int(sys.argv[2]) if len(sys.argv) > 2 else 123
pseed = int(sys.argv[3]) if len(sys.argv) > 3 else 7000
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 16
dev = torch.device("cuda")
steps, n = 3000, 4096
extra = dict(batch_size=256 * n) if mode == "large" else {}
tr = bench.make_trainer(n, dev, steps, capacity=steps, workload="cart_ddpg", torch_seed=tseed, seed=pseed, **extra)
tr.vec.reset()
fl = tr.agent.flat


def report(tag):
    print(tag, "t =", tr._t, "flag word", int(tr.vec.ctrl[NF]))
    for name, mod in (("actor", tr.agent.actor), ("critic", tr.agent.critic)):
        for pn, p in mod.named_parameters():
            bad = int((~torch.isfinite(p.data)).sum())
            gb = int((~torch.isfinite(p.grad)).sum()) if p.grad is not None else -1
            if bad or gb > 0:
                print("   %s.%s: %d non-finite of %d (grad: %d); max finite |p| %.3g" % (
                    name, pn, bad, p.numel(), gb, float(torch.nan_to_num(p.data, nan=0.0, posinf=0.0, neginf=0.0).abs().max())))
    print("   nu", tr.agent.nju.weight.data.view(-1).tolist())
    for k, v in tr.last_losses.items():
        try:
            print("   loss", k, float(v))
        except Exception as e:  # noqa: BLE001
            print("   loss", k, "?", e)


def where_flat():
    bad = torch.nonzero(~torch.isfinite(fl.data)).view(-1).tolist()
    names = {}
    for name, mod in (("actor", tr.agent.actor), ("critic", tr.agent.critic)):
        for pn, p in mod.named_parameters():
            names[fl.offset[id(p)]] = ("%s.%s" % (name, pn), p.numel())
    offs = sorted(names)
    print("   non-finite flat indices (%d):" % len(bad), bad[:24], "total", fl.total, "ranges c/a", fl.critic_range, fl.actor_range, fl.policy_bucket)
    for b in bad[:24]:
        o = max(x for x in offs if x <= b) if any(x <= b for x in offs) else None
        if o is not None:
            nm, num = names[o]
            print("     index %d = %s + %d (numel %d)%s" % (b, nm, b - o, num, "  <- PADDING" if b - o >= num else ""))
        else:
            print("     index %d before the first tensor" % b)
    for opt_name in ("critic_optim", "actor_optim", "nju_optim", "lamb_optim"):
        opt = getattr(tr.agent, opt_name, None)
        for attr in ("exp_avg", "exp_avg_sq"):
            t = getattr(opt, attr, None) if opt is not None else None
            if t is not None:
                print("     %s.%s non-finite: %d" % (opt_name, attr, int((~torch.isfinite(t)).sum())))
    print("     grad non-finite:", torch.nonzero(~torch.isfinite(fl.grad)).view(-1).tolist()[:16])


# padding floats of the flat buffer (behind every tensor, up to the next multiple of 4): must stay zero everywhere
pad = torch.ones(fl.total, dtype=torch.bool, device=dev)
for mod in (tr.agent.actor, tr.agent.critic, tr.agent.nju):
    for p_ in mod.parameters():
        o = fl.offset.get(id(p_))
        if o is not None:
            pad[o:o + p_.numel()] = False
pad_idx = torch.nonzero(pad).view(-1)
print("padding floats:", pad_idx.tolist())
harvest_at = int(os.environ.get("DBG_HARVEST_AT", "0"))


def pad_state(tag):
    co = tr.agent.critic_optim
    lo, hi = fl.critic_range
    items = [("data", fl.data), ("grad", fl.grad)]
    out = []
    for nm, t in items:
        v = t[pad_idx]
        if bool((v != 0).any()):
            out.append((nm, [(int(i), float(x)) for i, x in zip(pad_idx.tolist(), v.tolist()) if x != 0]))
    cpad = pad_idx[(pad_idx >= lo) & (pad_idx < hi)] - lo
    for nm, t in (("critic.exp_avg", co.exp_avg), ("critic.exp_avg_sq", co.exp_avg_sq)):
        v = t[cpad]
        if bool((v != 0).any()):
            out.append((nm, [(int(i) + lo, float(x)) for i, x in zip(cpad.tolist(), v.tolist()) if x != 0]))
    if out:
        print(tag, "t =", tr._t, "NON-ZERO padding:", out)
    return bool(out)


done = False
seen_pad = False
while tr._t < steps and not done:
    if harvest_at and tr._t == harvest_at:
        tr._harvest()
        print("manual harvest at", tr._t)
    if not seen_pad:
        seen_pad = pad_state("after chunk;")
    try:
        tr.run_steps(min(chunk, steps - tr._t))
    except NonFiniteError as e:
        print("NonFiniteError:", str(e)[:60])
        done = True
    torch.cuda.synchronize()
    if not torch.isfinite(fl.data).all() or int(tr.vec.ctrl[NF]) != 0:
        report("FIRST non-finite state after the chunk ending at")
        where_flat()
        t1 = tr._t
        rows = tr.buffer.rows[max(0, t1 - chunk) * n:t1 * n]
        bad = ~torch.isfinite(rows[:, :24]).all(dim=1)
        print("   ring rows of the chunk with non-finite entries:", int(bad.sum()))
        mx = rows[:, :24].abs().amax(dim=0)
        print("   column-wise max |value| over the chunk:", [round(float(x), 3) for x in mx.cpu()])
        done = True
if not done:
    report("clean run;")
