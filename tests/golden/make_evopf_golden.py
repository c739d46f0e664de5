"""EVOPF-v0 cross-check fixtures: the REFERENCE's rpo/env/electrical_grid/evopf.py run in this container on top of the
data-only pypower stand-in of ref_harness (IEEE-14 tables and makeYbus restated in oracle/evopf.py).

    python tests/golden/make_evopf_golden.py          # build container only

What this pins: the restatement of evopf.py's own logic (index bookkeeping, eq_resid / eq_jac / ineq_resid,
ineq_partial_grad, PFFunction forward + backward, Battery, step, update) and of rpo_ddpg.py's process_action on it.
What it cannot pin: pypower's case data and Ybus -- EVOPF parity stays "unpinned" (oracle/evopf.py header).
Every random draw the reference takes from a global RNG is replaced by explicit inputs stored in the fixture.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_harness  # noqa: E402

torch.set_num_threads(1)
torch.manual_seed(77)
torch.set_default_dtype(torch.float32)                       # scripts/evopf_exp.py:13
REF = ref_harness.load_reference_evopf()
from oracle import evopf as oe  # noqa: E402  (only its Philox episode data, injected into the reference's loaders)
RNG = np.random.RandomState(14)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print("wrote %s (%d arrays, %.1f KB)" % (name, len(arrays), os.path.getsize(path) / 1024))


def t32(x):
    return torch.tensor(np.asarray(x), dtype=torch.float32)


def states(n, seed=5):
    """Observations [n, 57] of random hours of random episodes (build's Philox episode data), random state of charge."""
    ids = np.arange(n)
    hours = RNG.randint(0, 24, size=n)
    s = np.concatenate([oe.episode_demand(seed, ids, 3, hours), RNG.uniform(0.1, 0.8, size=(n, 5)),
                        oe.episode_price(seed, ids, 3, hours)], axis=1)
    s[0, 28:33] = [0.1, 0.8, 0.2, 0.75, 0.15]              # bounds of the battery box active
    return s.astype(np.float32)


def partials(env, s):
    low, high = env.update(t32(s))
    u = RNG.uniform(0.05, 0.95, size=low.shape)
    ap = low + u * (high - low)
    ap[:, 4:9] = RNG.uniform(1.0, 1.06, size=(s.shape[0], 5))     # generator voltages near nominal
    ap[:, :4] = RNG.uniform(0.0, 0.6, size=(s.shape[0], 4))
    return ap.astype(np.float32)


def gen_env():
    np.random.seed(0)
    env = REF.EVOPFEnv()
    n = 12
    S = states(n)
    AP = partials(env, S)
    low, high = env.update(t32(S))
    # equation solver, one row at a time (== the reference's rollout calls), with its backward
    A, DZ, DY = [], [], RNG.randn(n, 43).astype(np.float32)
    for r in range(n):
        ap = t32(AP[r:r + 1]).requires_grad_(True)
        a = env.complete_partial(t32(S[r:r + 1]), ap)
        a.backward(t32(DY[r:r + 1]))
        A.append(a.detach().numpy()[0])
        DZ.append(ap.grad.numpy()[0])
    A = np.array(A)
    a_batch = env.complete_partial(t32(S), t32(AP)).detach().numpy()     # batch call: batch-wide Newton stop test
    # constraint functions on perturbed (infeasible) actions
    AX = (A + 0.02 * RNG.randn(*A.shape)).astype(np.float32)
    AX[:, 38:] = np.clip(AX[:, 38:] * 3, -0.3, 0.3)
    eq = env.eq_resid(t32(S), t32(AX)).numpy()
    ineq = env.ineq_resid(t32(S), t32(AX)).numpy()
    jac = env.eq_jac(t32(AX)).numpy()
    ipg = env.ineq_partial_grad(t32(S), t32(AX)).numpy()
    ipg_a = env.ineq_partial_grad(t32(S), t32(A)).numpy()
    # the rest of the constraint API (evopf.py:564-594,663-707)
    extra = dict(eq_grad=env.eq_grad(t32(S), t32(AX)).numpy(), ineq_grad=env.ineq_grad(t32(S), t32(AX)).numpy(),
                 ineq_grad_eps=env.ineq_grad(t32(S), t32(AX), 0.02).numpy(),
                 ineq_grad_new=env.ineq_grad_new(t32(S), t32(AX)).numpy(), ineq_jac=env.ineq_jac(t32(S), t32(AX)).numpy(),
                 ineq_dist_np=env.ineq_dist_np(S[3], AX[3]), eq_resid_np=env.eq_resid_np(S[3], AX[3]))
    save("evopf_env", **extra, Yr=env.Ybusr.numpy(), Yi=env.Ybusi.numpy(), action_low=env.action_space.low,
         action_high=env.action_space.high, partial_actions=env.partial_actions, partial_vars=env.partial_vars,
         other_vars=env.other_vars, S=S, AP=AP, box_low=low, box_high=high, A=A, A_batch=a_batch, DY=DY, DZ=np.array(DZ),
         AX=AX, eq_resid=eq, ineq_resid=ineq, eq_jac=jac, ineq_partial_grad=ipg, ineq_partial_grad_feasible=ipg_a,
         obj=env.obj_fn(AX).numpy())


def gen_step():
    """env.step with the loaders' caches injected (demand.py:35-44, price.py:29-38 draw them from np.random)."""
    np.random.seed(1)
    env = REF.EVOPFEnv()
    seed, n = 9, 10
    ids = np.arange(n)
    out = dict(seed=seed, hour=[], soc=[], action=[], state=[], next_state=[], reward=[], done=[], eq_viol=[], ineq_viol=[])
    for r in range(n):
        hour = [0, 5, 22, 23, 11, 1, 17, 23, 8, 3][r]
        env.reset()
        env.data.cache = np.stack([oe.episode_demand(seed, ids[r:r + 1], 2, h)[0] for h in range(24)])
        day = np.concatenate([oe.episode_price(seed, ids[r:r + 1], 2, 0)[0], np.zeros(24)]) * 100.0   # before /= genbase
        env.p_data.cache = day
        env.data.counter = env.p_data.counter = hour + 1
        soc = RNG.uniform(0.1, 0.8, size=5)
        if r == 3:
            soc[:] = [0.1, 0.8, 0.79, 0.12, 0.45]
        state = np.concatenate([env.data.cache[hour], soc, day[hour:hour + 24] / 100.0])
        env.state = state.copy()
        env.evs.state = state[28:].copy()
        ap = partials(env, state[None].astype(np.float32))
        a = env.complete_partial(t32(state[None]), t32(ap)).numpy()[0]
        a[38:] += RNG.uniform(-0.15, 0.15, size=5).astype(np.float32)       # exercise the battery clip
        nxt, reward, done, info = env.step(a.copy())
        for k, v in dict(hour=hour, soc=soc, action=a, state=state, next_state=nxt, reward=float(np.asarray(reward).reshape(-1)[0]),
                         done=done, eq_viol=info["eq_viol"][0], ineq_viol=info["ineq_viol"][0]).items():
            out[k].append(v)
    save("evopf_step", **out)


def gen_project():
    """process_action of RPODDPG (rpo_ddpg.py:72-77,266-305) with the hyper-parameters of scripts/evopf_exp.py:29-31,
    one row at a time; corr_lr is also raised so that the projection visibly moves the action."""
    np.random.seed(2)
    env = REF.EVOPFEnv()
    out = {}
    S = states(8, seed=21)
    AP = partials(env, S)
    low, high = env.update(t32(S))
    AP[:, 9:] = (high[:, 9:] + RNG.uniform(0.0, 0.1, size=(8, 5))).astype(np.float32)   # violates the charge bound
    AP[:4, 4:9] = 1.07                                                                   # violates vmax
    for tag, lr in (("script", 1e-4), ("large", 5e-4)):
        logger = REF.Logger(("epoch", "reward", "max_ineq", "max_eq"), times=1, epochs=10, name="x")
        tr = REF.RPODDPG(env, "/tmp/rpo_evopf_golden", name="x", logger=logger, batch_size=256, max_steps=10, warmup=0,
                         lr_dual=2e-2, corr_lr=lr, eps=0.0001, eps_start=0.0001, eps_epoch=20000, eval_lr=lr,
                         eval_steps=50, grad_eps=0.02, corr_momentum=0.0, policy_fre=4, ex_action_dim=1, gamma=0.95,
                         max_epochs=10, capacity=100, clip_thres=0.2, shared_param=False, value_type="cat",
                         device=torch.device("cpu"))
        train, evala, iters = [], [], []
        with torch.no_grad():
            for r in range(S.shape[0]):
                train.append(tr.process_action(t32(S[r:r + 1]), t32(AP[r:r + 1])).numpy()[0])
                a, k = tr.process_action(t32(S[r:r + 1]), t32(AP[r:r + 1]), train=False)
                evala.append(a.numpy()[0])
                iters.append(k)
        out.update({tag + "_lr": lr, tag + "_train": np.array(train), tag + "_eval": np.array(evala),
                    tag + "_eval_iters": np.array(iters)})
    save("evopf_project", S=S, AP=AP, **out)


EVOPF_HP = dict(batch_size=256, max_steps=10, warmup=0, lr_dual=2e-2, corr_lr=1e-4, eps=0.0001, eps_start=0.0001,
                eps_epoch=20000, eval_lr=1e-4, eval_steps=50, grad_eps=0.02, corr_momentum=0.0, policy_fre=4,
                ex_action_dim=1, gamma=0.95, clip_thres=0.2, shared_param=False, value_type="cat")   # scripts/evopf_exp.py:29-31


EVOPF_SAC_HP = dict({k: v for k, v in EVOPF_HP.items() if k != "gamma"}, grad_eps=0.1, alpha=0.001,
                    automatic_entropy_tuning=False, fixed=False)                                # scripts/evopf_exp_sac.py:29-32


def gen_train_steps(width=64, sub=1, algo="ddpg"):
    """RPODDPG.train / RPOSAC.train (rpo_ddpg.py:163-205, rpo_sac.py:167-219) for t = 1..4 on EVOPF with the scripts'
    hyper-parameters, every random draw recorded; init_nju > 0 so that the Lagrangian term has a gradient.  width = 64:
    small networks, stored in full; width = 256 (the scripts' sizes, what the MLP kernels support): parameters stored as
    every `sub`-th element."""
    import torch.distributions.normal as tdn
    torch.manual_seed(123)
    np.random.seed(111)
    env = REF.EVOPFEnv()
    logger = REF.Logger(("epoch", "reward", "max_ineq", "max_eq"), times=1, epochs=10, name="x")
    cls, hp = (REF.RPODDPG, EVOPF_HP) if algo == "ddpg" else (REF.RPOSAC, EVOPF_SAC_HP)
    tr = cls(env, "/tmp/rpo_evopf_golden", name="g", logger=logger, max_epochs=10, capacity=512, embed_dim=width,
             hidden_dim=width, init_nju=0.1, device=torch.device("cpu"), **hp)
    pack = lambda v: v.numpy().reshape(-1)[::sub].copy() if sub > 1 else v.numpy().copy()   # noqa: E731
    out = {"actor0." + k: pack(v) for k, v in tr.agent.actor.state_dict().items()}
    out.update({"critic0." + k: pack(v) for k, v in tr.agent.critic.state_dict().items()})
    out["sub"] = sub
    trans = {k: [] for k in ("state", "action", "next_state", "reward", "done", "eq_viol", "ineq_viol")}
    s = env.reset()
    with torch.no_grad():
        for i in range(300):
            ap = partials(env, s[None].astype(np.float32))
            a = tr.process_action(t32(s[None]), t32(ap)).numpy()[0]
            s2, r, d, info = env.step(a)
            vals = (s, a, s2, float(np.asarray(r).reshape(-1)[0]), d, info["eq_viol"].reshape(-1), info["ineq_viol"].reshape(-1))
            for k, v in zip(trans, vals):
                trans[k].append(np.asarray(v, dtype=np.float32))
            tr.agent.add(*vals)
            s = env.reset() if d else s2
    out.update({"buf." + k: np.stack(v) for k, v in trans.items()})
    draws = {"idx": [], "noise": []}
    orig_randint, orig_randn_like, orig_std_normal = np.random.randint, torch.randn_like, tdn._standard_normal

    def rec_std_normal(shape, dtype, device):
        v = orig_std_normal(shape, dtype, device)
        draws["noise"].append(v.numpy().copy())
        return v

    def rec_randint(*a, **k):
        v = orig_randint(*a, **k)
        draws["idx"].append(np.asarray(v).copy())
        return v

    def rec_randn_like(x, *a, **k):
        v = orig_randn_like(x, *a, **k)
        draws["noise"].append(v.numpy().copy())
        return v

    losses = {"critic": [], "actor": []}
    oc, oa = tr.critic_loss, tr.actor_loss

    def rec_c(*a, **k):
        v = oc(*a, **k)
        losses["critic"].append(float(v))
        return v

    def rec_a(*a, **k):
        v = oa(*a, **k)
        losses["actor"].append(float(v[0] if isinstance(v, tuple) else v))
        return v
    tr.critic_loss, tr.actor_loss = rec_c, rec_a
    np.random.randint, torch.randn_like, tdn._standard_normal = rec_randint, rec_randn_like, rec_std_normal
    try:
        for t in range(1, 5):
            tr.train(t)
            if t in (1, 4):
                out.update({"critic%d.%s" % (t, k): pack(v) for k, v in tr.agent.critic.state_dict().items()})
    finally:
        np.random.randint, torch.randn_like, tdn._standard_normal = orig_randint, orig_randn_like, orig_std_normal
    out.update({"actor4." + k: pack(v) for k, v in tr.agent.actor.state_dict().items()})
    out.update({"critic_target4." + k: pack(v) for k, v in tr.agent.critic_target.state_dict().items()})
    if algo == "ddpg":
        out.update({"actor_target4." + k: pack(v) for k, v in tr.agent.actor_target.state_dict().items()})
    out["nju4"] = tr.agent.nju.weight.detach().numpy().copy()
    out["idx"] = np.stack(draws["idx"])
    for i, z in enumerate(draws["noise"]):
        out["noise%d" % i] = z
    out["n_noise"] = len(draws["noise"])
    out["critic_losses"] = np.array(losses["critic"])
    out["actor_losses"] = np.array(losses["actor"])
    save("train_steps_%s_evopf" % algo + ("" if width == 64 else str(width)), **out)


def _stats_run(job):
    """One reference training run on EVOPF (worker of gen_training_stats)."""
    import contextlib
    import io
    algo, seed, steps = job
    torch.set_num_threads(1)
    np.random.seed(111 + seed)
    torch.manual_seed(123 + seed)
    env = REF.EVOPFEnv()
    logger = REF.Logger(("epoch", "reward", "max_ineq", "max_eq"), times=1, epochs=steps, name="x")
    cls, hp = (REF.RPODDPG, EVOPF_HP) if algo == "ddpg" else (REF.RPOSAC, EVOPF_SAC_HP)
    tr = cls(env, "/tmp/rpo_evopf_golden", name="x", logger=logger, max_epochs=steps, capacity=20000,
             device=torch.device("cpu"), **hp)
    with contextlib.redirect_stdout(io.StringIO()):
        tr.run(eval=False)
    n = logger.pointer
    mi, me, rw = [logger.tracker[k][:n] for k in ("max_ineq", "max_eq", "reward")]
    viol = np.maximum(mi, me) > 1e-3
    row = [n, viol.mean(), mi.mean(), me.mean(), me.max(), rw.mean(), rw[n // 2:].mean()]
    print("seed", seed, row, flush=True)
    return row


def gen_training_stats(steps=960, n_seeds=24, algo="ddpg"):
    """Statistics of short training runs of the reference on EVOPF (scripts/evopf_exp{,_sac}.py hyper-parameters; ~0.22 s per
    step on one core here): per seed [logged steps, violation rate (max(max_ineq, max_eq) > 1e-3), mean max_ineq, mean
    max_eq, max max_eq, mean episodic return, the same over the second half].  Seeds already recorded in the fixture are
    kept (seed s is a function of s alone), the missing ones up to `n_seeds` (RPO_STATS_SEEDS) are farmed over
    RPO_STATS_WORKERS processes."""
    import multiprocessing as mp
    n_seeds = int(os.environ.get("RPO_STATS_SEEDS", n_seeds))
    workers = int(os.environ.get("RPO_STATS_WORKERS", "6"))
    path = os.path.join(HERE, "training_stats_%s_evopf.npz" % algo)
    have = np.zeros((0, 7))
    if os.path.exists(path) and int(np.load(path)["steps"]) == steps:
        have = np.load(path)["stats"]
    jobs = [(algo, seed, steps) for seed in range(len(have), n_seeds)]
    with mp.get_context("fork").Pool(workers) as pool:
        rows = pool.map(_stats_run, jobs, chunksize=1)
    rows = np.concatenate([have, np.array(rows).reshape(-1, 7)])
    print(algo, "evopf mean", rows.mean(0), "se", rows.std(0) / np.sqrt(len(rows)))
    save("training_stats_%s_evopf" % algo, stats=rows, steps=steps,
         columns=["logged", "viol_rate", "mean_max_ineq", "mean_max_eq", "max_max_eq", "mean_return", "mean_return_2nd_half"])


if __name__ == "__main__":
    if sys.argv[1:2] == ["stats"]:                             # ~3.5 core-minutes per seed, generated on request only
        gen_training_stats(algo=(sys.argv[2:] or ["ddpg"])[0])
        sys.exit(0)
    if sys.argv[1:] == ["sac"]:
        gen_train_steps(width=256, sub=8, algo="sac")
        sys.exit(0)
    gen_env()
    if sys.argv[1:] == ["env"]:
        sys.exit(0)
    gen_step()
    gen_project()
    gen_train_steps()
    gen_train_steps(width=256, sub=8)
    gen_train_steps(width=256, sub=8, algo="sac")
