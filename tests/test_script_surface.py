"""The reference scripts' import lines and constructor calls work against this repository (CPU-only part: everything up
to the first kernel launch, which must fail loudly without a GPU -- there is no CPU fallback)."""
import subprocess
import sys
import textwrap

SNIPPET = textwrap.dedent('''
    import numpy as np
    import os
    from rpo.algo import RPODDPG, RPOSAC, DDPG_LA, SAC_LA
    from rpo.env import *
    from rpo.utils.logger import Logger
    from rpo.utils.monitor import get_monitor
    import gym
    import torch

    np.random.seed(123)
    torch.manual_seed(123)
    logger = Logger(("epoch", "reward", "max_ineq", "max_eq"), times=5, epochs=20000, name="cart_ddpg")
    env = gym.make("CartSafe-v0")
    assert list(env.partial_actions) == [1], env.partial_actions     # same draw as the reference after seed 123
    agent = RPODDPG(env, "./test", name="cart_ddpg", logger=logger, batch_size=256, max_steps=10, warmup=0, lr_dual=0.2,
                    corr_lr=2e-2, eps=1.0, eps_start=1.0, lr_actor=1e-4, lr_critic=3e-4, eps_epoch=20000, eval_lr=2e-2,
                    eval_steps=50, grad_eps=0.1, corr_momentum=0.0, policy_fre=4, max_epochs=20000, capacity=200,
                    shared_param=True, value_type="add", clip_thres=0.2, embed_dim=128, hidden_dim=256)
    assert agent.agent.flat.unique_numel == 67842                    # SURVEY 8c: 12 unique tensors, 67 842 floats
    assert sum(p.numel() for p in agent.agent.actor.parameters()) == 34177
    assert sum(p.numel() for p in agent.agent.critic.parameters()) == 34561
    env2 = gym.make("SpringPendulum-v0")
    sac = RPOSAC(env2, "./test", name="pen_sac", logger=logger, batch_size=256, max_steps=10, warmup=0, lr_dual=0.01,
                 corr_lr=2e-3, eps=1e-2, eps_start=1e-2, eps_epoch=20000, eval_lr=2e-3, eval_steps=50, grad_eps=0.1,
                 corr_momentum=0.0, policy_fre=4, max_epochs=20000, alpha=0.01, automatic_entropy_tuning=False,
                 capacity=200, shared_param=False, value_type="add", clip_thres=0.2, embed_dim=128, hidden_dim=256)
    env3 = gym.make("EVOPF-v0")                                      # scripts/evopf_exp.py:27-31
    assert (env3.state_dim, env3.action_dim, env3.eq_num, env3.ineq_num, env3.volatile) == (57, 43, 28, 58, True)
    assert list(env3.partial_actions) == [1, 2, 3, 4, 10, 11, 12, 15, 17, 38, 39, 40, 41, 42]
    evopf = RPODDPG(env3, "./test", name="evopf_ddpg", logger=logger, batch_size=256, max_steps=10, warmup=0,
                    lr_dual=2e-2, corr_lr=1e-4, eps=0.0001, eps_start=0.0001, eps_epoch=20000, eval_lr=1e-4,
                    eval_steps=50, grad_eps=0.02, corr_momentum=0.0, policy_fre=4, ex_action_dim=1, gamma=0.95,
                    max_epochs=40000, capacity=200, clip_thres=0.2, shared_param=False, value_type="cat")
    assert evopf.agent.actor.affines[-1].weight.shape[0] == 14
    la = DDPG_LA(env, "./test", name="cart_la", logger=logger, batch_size=256, warmup=0, lr_dual=0.2, capacity=200,
                 embed_dim=128, hidden_dim=256, clip_thres=0.2)                      # rpo/algo/ddpg_lag.py:13-21
    assert la.agent.actor.affines[-1].weight.shape[0] == 2 and la.process_action(None, 7) == 7
    if not torch.cuda.is_available():
        try:
            agent.run()
        except Exception as e:
            assert type(e).__name__ == "RpoHipError" and "no CPU fallback" in str(e), repr(e)
            print("LOUD-FAILURE-OK")
        else:
            raise SystemExit("run() must not succeed without a GPU")
    env.close()
    print("SURFACE-OK")
''')


def test_reference_script_surface():
    r = subprocess.run([sys.executable, "-c", SNIPPET], capture_output=True, text=True, cwd=".", timeout=300,
                       env={**__import__("os").environ, "PYTHONPATH": ".", "RPO_VERBOSE": "0"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert "SURFACE-OK" in r.stdout
    import torch
    if not torch.cuda.is_available():
        assert "LOUD-FAILURE-OK" in r.stdout
