"""Oracle for the replay buffer (reference: rpo/utils/buffer.py:3-47).  Test infrastructure only."""
import numpy as np

KEYS = ("state", "action", "next_state", "reward", "done", "eq_viol", "ineq_viol")   # agent/ddpg_pa.py:70-71


class ReplayBuffer(object):
    """The reference's single-env ring buffer: dict of arrays, one transition per ``add`` (buffer.py:22-29),
    uniform-with-replacement ``sample`` driven by the GLOBAL numpy RNG (buffer.py:31-34)."""

    def __init__(self, capacity, state_dim, action_dim, eq_num, ineq_num):
        self.capacity, self.pointer, self.size = capacity, 0, 0
        shapes = dict(state=state_dim, action=action_dim, next_state=state_dim, reward=1, done=1, eq_viol=eq_num,
                      ineq_viol=ineq_num)
        self.buffer = {k: np.zeros((capacity, shapes[k]), dtype=bool if k == "done" else np.float32) for k in KEYS}

    def __len__(self):
        return self.size

    def add(self, **kw):
        for k in KEYS:
            self.buffer[k][self.pointer] = kw[k]
        self.pointer = (self.pointer + 1) % self.capacity
        self.size = min(self.size + 1, self.capacity)

    def sample(self, num, index=None):
        if index is None:
            index = np.random.randint(0, self.size, size=num)
        return {k: self.buffer[k][index] for k in KEYS}


class VectorRing(object):
    """Row-ring semantics of the HIP path (include/rpo_hip.h): vector step t writes the N transition rows
    (t % cap_steps) * N + lane; valid rows = min(t, cap_steps) * N.  Each env's own history is exactly a reference
    ReplayBuffer of capacity cap_steps (`capacity` is per env)."""

    def __init__(self, cap_steps, n_envs, row_floats):
        self.cap_steps, self.n_envs = cap_steps, n_envs
        self.rows = np.zeros((cap_steps * n_envs, row_floats), dtype=np.float32)
        self.t = 0

    def add(self, rows):
        base = (self.t % self.cap_steps) * self.n_envs
        self.rows[base: base + self.n_envs] = rows
        self.t += 1

    @property
    def n_valid(self):
        return min(self.t, self.cap_steps) * self.n_envs
