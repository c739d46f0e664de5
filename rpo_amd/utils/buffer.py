"""Device replay ring with the reference's ``ReplayBuffer`` interface (rpo/utils/buffer.py:3-47).

The reference keeps a dict of numpy arrays on the host and adds one transition per env step.  Here the transitions of
N vectorised envs live in HBM as one row-major ring ``rows[cap_steps * n_envs, W]`` (W = ring stride: 32 floats for CartSafe
-- 24 of transition + 8 of padding, one 128-byte line per row --, 16 for SpringPendulum; layout in include/rpo_hip.h).  The fused *_step kernels scatter a whole vector step into it and
``sample`` draws + gathers a batch in one kernel.  ``capacity`` keeps the reference's meaning per env:
each env's own history is a ring of ``capacity`` transitions, total rows = capacity * n_envs.
"""
import torch

from .. import ops as hip_ops


class ReplayBuffer(object):

    def __init__(self, capacity, n_envs, kernels, device, seed=0, ctrl=None, ops=hip_ops):
        self.capacity = int(capacity)                 # per env == number of vector steps the ring holds
        self.n_envs = int(n_envs)
        self.kernels = kernels
        self.cols = kernels.cols
        self.keys = tuple(self.cols.keys())
        self.device = device
        self.seed = int(seed)
        self._ops = ops
        # ring rows are kernels.ring_floats apart (CartSafe: 32 floats = one 128-byte line per 96-byte transition); the columns
        # of a transition (kernels.cols) are its first kernels.row_floats floats
        self.rows = torch.zeros(self.capacity * self.n_envs, getattr(kernels, "ring_floats", kernels.row_floats), device=device)
        # ctrl[0] = vector steps taken so far; owned by the step kernel, shared with the env state
        self.ctrl = ctrl if ctrl is not None else torch.zeros(hip_ops.CTRL_LEN, dtype=torch.int64, device=device)
        self._steps_host = 0                          # host mirror of ctrl[0] (no device sync needed)
        # the counter words the SAMPLER reads (valid range, Philox index): the trainers point it at their update clock
        self.sample_ctrl = self.ctrl

    # -- reference surface -------------------------------------------------------------------------------------
    def __len__(self):
        return self.size

    @property
    def size(self):
        return min(self._steps_host, self.capacity) * self.n_envs

    @property
    def pointer(self):
        return (self._steps_host % self.capacity) * self.n_envs

    def note_step(self):
        """Called by the trainer after each fused step launch (which scattered n_envs rows and advanced ctrl[0])."""
        self._steps_host += 1

    def add(self, **kwargs):
        """One transition per env, as tensors/arrays of shape [n_envs, dim] (buffer.py:22-29).  The training loop
        never calls this -- the step kernel writes the rows itself; it exists for API compatibility and tests."""
        assert set(kwargs.keys()) == set(self.keys), "error keys!"
        base = self.pointer
        for key, (lo, hi) in self.cols.items():
            v = torch.as_tensor(kwargs[key], dtype=torch.float32, device=self.device).reshape(self.n_envs, hi - lo)
            self.rows[base:base + self.n_envs, lo:hi] = v
        self._steps_host += 1
        self.ctrl[0] = self._steps_host
        if self.sample_ctrl is not self.ctrl:
            self.sample_ctrl[0] = self._steps_host

    def sample_rows(self, num, out=None, idx_out=None, salt=0):
        """Uniform-with-replacement draw + gather of ``num`` rows (buffer.py:31-34) -> [num, row_floats] device tensor (a
        transition is the first `row_floats` floats of a ring row; the ring's padding is not part of a sample)."""
        if out is None:
            out = torch.empty(num, self.kernels.row_floats, device=self.device)
        self._ops.replay_sample_gather(self.rows, self.capacity, self.n_envs, out, idx_out, self.seed, salt, self.sample_ctrl)
        return out

    def split(self, batch):
        """Column views of a gathered batch, keyed like the reference's sample dict."""
        return dict((k, batch[:, lo:hi]) for k, (lo, hi) in self.cols.items())

    def sample(self, num):
        return self.split(self.sample_rows(num))
