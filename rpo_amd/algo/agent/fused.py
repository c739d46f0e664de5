"""Descriptors and scratch for running the agents' networks through the hand-written MLP kernels
(rpo_mlp_forward / rpo_mlp_backward, exact f32 MFMA) instead of the rocBLAS + elementwise chain of the torch modules.

The torch modules stay the owners of the parameters (state_dict, checkpoints, the `agent.actor(...)` API); the kernels
read the same flat buffers through pointers.  Unsupported shapes (hidden_layer != 1, embed_layer != 1, partial critics,
sizes other than those of ``rpo_mlp_supported``) return ``None`` from ``build`` and the trainers keep the torch path.
"""
import torch

from ..model.nets import (DoubleValueAdd, DoubleValueCat, GaussianSharedPolicy, SharedPolicy, SharedValueAdd,
                          SharedValueCat)


def _lin(m):
    return m.weight, m.bias


class FusedNets(object):

    def __init__(self, backend, descs, device):
        self.backend, self.descs, self.device = backend, descs, device
        self._scratch = {}

    @classmethod
    def build(cls, agent, backend, device):
        if not hasattr(backend, "mlp_forward"):
            return None
        try:
            descs = {}
            for name in ("actor", "actor_target"):
                net = getattr(agent, name, None)
                if net is not None:
                    descs[name] = cls._actor_desc(backend, net)
            for name in ("critic", "critic_target"):
                net = getattr(agent, name)
                for suffix, d in cls._critic_descs(backend, net):
                    descs[name + suffix] = d
        except _Unsupported:
            return None
        return cls(backend, descs, device)

    # ------------------------------------------------------------------------------------------ descriptors
    @staticmethod
    def _embed(e):
        if len(e.embeds) != 1:
            raise _Unsupported()
        return _lin(e.embeds[0])

    @classmethod
    def _actor_desc(cls, backend, net):
        Ws, bs = cls._embed(net.state_embed)
        S, E = Ws.shape[1], Ws.shape[0]
        head_dim = 1
        if isinstance(net, SharedPolicy):
            if len(net.affines) != 2 or net.affines[1].weight.shape[0] > 16:
                raise _Unsupported()
            W0, b0 = _lin(net.affines[0])
            W1, b1 = _lin(net.affines[1])
            t, n_out, head_dim = dict(Ws=Ws, bs=bs, W0=W0, b0=b0, W1=W1, b1=b1), 1, W1.shape[0]
        elif isinstance(net, GaussianSharedPolicy):
            if len(net.affines) != 1 or net.affine_mean.weight.shape[0] > 16:
                raise _Unsupported()
            W0, b0 = _lin(net.affines[0])
            W1, b1 = _lin(net.affine_mean)
            W1b, b1b = _lin(net.affine_log_std)
            t, n_out, head_dim = dict(Ws=Ws, bs=bs, W0=W0, b0=b0, W1=W1, b1=b1, W1b=W1b, b1b=b1b), 2, W1.shape[0]
        else:
            raise _Unsupported()
        H = W0.shape[0]
        if net.box_constraint is None or not backend.mlp_supported(E, H, False):
            raise _Unsupported()
        # multi-output actors return raw outputs and leave the (state-dependent) box to the env kernels; a scalar actor
        # with a state-dependent box has no such kernel
        if (head_dim > 1) != bool(net.box_constraint.volatile) or (head_dim > 1 and E > 256):
            raise _Unsupported()
        return backend.MlpDesc(t, S, 0, E, H, n_out, False, head_dim=head_dim)

    @classmethod
    def _critic_descs(cls, backend, net):
        if getattr(net, "partial", False):
            raise _Unsupported()
        if isinstance(net, (SharedValueAdd, SharedValueCat)):
            parts = [("", net.state_embed, net.action_embed, net.affines)]
        elif isinstance(net, (DoubleValueAdd, DoubleValueCat)):
            parts = [("1", net.state_embed1, net.action_embed1, net.affines1),
                     ("2", net.state_embed2, net.action_embed2, net.affines2)]
        else:
            raise _Unsupported()
        cat = isinstance(net, (SharedValueCat, DoubleValueCat))
        out = []
        for suffix, se, ae, affines in parts:
            if len(affines) != 2:
                raise _Unsupported()
            Ws, bs = cls._embed(se)
            Wa, ba = cls._embed(ae)
            W0, b0 = _lin(affines[0])
            W1, b1 = _lin(affines[1])
            S, A, E, H = Ws.shape[1], Wa.shape[1], Ws.shape[0], W0.shape[0]
            if not backend.mlp_supported(E, H, cat) or S > 64 or A > 48:
                raise _Unsupported()
            out.append((suffix, backend.MlpDesc(dict(Ws=Ws, bs=bs, Wa=Wa, ba=ba, W0=W0, b0=b0, W1=W1, b1=b1), S, A, E, H,
                                                1, cat)))
        return out

    def enable_splitk(self, batch_size):
        """Large update batches (>= RPO_SPLITK_FROM rows): give every trainable network a scratch buffer for the split-K
        weights pass of the backward kernels -- Z copies of the network's OWN gradient span (first to last gradient element in
        the flat buffer, what splitk_plan in csrc/mlp_bwd.h addresses), Z = min(256, batch_size // 512) as that plan uses.
        (Round 3 allocated 256 x the whole flat gradient per network: ~64 x what a batch of 16384 touches)"""
        z = max(2, min(256, int(batch_size) // 512))
        for name, d in self.descs.items():
            if "target" in name or d.splitk is not None:
                continue
            grads = [t.grad for t in d.tensors.values() if t is not None and t.grad is not None]
            lo = min(g.data_ptr() for g in grads)
            hi = max(g.data_ptr() + 4 * g.numel() for g in grads)
            d.splitk = torch.zeros(z * ((hi - lo) // 4), device=self.device)

    # ------------------------------------------------------------------------------------------ execution
    def buf(self, key, *shape):
        k = (key,) + shape
        t = self._scratch.get(k)
        if t is None:
            t = self._scratch[k] = torch.zeros(*shape, device=self.device)
        return t

    def _inference_scratch(self, name, n, d):
        """Pre-activation scratch of an inference forward of a wide network -- one pair per STREAM: the overlapped windows run
        the rollout's actor forward on a second stream beside the update's forward of the same network (RPOSAC)."""
        sid = torch.cuda.current_stream().cuda_stream if self.device.type == "cuda" else 0
        return self.buf(name + ".x0i@%x" % sid, n, d.ein), self.buf(name + ".h1i@%x" % sid, n, d.H)

    def forward(self, name, s, a, out, save=False, tanh_box=None):
        """out [n, n_out]; ``save``: keep the pre-activations for ``backward``; ``tanh_box`` = (scale, base)."""
        d = self.descs[name]
        n = out.shape[0]
        x0 = self.buf(name + ".x0", n, d.ein) if save else None
        h1 = self.buf(name + ".h1", n, d.H) if save else None
        if not save and d.E == 256 and n <= 16384:
            # wide networks run layer by layer (rpo_amd/csrc/mlp_gemm.h): the pre-activations travel through memory, so an
            # inference call brings scratch for them too (buffers of their own: a saved forward may still be pending)
            x0, h1 = self._inference_scratch(name, n, d)
        mode, scale, base = (1, tanh_box[0], tanh_box[1]) if tanh_box is not None else (0, 1.0, 0.0)
        self.backend.mlp_forward(d, s, a, out, x0, h1, mode, scale, base)
        return out

    def forward_multi(self, calls):
        """calls = [(name, s, a, out, save), ...] for same-shaped scalar-head networks: one launch where the backend has
        rpo_mlp_forward_multi, else one launch each.  Returns the ``out`` tensors."""
        descs = [self.descs[c[0]] for c in calls]
        same = all((d.S, d.A, d.E, d.H, d.n_out, d.cat) == (descs[0].S, descs[0].A, descs[0].E, descs[0].H, descs[0].n_out,
                                                               descs[0].cat) and d.head_dim <= 1 for d in descs)
        if not same or len(calls) > 4 or not hasattr(self.backend, "mlp_forward_multi"):
            return [self.forward(name, s, a, out, save=save) for name, s, a, out, save in calls]
        packed = []
        for (name, s, a, out, save), d in zip(calls, descs):
            n = out.shape[0]
            wide = (not save) and d.E == 256 and n <= 16384       # (see forward)
            x0i, h1i = self._inference_scratch(name, n, d) if wide else (None, None)
            packed.append((d, s, a, out, self.buf(name + ".x0", n, d.ein) if save else x0i,
                           self.buf(name + ".h1", n, d.H) if save else h1i))
        self.backend.mlp_forward_multi(packed)
        return [c[3] for c in calls]

    def backward(self, name, s, a, dout, da=None, param_grads=True, first_layer_state_only=False, gradmax=None, td=None):
        """``gradmax`` (1-element tensor or None): the weights pass leaves the inf-norm of the gradients it wrote there,
        which saves the separate rpo_absmax launch of clip_grad_norm_ when the buffers were zero before.
        ``td`` (backend.Td): dout is produced by the TD / Huber prologue of the rows pass instead of being read."""
        d = self.descs[name]
        n = dout.shape[0] if td is None else td.dq_out.shape[0]
        kw = {} if gradmax is None else dict(gradmax=gradmax)
        if td is not None:
            kw["td"] = td
        self.backend.mlp_backward(d, s, a, self.buf(name + ".x0", n, d.ein), self.buf(name + ".h1", n, d.H), dout,
                                  self.buf(name + ".dh", n, d.H), self.buf(name + ".dx0", n, d.ein), da, param_grads,
                                  first_layer_state_only, **kw)


    def backward_pair(self, name1, name2, s, a, dout1, dout2, da1=None, da2=None, param_grads=True,
                      first_layer_state_only=False, gradmax=None, td1=None, td2=None):
        """Backward of two same-shaped networks (twin critics) in one pair of launches where the backend has it.
        Returns True when ``gradmax`` holds the inf-norm of the gradients afterwards (not with a shared embedding,
        whose gradient is the sum of two passes)."""
        d1, d2 = self.descs[name1], self.descs[name2]
        shared = any(t is not None and d2.tensors[k] is not None and t.data_ptr() == d2.tensors[k].data_ptr()
                     for k, t in d1.tensors.items())           # a shared embedding: both would accumulate into it at once
        if shared or not hasattr(self.backend, "mlp_backward_pair"):
            gm = None if shared else gradmax
            self.backward(name1, s, a, dout1, da1, param_grads, first_layer_state_only, gradmax=gm, td=td1)
            self.backward(name2, s, a, dout2, da2, param_grads, first_layer_state_only, gradmax=gm, td=td2)
            return gm is not None
        n = dout1.shape[0] if td1 is None else td1.dq_out.shape[0]
        b = self.buf
        self.backend.mlp_backward_pair(
            d1, d2, s, a, b(name1 + ".x0", n, d1.ein), b(name1 + ".h1", n, d1.H), dout1, b(name1 + ".dh", n, d1.H),
            b(name1 + ".dx0", n, d1.ein), da1, b(name2 + ".x0", n, d2.ein), b(name2 + ".h1", n, d2.H), dout2,
            b(name2 + ".dh", n, d2.H), b(name2 + ".dx0", n, d2.ein), da2, param_grads, first_layer_state_only,
            **dict(({} if gradmax is None else dict(gradmax=gradmax)), **({} if td1 is None else dict(td1=td1, td2=td2))))
        return gradmax is not None


class _Unsupported(Exception):
    pass
