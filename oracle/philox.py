"""Oracle for the build's counter-based RNG (Philox4x32-10, Salmon et al. SC'11).  Test infrastructure only.

The reference draws from global numpy / torch generators (scripts/cart_exp.py:9-10, cartpole.py:233,
agent/ddpg_pa.py:109, rpo/utils/buffer.py:32); those streams are not reproducible on a GPU, so the build defines its
own keyed streams: key = seed, counter = (id, index, stream_tag, 0).  Integer outputs are compared bit-exactly with the
HIP implementation (rpo_amd/csrc/common.h).  Known-answer vectors: Random123 kat_vectors, philox4x32-10.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
STREAM_RESET, STREAM_ACT, STREAM_SAMPLE, STREAM_POLICY = 1, 2, 3, 4


def philox4x32(counter, key, rounds=10):
    """counter [n,4] uint32, key [2] or [n,2] uint32 -> [n,4] uint32."""
    c = np.array(counter, dtype=np.uint32, copy=True).reshape(-1, 4)
    k = np.broadcast_to(np.asarray(key, dtype=np.uint32).reshape(-1, 2), (c.shape[0], 2)).copy()
    with np.errstate(over="ignore"):
        for _ in range(rounds):
            p0 = M0 * c[:, 0].astype(np.uint64)
            p1 = M1 * c[:, 2].astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c[:, 1] ^ k[:, 0]
            n1 = p1.astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c[:, 3] ^ k[:, 1]
            n3 = p0.astype(np.uint32)
            c = np.stack([n0, n1, n2, n3], axis=1)
            k = np.stack([k[:, 0] + W0, k[:, 1] + W1], axis=1)
    return c


def draw(seed, ids, index, stream, sub=0):
    """Raw words for key = seed and counter = (id, index, stream, sub); ``ids`` is an array, the rest scalars or arrays."""
    ids = np.asarray(ids, dtype=np.uint32).reshape(-1)
    n = ids.shape[0]
    ctr = np.zeros((n, 4), dtype=np.uint32)
    ctr[:, 0] = ids
    ctr[:, 1] = np.asarray(index, dtype=np.uint64).astype(np.uint32)
    ctr[:, 2] = np.asarray(stream, dtype=np.uint32)
    ctr[:, 3] = np.asarray(sub, dtype=np.uint64).astype(np.uint32)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    return philox4x32(ctr, key)


def u01(words):
    """[0,1) float32 from the top 24 bits."""
    return ((np.asarray(words, dtype=np.uint32) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def normal(a, b):
    """Box-Muller standard normal (float32) from two words; u1 in (0,1]."""
    u1 = (((np.asarray(a, dtype=np.uint32) >> np.uint32(8)).astype(np.float64) + 1.0) * 2.0 ** -24).astype(np.float32)
    u2 = u01(b)
    return (np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.283185307179586) * u2)).astype(np.float32)


def cart_reset(seed, env_ids, episode):
    """State drawn by rpo_cartsafe_reset / the auto-reset of rpo_cartsafe_step: U(-0.05, 0.05)^6 (cartpole.py:233)."""
    r0 = draw(seed, env_ids, episode, STREAM_RESET)
    r1 = draw(seed, env_ids, episode, STREAM_RESET + 0x100)
    u = np.concatenate([u01(r0), u01(r1[:, :2])], axis=1)
    return (np.float32(-0.05) + u * np.float32(0.1)).astype(np.float32)


def pendulum_reset(seed, env_ids, episode):
    """Internal state drawn by rpo_pendulum_reset: U(low, high) of pendulum.py:131-133."""
    lo = np.array([-np.pi / 12, -1.0, 0.95, -0.05], dtype=np.float32)
    hi = np.array([np.pi / 12, 1.0, 1.05, 0.05], dtype=np.float32)
    return (lo + u01(draw(seed, env_ids, episode, STREAM_RESET)) * (hi - lo)).astype(np.float32)


def sample_indices(seed, batch, t, salt, n_valid, updates=0):
    """Row indices drawn by rpo_replay_sample_gather (uniform with replacement, buffer.py:32)."""
    r = draw(seed, np.arange(batch), (int(t) + int(salt)) & 0xFFFFFFFF, STREAM_SAMPLE, updates)
    x = (r[:, 0].astype(np.uint64) << np.uint64(32)) | r[:, 1].astype(np.uint64)
    return np.array([(int(v) * int(n_valid)) >> 64 for v in x], dtype=np.int64)
