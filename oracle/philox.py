"""Philox4x32-10 + Box-Muller, numpy restatement (oracle; test infrastructure only).

The reference draws its reparameterisation noise with an UNSEEDED ``tf.random_normal``
(phiseg/model_zoo/posteriors.py:108,128; priors.py:100,120 -- SURVEY.md Q10), so
"identical seeds" has to be defined by the build.  Contract (shared with
``phiseg_code_amd/csrc/philox.h``):

  counter = (block, sample, stream, step)     key = (seed_lo, seed_hi)
  one Philox call -> 4 uint32 (x0..x3) -> 4 normals:
      u1 = ((x0 >> 8) + 1) * 2^-24 in (0, 1]   u2 = (x1 >> 8) * 2^-24 in [0, 1)
      (24-bit mantissas: exactly representable in fp32, so the HIP kernel forms the same u1/u2)
      n0 = sqrt(-2 ln u1) * cos(2 pi u2)     n1 = sqrt(-2 ln u1) * sin(2 pi u2)
      n2, n3 likewise from (x2, x3)
  element e of sample b (e indexes the per-sample NHWC-flattened tensor):
      block = e // 4, lane = e % 4, sample = global sample index b

so the noise is invariant to how a global batch is sharded over ranks.
Known-answer vectors: Random123 ``philox4x32-10`` (see tests/test_oracle_philox.py).
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: uint32 array [..., 4]; key: uint32 array [..., 2] (broadcastable). -> uint32 [..., 4]."""
    ctr = np.asarray(ctr, dtype=np.uint32)
    key = np.asarray(key, dtype=np.uint32)
    c0, c1, c2, c3 = (ctr[..., i].astype(np.uint64) for i in range(4))
    k0 = np.broadcast_to(key[..., 0], c0.shape).astype(np.uint32)
    k1 = np.broadcast_to(key[..., 1], c0.shape).astype(np.uint32)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c0
            p1 = _M1 * c2
            hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
            hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
            n0 = (hi1 ^ c1 ^ k0.astype(np.uint64)) & _MASK
            n2 = (hi0 ^ c3 ^ k1.astype(np.uint64)) & _MASK
            c0, c1, c2, c3 = n0, lo1, n2, lo0
            k0 = (k0 + _W0).astype(np.uint32)
            k1 = (k1 + _W1).astype(np.uint32)
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def normal(seed, step, stream, n_samples, per_sample, sample_offset=0, dtype=np.float32):
    """Standard normals [n_samples, per_sample] under the contract in the module docstring."""
    nblk = (per_sample + 3) // 4
    blk = np.arange(nblk, dtype=np.uint32)[None, :]
    smp = (np.arange(n_samples, dtype=np.uint64) + np.uint64(sample_offset)).astype(np.uint32)[:, None]
    ctr = np.empty((n_samples, nblk, 4), dtype=np.uint32)
    ctr[..., 0] = blk
    ctr[..., 1] = smp
    ctr[..., 2] = np.uint32(stream)
    ctr[..., 3] = np.uint32(step)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    x = (philox4x32_10(ctr, key) >> np.uint32(8)).astype(np.float64)
    two24 = 2.0 ** -24
    out = np.empty((n_samples, nblk, 4), dtype=np.float64)
    for j in (0, 2):
        u1 = (x[..., j] + 1.0) * two24
        u2 = x[..., j + 1] * two24
        r = np.sqrt(-2.0 * np.log(u1))
        out[..., j] = r * np.cos(2.0 * np.pi * u2)
        out[..., j + 1] = r * np.sin(2.0 * np.pi * u2)
    return out.reshape(n_samples, nblk * 4)[:, :per_sample].astype(dtype)


def uniform01(seed, step, stream, n, dtype=np.float64):
    """n uniforms in [0,1) (used for synthetic data / weight init streams): sample index 0."""
    nblk = (n + 3) // 4
    ctr = np.zeros((nblk, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(nblk, dtype=np.uint64).astype(np.uint32)
    ctr[:, 1] = (np.arange(nblk, dtype=np.uint64) >> np.uint64(32)).astype(np.uint32)
    ctr[:, 2] = np.uint32(stream)
    ctr[:, 3] = np.uint32(step)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    x = philox4x32_10(ctr, key).astype(np.float64) * 2.0 ** -32
    return x.reshape(-1)[:n].astype(dtype)
