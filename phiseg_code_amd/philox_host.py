"""Host mirror of ``csrc/philox.h`` (Philox4x32-10 + Box-Muller) for everything the product draws on the host:
variable initialisation (R14, tfwrapper/utils.py:214-271 of the reference) and the synthetic input generator.

The reference never seeds TensorFlow, so "identical seeds" is this build's contract:

    counter = (block, sample, stream, step)    key = (seed & 0xffffffff, seed >> 32)
    4 outputs x0..x3 of one call -> 4 normals by Box-Muller on 24-bit uniforms
        u1 = ((x >> 8) + 1) / 2^24,  u2 = (x' >> 8) / 2^24,  n = sqrt(-2 ln u1) * {cos, sin}(2 pi u2)
    element e of sample b: block = e // 4, lane = e % 4

A variable's stream id is ``crc32(tf_variable_name) & 0x3fffffff``: the value of a variable depends on (seed, name,
shape) only -- not on creation order, the rank, or which other variables exist -- so every data-parallel replica and the
CPU oracle start from bit-identical weights.
"""
import zlib

import numpy as np

_MUL_A, _MUL_B = 0xD2511F53, 0xCD9E8D57
_WEYL_A, _WEYL_B = 0x9E3779B9, 0xBB67AE85
_U32 = 0xFFFFFFFF


def philox4x32_10(counter, key):
    """counter: [..., 4] uint32, key: (k0, k1) python ints -> [..., 4] uint32 (ten rounds)."""
    c = [np.asarray(counter[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = int(key[0]) & _U32, int(key[1]) & _U32
    for _ in range(10):
        pa = c[0] * np.uint64(_MUL_A)
        pb = c[2] * np.uint64(_MUL_B)
        c = [((pb >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)) & np.uint64(_U32), pb & np.uint64(_U32),
             ((pa >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)) & np.uint64(_U32), pa & np.uint64(_U32)]
        k0 = (k0 + _WEYL_A) & _U32
        k1 = (k1 + _WEYL_B) & _U32
    return np.stack(c, axis=-1).astype(np.uint32)


def _key(seed):
    seed = int(seed)
    return seed & _U32, (seed >> 32) & _U32


def _words(seed, step, stream, sample, nblk):
    ctr = np.zeros((nblk, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(nblk, dtype=np.uint64) & np.uint64(_U32)
    ctr[:, 1] = np.uint32(sample)
    ctr[:, 2] = np.uint32(stream)
    ctr[:, 3] = np.uint32(step)
    return philox4x32_10(ctr, _key(seed))


def normals(seed, step, stream, n, sample=0):
    """n standard normals (float64) of one sample of one stream."""
    nblk = (int(n) + 3) // 4
    x = (_words(seed, step, stream, sample, nblk) >> np.uint32(8)).astype(np.float64)
    out = np.empty((nblk, 4), dtype=np.float64)
    for j in (0, 2):
        r = np.sqrt(-2.0 * np.log((x[:, j] + 1.0) * 2.0 ** -24))
        ang = 2.0 * np.pi * (x[:, j + 1] * 2.0 ** -24)
        out[:, j] = r * np.cos(ang)
        out[:, j + 1] = r * np.sin(ang)
    return out.reshape(-1)[:int(n)]


def uniforms(seed, step, stream, n):
    """n uniforms in [0, 1) (float64): 32-bit words / 2^32; the block index spills into the sample word past 2^32."""
    nblk = (int(n) + 3) // 4
    ctr = np.zeros((nblk, 4), dtype=np.uint32)
    idx = np.arange(nblk, dtype=np.uint64)
    ctr[:, 0] = idx & np.uint64(_U32)
    ctr[:, 1] = idx >> np.uint64(32)
    ctr[:, 2] = np.uint32(stream)
    ctr[:, 3] = np.uint32(step)
    return (philox4x32_10(ctr, _key(seed)).astype(np.float64) * 2.0 ** -32).reshape(-1)[:int(n)]


def stream_of(name):
    return zlib.crc32(name.encode()) & 0x3FFFFFFF


class VariableStream:
    """The random source handed to a variable's initializer (graph.Variable.initial_value)."""

    def __init__(self, seed, name):
        self.seed, self.stream = int(seed), stream_of(name)

    def standard_normal(self, shape, step=1):
        n = int(np.prod(shape))
        return normals(self.seed, step, self.stream, n).reshape(shape)

    def uniform(self, lo, hi, shape, step=2):
        n = int(np.prod(shape))
        return lo + (hi - lo) * uniforms(self.seed, step, self.stream, n).reshape(shape)

    def truncated_normal(self, shape):
        """TF's truncated_normal semantics (resample beyond two standard deviations) realised by rejection on ONE
        stream: the first prod(shape) draws with |n| <= 2 out of 2 n + 64 candidates (P(reject) = 4.55 %)."""
        n = int(np.prod(shape))
        raw = normals(self.seed, 0, self.stream, 2 * n + 64)
        keep = raw[np.abs(raw) <= 2.0]
        if keep.size < n:
            raise RuntimeError("truncated_normal: candidate pool exhausted (n=%d)" % n)
        return keep[:n].reshape(shape)
