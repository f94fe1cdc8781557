"""R14 (reference tfwrapper/utils.py:214-271): variable initialisation on the documented Philox stream.

The product's initialiser (phiseg_code_amd/philox_host.py + tfwrapper/utils.py) and the oracle's (oracle/init.py) are
independent implementations of one contract; here they are compared bit for bit (fp32), and the distribution is
checked against TF 1.12's variance_scaling_initializer(factor=2, FAN_IN, uniform=False) semantics."""
import math

import numpy as np
import pytest

from oracle import init as oinit
from oracle import philox as ophilox
from phiseg_code_amd import philox_host


def test_philox_known_answers_product_side():
    """Random123 philox4x32-10 KATs (same vectors the oracle is pinned with, tests/test_oracle_philox.py)."""
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
             (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kats:
        got = philox_host.philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0]
        assert tuple(int(v) for v in got) == want


def test_normals_and_uniforms_match_oracle_streams():
    for seed, step, stream, n in [(0, 0, 5, 17), (42, 3, 123456, 1000), ((7 << 32) | 9, 1, 2 ** 29, 33)]:
        a = philox_host.normals(seed, step, stream, n)
        b = ophilox.normal(seed, step, stream, 1, n, dtype=np.float64)[0]
        assert np.array_equal(a, b)
        assert np.array_equal(philox_host.uniforms(seed, step, stream, n), ophilox.uniform01(seed, step, stream, n))


def _model(norm_name="batch_norm", n0=4):
    import types
    from phiseg_code_amd.phiseg import phiseg_model
    from phiseg_code_amd.phiseg.experiments import phiseg_7_5 as base
    from phiseg_code_amd.tfwrapper import normalisation as tfnorm
    cfg = types.SimpleNamespace(**{k: getattr(base, k) for k in dir(base) if not k.startswith("_")})
    cfg.n0 = n0
    cfg.image_size = (64, 64, 1)
    cfg.layer_norm = getattr(tfnorm, norm_name)
    return phiseg_model.phiseg(cfg)


@pytest.mark.parametrize("norm", ["batch_norm", "group_norm2D", "instance_norm2D"])
@pytest.mark.parametrize("seed", [0, 3])
def test_product_init_is_bit_equal_to_oracle_init(norm, seed):
    """Every variable of the graph (700 for phiseg_7_5): product initial_value(seed) == oracle variable_value(seed), fp32."""
    model = _model(norm)
    n = 0
    for name, v in model.graph.variables.items():
        got = v.initial_value(seed)
        want = np.asarray(oinit.variable_value(name, list(v.shape), seed, perturbed=False), dtype=np.float32).reshape(v.shape)
        assert got.dtype == np.float32 and got.shape == tuple(v.shape)
        assert np.array_equal(got, want), name
        n += 1
    assert n > 300


def test_he_normal_distribution():
    """sigma = sqrt(1.3 * 2 / fan_in), |w| <= 2 sigma, and the truncated normal's own standard deviation (0.8796 sigma)."""
    model = _model()
    checked = 0
    for name, v in model.graph.variables.items():
        if not name.endswith("/W") or v.size < 4096:
            continue
        w = v.initial_value(0).astype(np.float64)
        fan_in = int(np.prod(v.shape[:-1]))
        sigma = math.sqrt(1.3 * 2.0 / fan_in)
        assert np.abs(w).max() <= 2.0 * sigma * (1 + 1e-6)
        assert abs(w.std() / sigma - 0.8796) < 0.03, name
        assert abs(w.mean()) < 4 * sigma / math.sqrt(w.size)
        checked += 1
    assert checked >= 5
    # biases and batch-norm variables: tfwrapper/utils.py:261-271, tf.contrib.layers.batch_norm defaults
    for name, v in model.graph.variables.items():
        leaf = name.rsplit("/", 1)[-1]
        x = v.initial_value(0)
        if leaf in ("b", "beta", "moving_mean"):
            assert not x.any()
        if leaf in ("gamma", "moving_variance"):
            assert (x == 1).all()


def test_init_is_independent_of_creation_order_and_rank():
    a = _model().graph.variables
    b = _model().graph.variables
    name = [n for n in a if n.endswith("/W")][7]
    assert np.array_equal(a[name].initial_value(5), b[name].initial_value(5))
    assert not np.array_equal(a[name].initial_value(5), a[name].initial_value(6))


def test_philox_synthetic_batch_matches_oracle_and_shards():
    """The product's seeded synthetic batch equals the oracle's (bench.py feeds the GPU leg and the CPU baseline from it),
    and a rank's shard is a slice of the global batch."""
    from phiseg_code_amd.data import synthetic
    x, s = synthetic.philox_batch(5, 32, 4, seed=1234)
    xo, so = oinit.synthetic_batch(5, 32, 4, 1234)
    assert np.array_equal(x, xo) and np.array_equal(s, so)
    x2, s2 = synthetic.philox_batch(2, 32, 4, seed=1234, sample_offset=3)
    assert np.array_equal(x2, x[3:5]) and np.array_equal(s2, s[3:5])
