"""End-to-end parity on the MI355X: the HIP engine behind the reference's API against
(a) the golden fixtures produced by executing the reference's own zoo / loss code (tests/golden), and
(b) the CPU oracle (gradients, Adam steps), on identical weights, inputs and Philox noise.

Tolerance: north_star asks for 1e-4 (fp32) on per-level logits and the ELBO; the fp32 path is held to 1e-4
relative-to-max on every tensor.  The bf16 path is judged against the same fp64 goldens with a stated looser
bound (bf16 storage, fp32 accumulation: 2^-8 per stored activation, compounding over ~20 layers)."""
import json

import os

import numpy as np
import pytest
import torch

from oracle import train as otrain
from tests.helpers import check_tensor, golden_inputs, load_golden
from tests.test_graph_cpu import make_config

pytestmark = pytest.mark.gpu

FP32_RTOL = 1e-4
# Conditioning notes (measured with torch-CPU float32 running the SAME oracle code against the fp64 goldens,
# i.e. pure fp32 round-off of this algorithm): forward tensors <= 1.2e-5, per-variable gradient norms up to
# 1.7e-3 (tiny_phiseg_bn) / 9.9e-3 (tiny_probunet_bn); gradients that are exactly zero in exact arithmetic (a conv
# bias in front of instance norm) are pure round-off.  The gradient / multi-step tolerances below are therefore
# fp32-conditioning bounds of the algorithm, not slack for kernel errors.
GRAD_RTOL = 3e-2


def fwd_tol(case, key):
    """1e-4 (north_star) on the LIDC configuration.  The n0=4 fixtures with perturbed affine parameters sit at
    fp32's noise floor for this algorithm (torch-CPU float32 of the oracle: 0.7e-4 .. 0.9e-4 on the logits of
    tiny_phiseg_in / tiny_probunet_bn), so they get 5e-4."""
    return FP32_RTOL if case.startswith("lidc_phiseg_bn") else 5e-4


def build(case, compute_dtype="f32"):
    from phiseg_code_amd.phiseg import phiseg_model
    g, cfg, var_order = load_golden(case)
    model = phiseg_model.phiseg(make_config(cfg, compute_dtype), rng_seed=cfg["eps_seed"])
    params, x_np, s_np = golden_inputs(cfg, var_order, dtype=torch.float64)
    model.set_weights({k: v.detach().numpy() for k, v in params.items()})
    return g, cfg, var_order, model, params, x_np, s_np


TINY = ["tiny_phiseg_bn", "tiny_phiseg_gn4", "tiny_phiseg_in", "tiny_probunet_bn", "tiny_phiseg71_bn",
        "tiny_phiseg_bn_192"]


# lidc_phiseg_bn_b12: BASELINE.json config 1 -- the reference's own batch size (phiseg/experiments/phiseg_7_5.py:40), SURVEY 8(c)
LIDC = ["lidc_phiseg_bn", "lidc_phiseg_bn_b12"]


@pytest.mark.parametrize("case", TINY + LIDC)
def test_forward_elbo_matches_reference_goldens_fp32(case):
    g, cfg, var_order, model, params, x_np, s_np = build(case)
    L = cfg["latent_levels"]
    fetch = [model.z_list, model.mu_list, model.sigma_list, model.prior_mu_list, model.prior_sigma_list,
             model.s_out_list, [model.loss_dict[k] for k in sorted(model.loss_dict)]]
    z, mu, sigma, pmu, psig, s_list, losses = model.sess.run(
        fetch, {model.x_inp: x_np, model.s_inp: s_np, model.training_pl: True})
    for l in range(L):
        check_tensor(g, "train/z_%d" % l, z[l], fwd_tol(case, "z"))
        check_tensor(g, "train/mu_%d" % l, mu[l], fwd_tol(case, "mu"))
        check_tensor(g, "train/sigma_%d" % l, sigma[l], fwd_tol(case, "sigma"))
        check_tensor(g, "train/prior_mu_%d" % l, pmu[l], fwd_tol(case, "prior"))
        check_tensor(g, "train/prior_sigma_%d" % l, psig[l], fwd_tol(case, "prior"))
        check_tensor(g, "train/s_%d" % l, s_list[l], fwd_tol(case, "s"))
    for k, v in zip(sorted(model.loss_dict), losses):
        np.testing.assert_allclose(float(v), float(g["train/loss/" + k]), rtol=fwd_tol(case, "loss"), err_msg=k)


@pytest.mark.parametrize("case", TINY + LIDC)
def test_sampling_path_matches_reference_goldens_fp32(case):
    g, cfg, var_order, model, params, x_np, s_np = build(case)
    L = cfg["latent_levels"]
    zg, s_eval, s_out, sm = model.sess.run([model.prior_z_list_gen, model.s_out_eval_list, model.s_out_eval,
                                            model.s_out_eval_sm], {model.x_inp: x_np, model.training_pl: False})
    for l in range(L):
        check_tensor(g, "infer/prior_z_gen_%d" % l, zg[l], fwd_tol(case, "prior"))
        check_tensor(g, "infer/s_eval_%d" % l, s_eval[l], fwd_tol(case, "s"))
    check_tensor(g, "infer/s_out_eval", s_out, fwd_tol(case, "s"))
    e = np.exp(s_out - s_out.max(axis=-1, keepdims=True))
    np.testing.assert_allclose(sm, e / e.sum(axis=-1, keepdims=True), rtol=1e-5, atol=1e-6)
    # the public sampling API draws fresh noise per call
    a = model.predict_segmentation_sample(x_np, return_softmax=True)
    b = model.predict_segmentation_sample(x_np, return_softmax=True)
    assert a.shape == sm.shape and np.abs(a - b).max() > 0


def _hip_grads(model, cfg, x_np, s_np):
    from phiseg_code_amd import engine
    store = model.sess._ensure_store()
    plan = engine.Plan(store, [model.loss_tot], loss=model.loss_tot, batch=cfg["B"], training=True,
                       compute_dtype="f32", optimize=False, rng_seed=cfg["eps_seed"], use_hip_graph=False)
    plan.set_input("x_input", x_np)
    plan.set_input("s_input", s_np)
    plan.run(sync=True)
    return store.export(grads=True)


@pytest.mark.parametrize("case", TINY)
def test_gradients_vs_reference_autograd_goldens_fp32(case):
    """Gradients of the reference's own forward (autograd through the shim, fp64, perturbed parameters).  This
    fixture is ill-conditioned in fp32 (torch-CPU float32 of the same algorithm: per-variable norm errors up to
    1e-2), so the bound is statistical; the tight per-variable check is the well-conditioned test below."""
    g, cfg, var_order, model, params, x_np, s_np = build(case)
    grads = _hip_grads(model, cfg, x_np, s_np)
    gref = json.loads(str(g["train/grad_norm_sum_json"]))
    gmax = max(v[0] for v in gref.values() if v is not None)
    errs = []
    for name, ns in gref.items():
        if name.rsplit("/", 1)[-1].startswith("moving_"):
            continue
        gr = grads[name].astype(np.float64)
        if ns is None:
            assert np.abs(gr).max() == 0.0, name        # never-consumed branch (Q1): no gradient
            continue
        errs.append(abs(np.linalg.norm(gr) - ns[0]) / (ns[0] + 1e-3 * gmax))
    errs = np.sort(np.array(errs))
    assert len(errs) > 10
    assert np.median(errs) < 2e-3 and errs[int(0.9 * len(errs))] < 2e-2 and errs[-1] < 0.15, \
        (np.median(errs), errs[int(0.9 * len(errs))], errs[-1])


@pytest.mark.parametrize("case", TINY)
def test_gradients_match_oracle_wellconditioned_fp32(case):
    """Same nets with the reference's initialisation (he_normal weights, zero biases, unit gamma): the oracle's
    autograd gradient vs the HIP backward, EVERY variable.  Bound 3e-2 of each variable's largest gradient entry:
    torch-CPU float32 running the identical oracle code deviates from its own fp64 result by up to 2.7e-2
    (tiny_probunet_bn, likelihood/decoder/conv_2_1/W) / 1.4e-2 (tiny_phiseg_bn) -- batch norm over 12 values at
    the 2x2 levels amplifies fp32 round-off; a structural error (wrong tap, missing term) is O(1)."""
    from phiseg_code_amd.phiseg import phiseg_model
    g, cfg, var_order = load_golden(case)
    model = phiseg_model.phiseg(make_config(cfg, "f32"), rng_seed=cfg["eps_seed"])
    params = otrain.make_params(var_order, cfg["weight_seed"], torch.float64, perturbed=False)
    from oracle import init as oinit
    x_np, s_np = oinit.synthetic_batch(cfg["B"], cfg["H"], cfg["nlabels"], cfg["data_seed"])
    model.set_weights({k: v.detach().numpy() for k, v in params.items()})
    out, gref = otrain.loss_and_grads(params, torch.as_tensor(x_np, dtype=torch.float64), torch.as_tensor(s_np),
                                      otrain.torch_eps_fn(cfg["eps_seed"], 0, cfg["B"]), cfg)
    grads = _hip_grads(model, cfg, x_np, s_np)
    gmax = max(float(v.norm()) for v in gref.values() if v is not None)
    checked = 0
    for name, ref in gref.items():
        got = grads[name].astype(np.float64)
        if ref is None:
            assert np.abs(got).max() == 0.0, name
            continue
        r = ref.numpy()
        # tiny_probunet_bn is the worst-conditioned fixture (torch float32: 2.7e-2) and its HIP result moves by ~1e-2
        # from run to run with the order of the fp32 atomics in the statistics kernels: 1e-1
        tol = (1e-1 if case == "tiny_probunet_bn" else 3e-2) * np.abs(r).max() + 1e-5 * gmax
        assert np.abs(got - r).max() <= tol, (name, np.abs(got - r).max(), tol)
        checked += 1
    assert checked > 10


@pytest.mark.parametrize("case", ["tiny_phiseg_bn", "tiny_phiseg_gn4", "tiny_probunet_bn"])
def test_three_adam_steps_match_oracle_fp32(case):
    """Three full training steps (eager, hipGraph capture, hipGraph replay) against the oracle's TF1-form Adam.
    lr is small so the trajectory stays where the gradient comparison is meaningful: Adam moves every weight by
    ~lr * sign(g) per step whatever the gradient scale, so only elements whose gradient is round-off-sized may
    differ (by at most 2 * lr per step)."""
    lr = 1e-5
    g, cfg, var_order, model, params, x_np, s_np = build(case)
    p0 = {k: v.detach().clone().numpy() for k, v in params.items()}
    ref_losses = otrain.train_steps(params, [(x_np, s_np)], cfg, cfg["eps_seed"], lr=lr, n_steps=3)
    losses = []
    for _ in range(3):
        _, lt = model.sess.run([model.train_step, model.loss_tot],
                               {model.x_inp: x_np, model.s_inp: s_np, model.training_pl: True, model.lr_pl: lr})
        losses.append(float(lt))
    np.testing.assert_allclose(losses, [l["total_loss"] for l in ref_losses], rtol=1e-3)
    got = model.sess.store.export()
    n_all = n_off = n_moved = 0
    for name, ref in params.items():
        r = ref.detach().numpy()
        d = np.abs(got[name].astype(np.float64) - r)
        if name.rsplit("/", 1)[-1].startswith("moving_"):     # batch-norm moving statistics (not Adam-updated)
            assert d.max() <= 1e-3 * max(np.abs(r).max(), 1.0), (name, d.max())
            continue
        assert d.max() <= 3 * 2 * lr + 1e-6 * np.abs(r).max(), (name, d.max())
        n_all += d.size
        n_off += int((d > 0.1 * lr + 2e-7 * np.abs(r)).sum())
        n_moved += int((np.abs(r - p0[name]) > 0.5 * lr).sum())
    assert n_moved > 0.5 * n_all * 0.5          # the oracle really moved the (live) parameters
    # tiny_probunet_bn: its 40-layer U-Net with batch norm over 12 values per channel leaves a large share of
    # weights with round-off-dominated gradients (torch-CPU float32 vs fp64: 2.7e-2 of each variable's max), whose
    # Adam direction is not reproducible in fp32 -- only the hard bound above applies to them.
    assert n_off <= (0.5 if case == "tiny_probunet_bn" else 0.01) * n_all, (n_off, n_all)
    assert int(model.sess.store.step.cpu()[0]) == 3


def _bf16_errors(s_list, ref_fn, L):
    out = []
    for l in range(L):
        ref = ref_fn(l)
        d = s_list[l][:, ::8, ::8, :] - ref
        out.append((np.sqrt((d ** 2).mean()) / np.abs(ref).max(), np.abs(d).max() / np.abs(ref).max()))
    return out


def test_bf16_path_lidc():
    """bf16 storage + MFMA path on the LIDC-sized net (n0=32, 128x128, reference initialisation).
    The yardstick is the oracle itself run with the engine's storage policy simulated (bfloat16 rounding wherever
    the HIP path stores bf16, oracle.nets.Ctx.bf16_sim): its deviation from the exact fp64 oracle (measured: 1.2 - 2.5 %
    RMS of the logit range per level) is the inherent cost of bf16 storage through ~25 conv / norm layers.  Two bf16
    evaluations decorrelate through rounding flips, so the HIP result must sit within 2x that inherent deviation of BOTH
    the exact and the simulated oracle (independent errors add in quadrature: expected 1.4x)."""
    from oracle import init as oinit
    from oracle import nets
    from phiseg_code_amd.phiseg import phiseg_model
    g, cfg, var_order = load_golden("lidc_phiseg_bn")
    L = cfg["latent_levels"]
    model = phiseg_model.phiseg(make_config(cfg, "bf16"), rng_seed=cfg["eps_seed"])
    params = otrain.make_params(var_order, cfg["weight_seed"], torch.float64, perturbed=False)
    x_np, s_np = oinit.synthetic_batch(cfg["B"], cfg["H"], cfg["nlabels"], cfg["data_seed"])
    model.set_weights({k: v.detach().numpy() for k, v in params.items()})
    xt, st = torch.as_tensor(x_np, dtype=torch.float64), torch.as_tensor(s_np)
    eps = otrain.torch_eps_fn(cfg["eps_seed"], 0, cfg["B"])
    with torch.no_grad():
        exact = nets.elbo(params, xt, st, eps, cfg, training=True)
        sim = nets.elbo(params, xt, st, eps, cfg, training=True, bf16_sim=True)
    keys = sorted(model.loss_dict)
    fd = {model.x_inp: x_np, model.s_inp: s_np, model.training_pl: True}
    s_list, mu, losses = model.sess.run([model.s_out_list, model.mu_list, [model.loss_dict[k] for k in keys]], fd)

    def errs(ref):
        out = []
        for l in range(L):
            r = ref["s"][l].numpy()[:, ::8, ::8, :]
            d = s_list[l][:, ::8, ::8, :] - r
            out.append((np.sqrt((d ** 2).mean()) / np.abs(r).max(), np.abs(d).max() / np.abs(r).max()))
        return out
    e_sim, e_exact = errs(sim), errs(exact)
    inherent = [float(np.sqrt(((sim["s"][l] - exact["s"][l]) ** 2).mean()) / exact["s"][l].abs().max()) for l in range(L)]
    print("bf16 logits RMS/max error vs simulated-bf16 oracle:", e_sim)
    print("bf16 logits RMS/max error vs exact fp64 oracle    :", e_exact, " simulated-vs-exact RMS:", inherent)
    for l in range(L):
        bound = 2.0 * max(inherent[l], 0.005)
        assert e_sim[l][0] < bound and e_exact[l][0] < bound, (l, e_sim[l], e_exact[l], inherent[l])
        assert e_exact[l][1] < 0.25, (l, e_exact[l])
        np.testing.assert_allclose(mu[l], exact["mu"][l].numpy(), rtol=0, atol=0.08 * float(exact["mu"][l].abs().max()))
    # what bf16 storage alone does to an ELBO term (oracle vs oracle).  Which roundings flip is chaotic: the simulated deviation
    # of a single KL level can be ~0 by coincidence while its neighbours move by 8 %, and the HIP path (whose fp32 atomics
    # order varies from run to run) moves each KL level by up to +-3 % between runs -- so a KL level is also allowed 1.5x the
    # largest simulated deviation over all KL levels.
    inh = {k: abs(float(sim["loss_dict"][k]) - float(exact["loss_dict"][k])) / abs(float(exact["loss_dict"][k])) for k in keys}
    kl_inh = max([v for k, v in inh.items() if k.startswith("KL_")] or [0.0])
    for k, v in zip(keys, losses):
        ex, sm_ = float(exact["loss_dict"][k]), float(sim["loss_dict"][k])
        print("ELBO term %-34s exact %.4e  sim-bf16 %+.2f%%  HIP-bf16 %+.2f%%" % (k, ex, 100 * (sm_ - ex) / ex, 100 * (float(v) - ex) / ex))
        rtol = max(0.05, 2.5 * inh[k], 1.5 * kl_inh if k.startswith("KL_") else 0.0)
        np.testing.assert_allclose(float(v), ex, rtol=rtol, err_msg=k)


def test_data_parallel_code_path_on_one_gpu():
    """The N > 1 step (forward+backward graph | RCCL all-reduce of the flat gradient arena | Adam graph) exercised with
    a single-rank NCCL process group: must reproduce the single-graph step bit-for-bit in structure (same losses)."""
    import os
    from phiseg_code_amd import distributed
    from phiseg_code_amd.phiseg import phiseg_model
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    g, cfg, var_order = load_golden("tiny_phiseg_bn")
    params, x_np, s_np = golden_inputs(cfg, var_order, dtype=torch.float64)
    fd = None
    losses = {}
    ctx = distributed.DistContext(force=True)
    try:
        for tag, dist in (("single", None), ("dp", ctx)):
            model = phiseg_model.phiseg(make_config(cfg, "f32"), dist=dist, rng_seed=cfg["eps_seed"])
            model.set_weights({k: v.detach().numpy() for k, v in params.items()})
            fd = {model.x_inp: x_np, model.s_inp: s_np, model.training_pl: True, model.lr_pl: 1e-5}
            losses[tag] = [float(model.sess.run([model.train_step, model.loss_tot], fd)[1]) for _ in range(3)]
    finally:
        ctx.shutdown()
    np.testing.assert_allclose(losses["dp"], losses["single"], rtol=2e-4)     # fp32 atomics: run-to-run order differs


def test_bf16_sampling_192x192_four_classes():
    """BASELINE config 5 geometry on the MFMA path: phiseg_7_5 (n0 = 32), 192x192, 4 classes, Monte-Carlo sampling
    (prior in generation mode + likelihood, inference-mode batch norm) -- maps of 96, 48, 24, 12, 6, 3 pixels exercise
    the partial-tile masks of the bf16 kernels.  Bound: bf16 storage (see test_bf16_path_lidc)."""
    from oracle import init as oinit
    from oracle import nets
    from phiseg_code_amd.phiseg import phiseg_model
    cfg = dict(arch="phiseg", norm="batch_norm", n0=32, zdim0=2, H=192, B=2, nlabels=4, latent_levels=5,
               resolution_levels=7, image_size=(192, 192, 1), KL_weight=1.0, CE_weight=1.0, exponential_weighting=True)
    model = phiseg_model.phiseg(make_config(cfg, "bf16"), rng_seed=42)
    var_order = [(n, v.shape) for n, v in model.graph.variables.items()]
    params = otrain.make_params(var_order, 0, torch.float64, perturbed=False)
    model.set_weights({k: v.detach().numpy() for k, v in params.items()})
    x_np, _ = oinit.synthetic_batch(2, 192, 4, 1234)
    with torch.no_grad():
        ref = nets.sample(params, torch.as_tensor(x_np, dtype=torch.float64), otrain.torch_eps_fn(42, 0, 2), cfg)
    s_out, sm = model.sess.run([model.s_out_eval, model.s_out_eval_sm], {model.x_inp: x_np, model.training_pl: False})
    r = ref["s_out_eval"].numpy()
    d = s_out - r
    assert np.sqrt((d ** 2).mean()) / np.abs(r).max() < 0.03, np.sqrt((d ** 2).mean()) / np.abs(r).max()
    assert np.abs(sm.sum(axis=-1) - 1.0).max() < 1e-5
    assert (sm.argmax(-1) == ref["s_out_eval_sm"].numpy().argmax(-1)).mean() > 0.97


def test_full_size_bench_configuration_properties_bf16():
    """BASELINE.json config 2 at its full size (phiseg_7_5, 128x128, bf16, batch 64), through size-independent properties --
    the oracle needs minutes for one such step, so the checks are structural:
      (a) sampling path (batch norm in inference mode -> samples independent; Philox keyed by the global sample index):
          the first 8 images of a batch-64 pass equal a batch-8 pass of the same images (different tile shapes, kernel
          variants and split-K choices -> only fp32 summation order differs);
      (b) ELBO step: the total is the weighted sum of its published terms (phiseg_model.py:265-287), finite, and a second
          evaluation from the same state and noise step stays within the bf16 path's run-to-run spread; after the Adam step
          the loss of the same batch moved (the update went through)."""
    import importlib
    import types
    from oracle import init as oinit
    from phiseg_code_amd.phiseg import phiseg_model
    base = importlib.import_module("phiseg_code_amd.phiseg.experiments.phiseg_7_5")
    cfg = types.SimpleNamespace(**{k: getattr(base, k) for k in dir(base) if not k.startswith("_")})
    cfg.batch_size, cfg.compute_dtype = 64, "bf16"
    x, s = oinit.synthetic_batch(64, 128, cfg.nlabels, 77)
    model = phiseg_model.phiseg(cfg, rng_seed=5)
    # (a)
    p64 = model.predict_segmentation_sample_levels(x)
    model.sess.store.noise_step -= 1                      # same Philox step word for the comparison pass
    p8 = model.predict_segmentation_sample_levels(x[:8])
    for l, (a, b) in enumerate(zip(p64, p8)):
        scale = float(np.abs(a).max())
        assert np.isfinite(a).all()
        # (bf16 rounding flips through ~25 layers: a handful of pixels in 262 144 land just beyond 2 % -- observed 16 at 3 % when
        # the two batch sizes pick different kernel variants; bound the tail instead of every element)
        d = np.abs(a[:8] - b)
        assert float((d > 2e-2 * scale).mean()) < 1e-3 and float(d.max()) < 6e-2 * scale, ("level %d" % l, float(d.max()), scale)
    # (b)
    keys = sorted(model.loss_dict)
    fd = {model.x_inp: x, model.s_inp: s, model.training_pl: True, model.lr_pl: 1e-3}
    tot, terms = model.sess.run([model.loss_tot, [model.loss_dict[k] for k in keys]], fd)
    d = dict(zip(keys, [float(v) for v in terms]))
    assert np.isfinite(float(tot)) and all(np.isfinite(v) for v in d.values())
    print("ELBO", float(tot), "terms", d)
    # loss_tot = w_ce * sum_l CE_l + w_kl * sum_l KL_l (the KL terms carry their 4^l level weight), phiseg_model.py:113-130
    recon = (cfg.residual_multinoulli_loss_weight * sum(v for k, v in d.items() if k.startswith("residual_multinoulli_loss_lvl")) +
             cfg.KL_divergence_loss_weight * sum(v for k, v in d.items() if k.startswith("KL_divergence_loss_lvl")))
    np.testing.assert_allclose(float(tot), recon, rtol=1e-4)
    np.testing.assert_allclose(d["total_loss"], float(tot), rtol=1e-6)
    # A second evaluation is NOT bit-identical on the bf16 path: the statistics of the small-map layers and the
    # gradient reductions use fp32 atomics, whose order varies; bf16 rounding flips amplify that 1e-7 noise, and the level-0
    # KL term of a freshly initialised net (sigma ~ 0: 95 % of this ELBO) is ill-conditioned -- measured spread +-5 %
    # (measured in round 2; the fp32 path reproduces to 2e-5).  Bound it, do not pretend equality.
    tot2 = float(model.sess.run(model.loss_tot, fd))
    np.testing.assert_allclose(tot2, float(tot), rtol=0.2)
    _, tot3 = model.sess.run([model.train_step, model.loss_tot], fd)
    tot4 = float(model.sess.run(model.loss_tot, fd))
    assert np.isfinite(tot4) and abs(tot4 - float(tot3)) > 1e-6 * abs(float(tot3))


# ---- training parity at the benchmark's network size (n0 = 32, 128 x 128): gradients and an 8-step trajectory -----------------
def _lidc_setup(compute_dtype, perturbed, norm=None):
    from oracle import init as oinit
    from phiseg_code_amd.phiseg import phiseg_model
    g, cfg, var_order = load_golden("lidc_phiseg_bn")
    if norm is not None:                              # same net under group / instance norm: conv biases, no moving statistics
        cfg = dict(cfg, norm=norm)
    model = phiseg_model.phiseg(make_config(cfg, compute_dtype), rng_seed=cfg["eps_seed"])
    if norm is not None:
        var_order = [(n, tuple(v.shape)) for n, v in model.graph.variables.items()]
    params = otrain.make_params(var_order, cfg["weight_seed"], torch.float32, perturbed=perturbed)
    x_np, s_np = oinit.synthetic_batch(cfg["B"], cfg["H"], cfg["nlabels"], cfg["data_seed"])
    model.set_weights({k: v.detach().numpy() for k, v in params.items()})
    return cfg, model, params, x_np, s_np


@pytest.mark.parametrize("norm", [None, "group_norm"])
def test_bf16_gradients_n0_32_vs_simulated_bf16_oracle(norm):
    """(norm = None: the golden's batch norm; group norm: conv biases -- their gradient comes out of the norm backward launches in
    closed form -- and the one-launch norm layers on the H <= 16 levels.  Instance norm is not run at this size: its 2 x 2 level
    normalises over FOUR values per statistic and bf16 storage moves the loss itself by 7 % between two evaluations; its kernels
    are covered by test_norm_small_one_launch_layer / test_norm_fwd_bwd and the fp32 golden tiny_phiseg_in.)
    Every variable's gradient of the bf16 training plan at the benchmark's network size (n0 = 32, 128 x 128, batch 2;
    3x3 MFMA forward / data-gradient / filter-gradient kernels, the deferred multi-layer filter-gradient launches and their
    reductions, the 1x1 head filter gradients) against torch autograd of the oracle -- exact (fp32) and with the engine's
    bf16 storage policy simulated (oracle.nets.Ctx.bf16_sim).  Bound per variable: the relative L2 error against the exact
    gradient may be at most 2.5x the simulated policy's own deviation (two bf16 evaluations decorrelate through rounding
    flips; independent errors add in quadrature -> 1.4x expected), with a floor of 3 % for variables the policy happens
    to leave almost untouched; the ratio of a single small variable has a tail, so up to two variables may reach 4x (see below),
    and the mean over all variables is the sharp assertion."""
    from oracle import nets
    tight = os.environ.get("PHX_TEST_TIGHT_GRADIENT_BOUND") == "1" and os.environ.get("PHX_DETERMINISTIC") == "1"
    cfg, model, params, x_np, s_np = _lidc_setup("bf16", perturbed=True, norm=norm)
    xt, st = torch.as_tensor(x_np, dtype=torch.float32), torch.as_tensor(s_np)

    def oracle_grads(sim):
        for v in params.values():
            v.grad = None
        eps = otrain.torch_eps_fn(cfg["eps_seed"], 0, cfg["B"], torch.float32)
        out = nets.elbo(params, xt, st, eps, cfg, training=True, bf16_sim=sim)
        out["loss_tot"].backward()
        return float(out["loss_tot"]), {k: v.grad.detach().double().numpy().copy() for k, v in params.items()
                                        if v.requires_grad and v.grad is not None}
    l_exact, g_exact = oracle_grads(False)
    l_sim, g_sim = oracle_grads(True)
    plan = model.sess.plan_for([model.loss_tot], True, cfg["B"], True)
    plan.set_input("x_input", x_np)
    plan.set_input("s_input", s_np)
    plan.optimize = False
    model.sess.store.set_lr(0.0)                      # the step applies Adam with lr = 0: parameters stay put
    plan.run()
    plan.sync()
    loss = float(plan.fetch(model.loss_tot))
    got = model.sess.store.export(grads=True)
    assert abs(loss - l_exact) <= max(0.05, 3 * abs(l_sim - l_exact) / abs(l_exact)) * abs(l_exact)
    worst, n_checked, tot_e, tot_inh, n_tail = (0.0, None), 0, 0.0, 0.0, 0
    for name, ge in g_exact.items():
        nrm = np.linalg.norm(ge)
        if nrm < 1e-8 * max(1.0, np.sqrt(ge.size)):
            continue                                  # never-consumed branches: zero gradient
        gh = got[name].astype(np.float64).reshape(ge.shape)
        inh = np.linalg.norm(g_sim[name] - ge) / nrm
        e = np.linalg.norm(gh - ge) / nrm
        e_s = np.linalg.norm(gh - g_sim[name]) / nrm
        # (the relative L2 error of a 2-element bias is itself a noisy statistic: a handful of elements get 3x)
        # (... and at batch 2 the H <= 4 levels normalise over 8 - 32 values per channel: a single variable's ratio has a tail --
        # 2.29x observed once in ~10 runs on a 192-element gamma of the 2 x 2 level; the MEAN over all variables below is the
        # sharp assertion)
        # Batch norm at batch 2 is the ill-conditioned case (mean error 0.42 of the simulated policy itself): single variables reached
        # 2.3 - 2.7x in about one run of five, so its per-variable bound only catches gross errors (a wrong gradient has e >= 1) and the
        # sharp per-variable check is the group-norm instance of this test (mean error 0.06, same kernels but for the norm family).
        # Group norm, same build run eight times (round 4): one run had ONE variable at 2.61x (posterior/z4_sigma/W, the 192 -> 2 head of
        # the 2 x 2 level, 0.143 against its usual 0.105 - 0.107; the atomics' summation order moves bf16 rounding flips upstream of it), the
        # other seven runs stayed below 2.0x everywhere.  So: at most two variables (of 474) may lie between 2.5x and 4x, none above 4x.
        # Under PHX_DETERMINISTIC=1 (fixed summation order; test_bf16_gradients_group_norm_deterministic_mode_tight_bound below runs
        # this test that way in a child process) there is no run-to-run tail to allow for: every variable inside 2.5x, none between.
        fac = 4.0 if norm is None else 2.5
        bound = (fac if ge.size >= 64 else fac + 1.0) * max(inh, 0.03)
        hard = bound * (1.0 if (norm is None or tight) else 1.6)
        assert e <= hard and e_s <= hard, (name, e, e_s, inh)
        if e > bound or e_s > bound:
            n_tail += 1
        if e / bound > worst[0]:
            worst = (e / bound, name, e, inh)
        tot_e += e; tot_inh += inh; n_checked += 1
    print("bf16 gradients: %d variables, mean rel. L2 error %.4f (simulated policy itself %.4f), worst %s" %
          (n_checked, tot_e / n_checked, tot_inh / n_checked, worst))
    assert n_checked >= 360                           # 368 live trainable tensors
    assert n_tail <= (0 if tight else 2), n_tail
    assert tot_e <= 1.3 * tot_inh + 0.03 * n_checked  # on average the HIP path deviates no more than the simulated policy itself


def test_bf16_gradients_group_norm_deterministic_mode_tight_bound():
    """The group-norm instance of the test above in the deterministic mode (child process: the mode is read when the engine is
    imported), where the fp32 atomics' summation order cannot move a bf16 rounding flip: the per-variable bound is the original 2.5x
    for EVERY variable -- no 1.6x head-room, no two-variable tail (those are for the default mode's run-to-run draw)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, PHX_DETERMINISTIC="1", PHX_TEST_TIGHT_GRADIENT_BOUND="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.abspath(__file__) + "::test_bf16_gradients_n0_32_vs_simulated_bf16_oracle[group_norm]"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


NSTEP = 8          # (12 in round 2: the oracle trajectory is ~15 s per step on a busy test box -- the fixture was a third of the suite)


@pytest.fixture(scope="module")
def lidc_trajectory():
    """The oracle's 8-step trajectory at n0 = 32, 128 x 128, batch 2 (fp32 torch-CPU, ~3.5 s per step), snapshots of (weights,
    Adam slots) before steps 3 and 7, and the SAME trajectory from an input perturbed by 1e-6 relative: TF1 Adam moves every weight
    by ~lr * sign-like(m / sqrt(v)), so weights whose gradient is round-off-sized change direction with the summation
    order and two exact implementations drift apart -- the perturbed run measures that drift (the chaos band)."""
    from oracle import init as oinit
    g, cfg, var_order = load_golden("lidc_phiseg_bn")
    lr = 2e-5
    x_np, s_np = oinit.synthetic_batch(cfg["B"], cfg["H"], cfg["nlabels"], cfg["data_seed"])
    params = otrain.make_params(var_order, cfg["weight_seed"], torch.float32, perturbed=True)
    p0 = {k: v.detach().clone().numpy() for k, v in params.items()}
    snaps = {3: None, 7: None}
    ref_terms = otrain.train_steps(params, [(x_np, s_np)], cfg, cfg["eps_seed"], lr=lr, n_steps=NSTEP, dtype=torch.float32,
                                   snapshots=snaps)
    ref = [l["total_loss"] for l in ref_terms]
    # the perturbed trajectory costs another 95 s of oracle time per session: its band is a committed fixture (tools/make_chaos_band.py,
    # oracle only), floored at the 1.4e-2 self-drift observed on the 128-thread host (thread count changes the oracle's own sums)
    band_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lidc_chaos_band.npz")
    if os.path.exists(band_file) and int(np.load(band_file)["nstep"]) == NSTEP and float(np.load(band_file)["lr"]) == lr:
        chaos = np.maximum(np.load(band_file)["chaos"], 0.0)
        chaos[-1] = max(chaos[-1], 1.4e-2)
    else:
        params2 = otrain.make_params(var_order, cfg["weight_seed"], torch.float32, perturbed=True)
        ref2 = [l["total_loss"] for l in otrain.train_steps(params2, [(x_np * np.float32(1.000001), s_np)], cfg, cfg["eps_seed"],
                                                             lr=lr, n_steps=NSTEP, dtype=torch.float32)]
        chaos = np.abs(np.array(ref2) - np.array(ref)) / np.abs(ref)
    return dict(cfg=cfg, lr=lr, x=x_np, s=s_np, p0=p0, ref=ref, ref_terms=ref_terms, chaos=chaos, snaps=snaps)


@pytest.mark.parametrize("compute_dtype", ["f32", "bf16"])
def test_loss_curve_n0_32(compute_dtype, lidc_trajectory):
    """8 free-running training steps (ELBO, backward, TF1 Adam, batch-norm moving statistics; eager, hipGraph capture,
    then replay) at n0 = 32, 128 x 128, batch 2 against the oracle's trajectory on identical weights / batch / Philox noise.
    Step 0 must agree to 1e-4 (fp32) -- after that the comparison is bounded by the drift the oracle shows against ITSELF
    under a 1e-6 input perturbation (3x its maximum, floor 1e-3): 1e-3 per step is below that drift at this size (the
    oracle's self-drift reaches ~1e-2 within ten steps).
    bf16: additionally the band between exact and simulated-bf16 oracle evaluation of the first step (x3, floor 2 %).
    The per-step exactness of the step function along the trajectory is checked separately, from the oracle's own
    snapshots (test_single_steps_from_oracle_snapshots_n0_32)."""
    from oracle import nets
    from phiseg_code_amd.phiseg import phiseg_model
    t = lidc_trajectory
    cfg, lr, x_np, s_np, ref = t["cfg"], t["lr"], t["x"], t["s"], t["ref"]
    model = phiseg_model.phiseg(make_config(cfg, compute_dtype), rng_seed=cfg["eps_seed"])
    model.set_weights(t["p0"])
    band0 = 1e-4
    extra = 0.0
    if compute_dtype == "bf16":
        params = {k: torch.as_tensor(v) for k, v in t["p0"].items()}
        xt, st = torch.as_tensor(x_np, dtype=torch.float32), torch.as_tensor(s_np)
        eps = otrain.torch_eps_fn(cfg["eps_seed"], 0, cfg["B"], torch.float32)
        with torch.no_grad():
            ex = float(nets.elbo(params, xt, st, eps, cfg, training=True)["loss_tot"])
            sm_ = float(nets.elbo(params, xt, st, eps, cfg, training=True, bf16_sim=True)["loss_tot"])
        extra = max(0.02, 3 * abs(sm_ - ex) / abs(ex))
        band0 = extra
    keys = sorted(model.loss_dict)
    terms = []
    for _ in range(NSTEP):
        out = model.sess.run([model.train_step] + [model.loss_dict[k] for k in keys],
                             {model.x_inp: x_np, model.s_inp: s_np, model.training_pl: True, model.lr_pl: lr})
        terms.append({k: float(v) for k, v in zip(keys, out[1:])})
    losses = [d["total_loss"] for d in terms]
    rel = np.abs(np.array(losses) - np.array(ref)) / np.abs(ref)
    band = max(1e-3, 3 * float(t["chaos"].max())) + extra
    print("%s %d-step curve: %.1f -> %.1f (oracle %.1f -> %.1f); rel. deviation step 0 %.1e, max %.2e; oracle self-drift max %.2e" %
          (compute_dtype, NSTEP, losses[0], losses[-1], ref[0], ref[-1], rel[0], rel.max(), t["chaos"].max()))
    assert ref[-1] < ref[0] and losses[-1] < losses[0]    # both trajectories descend
    assert rel[0] <= band0, rel[0]
    if compute_dtype == "f32":
        assert (rel[1:] <= band).all(), (rel, band)
    else:
        # Stated bf16 band.  The cross-entropy terms (the well-conditioned part of the ELBO: 60-80 % of it) follow the oracle
        # within 6 % at every step.  The KL terms do not have a tight band at batch 2: the coarsest levels normalise 8 values
        # per channel (2 x 2 maps x 2 images, eps 1e-3 -> rstd up to 31), which amplifies every bf16 rounding flip of the
        # stored activations; two IDENTICAL bf16 runs differ by up to 2-3x in a single KL level at a single step
        # (tools/debug_bf16_curve.py) while their cross-entropy terms agree to 1-3 %.  Bound: the KL sum within a factor 3 per
        # step (one step may spike, see below) and within 25 % in the median over the curve.
        ce = lambda d: sum(v for k, v in d.items() if k.startswith("residual"))
        kl = lambda d: sum(v for k, v in d.items() if k.startswith("KL_"))
        r_ce = np.array([ce(a) / ce(b) for a, b in zip(terms, t["ref_terms"])])
        r_kl = np.array([kl(a) / kl(b) for a, b in zip(terms, t["ref_terms"])])
        print("bf16 / oracle per step: cross-entropy %s  KL %s" % (np.round(r_ce, 3), np.round(r_kl, 2)))
        assert np.abs(r_ce - 1).max() <= 0.06, r_ce
        # Round 6: the per-step factor 3 was the edge of this quantity's own spread -- 28 repetitions of this test on one box (12 with
        # the one-launch batch-norm backward, 16 without: the same picture in both arms) put the KL ratio of step 4 at 0.92 ... 3.89
        # (3.1, 3.2, 3.3 and 3.9 in four of them; every other step 0.8 ... 2.2) while the cross-entropy ratios stayed inside 3.5 %: ONE
        # step may spike (<= 6x), the second largest stays within the factor 3, the median within 25 %.
        srt = np.sort(r_kl)
        assert srt[-1] <= 6.0 and srt[-2] <= 3.0 and srt[0] >= 1 / 3.0 and abs(np.median(r_kl) - 1) <= 0.25, r_kl


def test_single_steps_from_oracle_snapshots_n0_32(lidc_trajectory):
    """The step function along the trajectory, without drift: weights, Adam slots and the step counter of the ORACLE before
    its steps 3 and 7 are loaded into the engine, ONE HIP step (fp32) is taken and compared with the oracle's own step:
    loss to 1e-4; every weight moves like the oracle's (Adam: ~lr per step; bound 10 % of the update's L2 norm per filter)."""
    from phiseg_code_amd.phiseg import phiseg_model
    t = lidc_trajectory
    cfg, lr, x_np, s_np = t["cfg"], t["lr"], t["x"], t["s"]
    model = phiseg_model.phiseg(make_config(cfg, "f32"), rng_seed=cfg["eps_seed"])
    for step, (params, m, v) in sorted(t["snaps"].items()):
        before = {k: p.numpy() for k, p in params.items()}
        model.set_weights(before)
        model.sess.store.load_adam({k: (m[k].numpy(), v[k].numpy()) for k in m})
        model.sess.store.set_step(step)
        _, lt = model.sess.run([model.train_step, model.loss_tot],
                               {model.x_inp: x_np, model.s_inp: s_np, model.training_pl: True, model.lr_pl: lr})
        np.testing.assert_allclose(float(lt), t["ref"][step], rtol=1e-4)
        # the oracle's own update from the same state
        p2 = {k: p.clone().requires_grad_(not k.rsplit("/", 1)[-1].startswith("moving_")) for k, p in params.items()}
        out, grads = otrain.loss_and_grads(p2, torch.as_tensor(x_np), torch.as_tensor(s_np),
                                           otrain.torch_eps_fn(cfg["eps_seed"], step, cfg["B"], torch.float32), cfg)
        from oracle import tf1_ops as T
        got = model.sess.store.export()
        n_bad = n_all = 0
        for k, gk in grads.items():
            if gk is None or not k.endswith("/W"):
                continue
            pn, _, _ = T.adam_tf1_step(params[k], gk, m[k], v[k], step + 1, lr)
            dref = (pn - params[k]).numpy().ravel().astype(np.float64)
            dgot = (got[k] - before[k]).ravel().astype(np.float64)
            if np.linalg.norm(dref) < 1e-9:
                continue
            n_all += 1
            n_bad += int(np.linalg.norm(dgot - dref) > 0.1 * np.linalg.norm(dref))
        assert n_all > 100 and n_bad <= 0.02 * n_all, (step, n_bad, n_all)


def test_shared_encoder_sampling_equals_tiled_batch_fp32():
    """predict()'s one-pass graph (prior encoder once per image, features repeated for the n samples: graph.tile_batch) gives
    the samples the reference's own batching gives -- np.tile(x, [n,1,1,1]) through the full graph (phiseg_model.py:577-585):
    same Philox noise per (image, sample) row, so the soft-max maps agree to fp32 round-off; and predict() returns their mean."""
    g, cfg, var_order, model, params, x_np, s_np = build("tiny_phiseg_bn")
    n = 4
    x2 = x_np[:2]
    _, sm_multi = model.sampling_graph(n)
    fd = {model.training_pl: False, model.x_inp: x2}
    a = model.sess.run(sm_multi, fd)                                       # rows b * n + k
    xt = np.repeat(x2, n, axis=0)                                          # the reference's tiling: rows b * n + k as well
    b = model.sess.run(model.s_out_eval_sm, {model.training_pl: False, model.x_inp: xt})
    assert a.shape == b.shape == (2 * n,) + x2.shape[1:3] + (cfg["nlabels"],)
    np.testing.assert_allclose(a, b, rtol=0, atol=2e-6)
    assert np.abs(a[0] - a[1]).max() > 1e-4                                # the n samples of an image differ
    seg, sm = model.predict(x2, num_samples=n, return_softmax=True)
    np.testing.assert_allclose(sm, a.reshape(2, n, *a.shape[1:]).mean(axis=1), rtol=0, atol=2e-6)
    assert seg.shape == x2.shape[:3] and (seg == sm.argmax(-1)).all()


def test_deterministic_unet_baseline_matches_reference_goldens_fp32():
    """experiments/detunet.py (reference likelihoods.py:10-79 with posteriors.dummy / priors.dummy, no KL term): logits, the
    cross-entropy ELBO, the evaluation instance and three Adam steps against the golden produced by executing the reference's
    det_unet2D, and against the oracle."""
    case = "tiny_detunet_bn"
    g, cfg, var_order, model, params, x_np, s_np = build(case)
    assert set(model.loss_dict) == {"total_loss", "residual_multinoulli_loss_lvl0"}
    s_list, losses = model.sess.run([model.s_out_list, [model.loss_dict[k] for k in sorted(model.loss_dict)]],
                                    {model.x_inp: x_np, model.s_inp: s_np, model.training_pl: True})
    check_tensor(g, "train/s_0", s_list[0], fwd_tol(case, "s"))
    for k, v in zip(sorted(model.loss_dict), losses):
        np.testing.assert_allclose(float(v), float(g["train/loss/" + k]), rtol=fwd_tol(case, "loss"), err_msg=k)
    s_eval, s_out = model.sess.run([model.s_out_eval_list, model.s_out_eval], {model.x_inp: x_np, model.training_pl: False})
    check_tensor(g, "infer/s_eval_0", s_eval[0], fwd_tol(case, "s"))
    check_tensor(g, "infer/s_out_eval", s_out, fwd_tol(case, "s"))
    lr = 1e-5
    ref_losses = otrain.train_steps(params, [(x_np, s_np)], cfg, cfg["eps_seed"], lr=lr, n_steps=3)
    got = []
    for _ in range(3):
        _, lt = model.sess.run([model.train_step, model.loss_tot],
                               {model.x_inp: x_np, model.s_inp: s_np, model.training_pl: True, model.lr_pl: lr})
        got.append(float(lt))
    np.testing.assert_allclose(got, [l["total_loss"] for l in ref_losses], rtol=1e-3)


def test_weight_decay_term_fp32():
    """add_weight_decay (phiseg_model.py:126-128, 290-299): loss_dict['weight_decay'] = weight * sum of tf.nn.l2_loss over the
    'weight_variables' collection -- EVERY conv filter, also those of the never-consumed branches -- and its gradient weight * W."""
    import types
    g, cfg, var_order = load_golden("tiny_phiseg_bn")
    from phiseg_code_amd.phiseg import phiseg_model
    c = make_config(cfg, "f32")
    c.weight_decay_weight = 0.37
    model = phiseg_model.phiseg(c, rng_seed=cfg["eps_seed"])
    params, x_np, s_np = golden_inputs(cfg, var_order, dtype=torch.float64)
    model.set_weights({k: v.detach().numpy() for k, v in params.items()})
    keys = sorted(model.loss_dict)
    out = model.sess.run([model.loss_dict[k] for k in keys], {model.x_inp: x_np, model.s_inp: s_np, model.training_pl: True})
    vals = dict(zip(keys, [float(v) for v in out]))
    wsum = sum(float((v.detach() ** 2).sum()) / 2 for k, v in params.items() if k.endswith("/W"))
    np.testing.assert_allclose(vals["weight_decay"], 0.37 * wsum, rtol=2e-6)
    np.testing.assert_allclose(vals["total_loss"], float(g["train/loss/total_loss"]) + 0.37 * wsum, rtol=5e-4)
    # gradient: the ELBO gradient + 0.37 * W (dead-branch filters get 0.37 * W alone)
    base = phiseg_model.phiseg(make_config(cfg, "f32"), rng_seed=cfg["eps_seed"])
    base.set_weights({k: v.detach().numpy() for k, v in params.items()})
    g0 = _hip_grads(base, cfg, x_np, s_np)
    g1 = _hip_grads(model, cfg, x_np, s_np)
    n_dead = 0
    for k, v in params.items():
        if not k.endswith("/W"):
            continue
        want = g0[k] + 0.37 * v.detach().numpy()
        if np.abs(g0[k]).max() == 0:                 # never-consumed branch: the decay term is the whole gradient, exact
            np.testing.assert_allclose(g1[k], want, rtol=1e-6, atol=1e-9, err_msg=k)
            n_dead += 1
        else:                                        # live filter: ELBO gradient (fp32 conditioning bound, GRAD_RTOL) + decay
            np.testing.assert_allclose(g1[k], want, rtol=0, atol=GRAD_RTOL * max(np.abs(want).max(), 1e-3), err_msg=k)
    assert n_dead > 0


def test_bn_double_update_switch_reproduces_update_ops_fp32():
    """SURVEY.md Q4 / phiseg_model.py:135-141: with control_dependencies(UPDATE_OPS) TF also runs the generation-mode prior and the
    evaluation likelihood every training step, so the moving statistics of the prior / likelihood layers are updated twice (the
    posterior's once).  bn_double_update=True reproduces that; checked against the oracle applying the two updates in turn."""
    from oracle import nets
    from phiseg_code_amd.phiseg import phiseg_model
    g, cfg, var_order = load_golden("tiny_phiseg_bn")
    model = phiseg_model.phiseg(make_config(cfg, "f32"), rng_seed=cfg["eps_seed"], bn_double_update=True)
    params, x_np, s_np = golden_inputs(cfg, var_order, dtype=torch.float64)
    model.set_weights({k: v.detach().numpy() for k, v in params.items()})
    model.sess.run([model.train_step, model.loss_tot], {model.x_inp: x_np, model.s_inp: s_np, model.training_pl: True, model.lr_pl: 0.0})
    got = model.sess.store.export()
    xt, st = torch.as_tensor(x_np, dtype=torch.float64), torch.as_tensor(s_np)
    eps = otrain.torch_eps_fn(cfg["eps_seed"], 0, cfg["B"])
    with torch.no_grad():
        first = nets.elbo(params, xt, st, eps, cfg, training=True)["moving_updates"]
        p2 = dict(params)
        p2.update(first)
        ctx = nets.Ctx(p2, cfg["norm"], True)
        kw = dict(n0=cfg["n0"], resolution_levels=cfg["resolution_levels"], latent_levels=cfg["latent_levels"])
        pz, _, _ = nets.prior_phiseg(ctx, None, xt, True, eps, zdim_0=cfg["zdim0"], **kw)
        nets.likelihood_phiseg(ctx, pz, cfg["image_size"], cfg["nlabels"], **kw)
        second = ctx.moving_updates
    n2 = 0
    for k, v in first.items():
        want = second.get(k, v)
        n2 += int(k in second)
        r = want.numpy()
        np.testing.assert_allclose(got[k], r, rtol=0, atol=1e-4 * max(np.abs(r).max(), 1.0), err_msg=k)
        if k in second and k.endswith("moving_mean"):
            assert np.abs(second[k].numpy() - v.numpy()).max() > 0       # the second update really moved it
    assert n2 > 50 and any(k.startswith("posterior/") and k not in second for k in first)


# ---- bf16 training parity where the benchmark lives: a well-conditioned batch, and config 4's real code path ---------------------
def _bf16_plan_vs_oracle(cfg, norm_name, fac, mean_slack, term_band):
    """One bf16 training step of the engine (lr = 0) against torch autograd of the oracle -- exact fp32 and with the bf16 storage
    policy simulated -- on identical weights, images and Philox noise.  -> the per-term relative deviations.
    Bounds: every loss term within max(term_band, 3x the simulated policy's own deviation) of the exact value; every variable's
    gradient within (fac + 1) x max(simulated deviation, 3 %) of the exact gradient (relative L2) and at least 98 % of them within
    fac x; on average no further from the exact gradient than mean_slack x the simulated policy."""
    from oracle import init as oinit
    from oracle import nets
    from phiseg_code_amd.phiseg import phiseg_model
    model = phiseg_model.phiseg(make_config(cfg, "bf16"), rng_seed=cfg["eps_seed"])
    var_order = [(n, tuple(v.shape)) for n, v in model.graph.variables.items()]
    params = otrain.make_params(var_order, cfg["weight_seed"], torch.float32, perturbed=True)
    x_np, s_np = oinit.synthetic_batch(cfg["B"], cfg["H"], cfg["nlabels"], cfg["data_seed"])
    model.set_weights({k: v.detach().numpy() for k, v in params.items()})
    xt, st = torch.as_tensor(x_np, dtype=torch.float32), torch.as_tensor(s_np)

    def oracle_eval(sim):
        for v in params.values():
            v.grad = None
        eps = otrain.torch_eps_fn(cfg["eps_seed"], 0, cfg["B"], torch.float32)
        out = nets.elbo(params, xt, st, eps, cfg, training=True, bf16_sim=sim)
        out["loss_tot"].backward()
        return ({k: float(v) for k, v in out["loss_dict"].items()},
                {k: v.grad.detach().double().numpy().copy() for k, v in params.items() if v.requires_grad and v.grad is not None})
    t_exact, g_exact = oracle_eval(False)
    t_sim, g_sim = oracle_eval(True)
    keys = sorted(model.loss_dict)
    plan = model.sess.plan_for([model.loss_dict[k] for k in keys], True, cfg["B"], True)
    plan.set_input("x_input", x_np)
    plan.set_input("s_input", s_np)
    model.sess.store.set_lr(0.0)
    plan.run()
    plan.sync()
    terms = {k: float(plan.fetch(model.loss_dict[k])) for k in keys}
    got = model.sess.store.export(grads=True)
    dev = {}
    for k in keys:
        inh = abs(t_sim[k] - t_exact[k]) / abs(t_exact[k])
        dev[k] = abs(terms[k] - t_exact[k]) / abs(t_exact[k])
        assert dev[k] <= max(term_band, 3.0 * inh), (k, terms[k], t_exact[k], t_sim[k])
    n_checked, tot_e, tot_inh, worst, viol = 0, 0.0, 0.0, (0.0, None), []
    for name, ge in g_exact.items():
        nrm = np.linalg.norm(ge)
        if nrm < 1e-8 * max(1.0, np.sqrt(ge.size)):
            continue
        gh = got[name].astype(np.float64).reshape(ge.shape)
        inh = np.linalg.norm(g_sim[name] - ge) / nrm
        e = np.linalg.norm(gh - ge) / nrm
        bound = (fac if ge.size >= 64 else fac + 1.0) * max(inh, 0.03)
        if e > bound:
            viol.append((name, ge.size, round(e, 3), round(inh, 3)))
        # (two bf16 evaluations decorrelate through rounding flips: independent errors add in quadrature -> 1.4x expected, with a tail;
        # observed 2.35x on two 192 / 384-element tensors in one run, none in the next: the hard limit sits one unit above `fac`)
        # (a 2-element bias whose gradient the simulated policy itself misses by 59 % -- posterior/z3_sigma/b: KL and likelihood
        # contributions cancel -- lands anywhere within a few of its own norms: 2.1 / 2.8 in two runs; such entries only count below)
        if ge.size >= 64 or inh <= 0.3:
            assert e <= bound * (fac + 1.0) / fac, (name, e, inh)
        if e / bound > worst[0]:
            worst = (e / bound, name, e, inh)
        tot_e += e; tot_inh += inh; n_checked += 1
    print("beyond %.1fx:" % fac, viol)
    assert len(viol) <= max(2, n_checked // 50), viol      # at most 2 % of the variables beyond `fac` x the simulated policy's deviation
    print("%s bf16 B=%d: %d variables, mean rel. L2 error %.4f (simulated policy %.4f), worst %s; term deviations %s" %
          (norm_name, cfg["B"], n_checked, tot_e / n_checked, tot_inh / n_checked, worst, {k: round(v, 4) for k, v in dev.items()}))
    assert tot_e <= mean_slack * tot_inh + 0.03 * n_checked
    return n_checked


def test_bf16_training_step_batch8_batch_norm_vs_oracle():
    """phiseg_7_5 at the benchmark's network size (n0 = 32, 128 x 128) with a WELL-CONDITIONED batch: at batch 8 the coarsest level's
    batch norm sees 32 values per channel (8 at the golden's batch 2, where every bf16 rounding flip is amplified: per-variable
    bound 4x, KL terms within a factor 3 there).  Here: every loss term -- 5 cross-entropy levels, 5 KL levels, the total -- within
    15 % of the exact oracle (or 3x the simulated bf16 policy's own deviation; measured: cross-entropy levels <= 0.3 %, KL levels
    0.2 - 1 % at levels 0 - 2, 10 % / 19 % at the 4 x 4 / 2 x 2 levels, total 2 %), at least 98 % of the variables' gradients within 2x
    the simulated policy's deviation and all within 3x (measured mean relative L2 error 0.337 against 0.325 for the simulated policy
    itself: bf16 STORAGE, not the kernels, sets these numbers)."""
    g, cfg, _ = load_golden("lidc_phiseg_bn")
    cfg = dict(cfg, B=8)
    n = _bf16_plan_vs_oracle(cfg, "batch norm", fac=2.0, mean_slack=1.3, term_band=0.15)
    assert n >= 360


def test_bf16_training_step_batch64_the_benchmarked_workload_vs_oracle():
    """BASELINE.json config 2 AS BENCHMARKED -- phiseg_7_5, n0 = 32, 128 x 128, bf16, batch norm, BATCH 64: the plan bench.py times
    (pair kernels, k_conv3x3_c32 with the unmaterialised level-0 activations, deferred multi-layer filter gradients, the one-launch
    small-map norm layers, the XCD-banded tile order) -- one training step against the oracle on the host (torch CPU autograd, exact
    fp32 and with the bf16 storage policy simulated; ~1 minute on the GPU box's 256 host cores): every loss term within 15 % (or 3x
    the simulated policy's deviation), the mean gradient error within 1.3x the simulated policy's (measured 1.03 - 1.12x over eight
    repetitions: 0.282 - 0.309 against 0.2755), every variable within 5x.
    The per-variable bound is wider than at batch 8 for a measured reason (tools/grad_error_table.py, tools/latent_forward_table.py,
    DESIGN.md section 4): at this batch the simulated policy's own deviation of the level-0 / level-1 latent heads is as small as
    2 - 4 %, while the KL gradient of such a level is the pixel sum of (mu_q - mu_p) / sigma_p^2, carried by the few pixels with a tiny
    prior sigma_p -- the roundings of a handful of bf16 activations decide it, the same ones in every repetition.  On the golden's seeds
    posterior/prior z1_mu/W, /b came out 9 - 14 % short in each of eight repetitions (3.1 - 3.9x the simulated 3.8 %), with seeds + 2 the
    simulation itself is 22 % short there and the engine follows it to 3.7 %, with seeds + 3 both are within 5 %; the forward means of
    mu_q, mu_p, sigma_p agree with the exact oracle to four digits in all of them."""
    g, cfg, _ = load_golden("lidc_phiseg_bn")
    cfg = dict(cfg, B=64)
    n = _bf16_plan_vs_oracle(cfg, "batch norm, the benchmarked batch", fac=4.0, mean_slack=1.3, term_band=0.15)
    assert n >= 360


def test_bf16_training_step_batch64_group_norm_vs_oracle():
    """BASELINE.json config 2 AS NAMED -- phiseg_7_5, n0 = 32, 128 x 128, bf16, GROUP norm, batch 64: the plan `bench.py --norm group`
    times (per-sample statistics NS = B, the convolution bias kept, conv + bias + group norm + activation in one launch on the maps up
    to 16 x 16, the phase form with the repeated bias, the closed-form bias gradient) -- one training step against torch autograd of
    the oracle, exact fp32 and with the bf16 storage policy simulated, bounds as for the batch-norm plan at this batch.  Group norm
    normalises 16 channels x H x W values per sample whatever the batch, so the coarsest levels are no better conditioned at batch
    64 than at batch 2 (2 x 2 maps: 64 values per statistic); the per-variable factor is the batch-norm test's."""
    g, cfg, _ = load_golden("lidc_phiseg_bn")
    cfg = dict(cfg, B=64, norm="group_norm")
    n = _bf16_plan_vs_oracle(cfg, "group norm, the benchmarked batch", fac=4.0, mean_slack=1.3, term_band=0.15)
    assert n >= 360


def test_bf16_training_step_probunet_n0_32_vs_oracle():
    """BASELINE.json config 4's real code path: prob_unet2D at n0 = 32, 128 x 128, bf16 -- the 1x1 recombination convolutions as the
    centre tap of the 3x3 MFMA kernels, the feature + z concat zero-padded from 38 to 64 channels, the global-average-pool latent
    heads and the broadcast of z over the pixels -- loss terms and every variable's gradient against the oracle (the n0 = 4 fp32 golden
    tiny_probunet_bn runs the direct kernels instead)."""
    cfg = dict(arch="prob_unet2D", norm="batch_norm", n0=32, zdim0=6, H=128, B=4, nlabels=2, latent_levels=1, resolution_levels=7,
               KL_weight=1.0, CE_weight=1.0, exponential_weighting=True, weight_seed=0, eps_seed=42, data_seed=1234)
    cfg["image_size"] = (128, 128, 1)
    n = _bf16_plan_vs_oracle(cfg, "prob_unet2D batch norm", fac=3.0, mean_slack=1.5, term_band=0.15)
    assert n >= 100


def test_bf16_trains_like_fp32_default_mode_eight_seeds():
    """Does the benchmarked precision train, in the mode that is benchmarked?  (round-4 review: the deterministic-mode gate below is
    evidence for a sibling of the benchmarked path, and its tolerance could not see a bias of several per cent.)
    phiseg_7_5 at the benchmark's width (n0 = 32, 128 x 128, batch norm), batch 12 (the reference's batch size, phiseg_7_5.py:40),
    200 training steps (TF1 Adam, lr 1e-3: the reference's schedule, phiseg_model.py:186-207) cycling over eight fixed synthetic
    batches, from the same initial weights on the bf16 engine and on the fp32 engine (the path pinned to the oracle at 1e-4), FOUR
    Philox noise seeds per dtype, DEFAULT (atomics) mode -- eight worker processes side by side (tests/convergence_worker.py).
    Asserted on the means over the last 50 steps: every run has come down > 10x from its first ELBO, and
    |mean(bf16) / mean(fp32) - 1| <= max(2 %, 3 x standard error of the ratio) on the ELBO and on the KL sum separately
    (phiseg_model.py:210-226), max(10 %, 3 x standard error) on the cross-entropy sum.  (Three standard errors, not two: the standard
    error itself is estimated from four runs per arm, and a two-sigma gate on a quantity with a real + 2.9 % offset -- the
    cross-entropy sum, below -- failed in one of the first three full-suite runs of this round: 1.082 +- 0.034.)
    Training is chaotic -- two fp32 runs that differ only in the noise seed end 6-13 % apart -- so four seeds resolve ~10 % on the
    ELBO; the numbers behind the gate are the 32-seed study profiles/r05_convergence_study_32_seeds.txt (tools/convergence_study.py,
    same experiment): ELBO bf16 / fp32 = 1.022 +- 0.035, KL sum 1.010 +- 0.087, no KL level off by more than its standard error
    (the 2 x 2 / 4 x 4 levels 0.98 / 1.04 with the fp32 pre-normalisation tensor of their batch-norm layers, 1.05 / 1.12 without),
    and the one statistically significant difference: the cross-entropy sum at 1.029 +- 0.012 -- bf16 storage of activations and
    gradients leaves a run a few steps behind at step 200 (the loss is still falling).  The four seeds of this test are FIXED, and
    on them the offset is larger than on the 32 (three full-suite runs: 1.082 +- 0.034, 1.071 +- 0.024 and one inside 6 %: only the
    atomics' summation order differs between them), which is why the floor of the cross-entropy sum is 10 %."""
    from tests.convergence_lib import run_all, summarise
    env_was = os.environ.pop("PHX_DETERMINISTIC", None)
    try:
        res = run_all([(dt, so) for so in range(8) for dt in ("f32", "bf16")], 8, 200, 50)
    finally:
        if env_was is not None:
            os.environ["PHX_DETERMINISTIC"] = env_was
    assert len(res) == 16, sorted(res)
    it = next(iter(res.values()))["keys"].index("total_loss")
    for key, run in res.items():
        assert run["finite"], key
        assert run["tail"][it] < 0.1 * run["first"][it], (key, run["first"][it], run["tail"][it])      # it trained
    rows = {name: (r, rse) for name, _, r, rse in summarise(res, ("f32", "bf16"))}
    # round 6 (advisor, round 5): EIGHT seeds per arm since the fp32 arm runs on the fp32 matrix kernels (4.7x faster): the standard
    # error of the cross-entropy ratio is ~2.4 % with eight, so its floor comes down from 10 % to 6 % -- a bias of the size the
    # round-5 KL defect had (+ 40 % on two levels, + 8 % on the ELBO) fails the ELBO / KL rows at three standard errors
    floors = {"ELBO": 0.02, "KL sum (unweighted)": 0.02, "CE sum": 0.06}
    bad = {name: rows[name] for name, fl in floors.items() if not abs(rows[name][0] - 1.0) <= max(fl, 3.0 * rows[name][1])}
    assert not bad, bad


def test_bf16_trains_like_fp32_n0_32_batch12_200_steps():
    """The REPRODUCIBLE record of the same experiment (see the four-seed default-mode gate above): two noise seeds per dtype under
    PHX_DETERMINISTIC=1 -- with ordered reductions the trajectories are bit-reproducible, so the outcome is one fixed set of numbers
    (in the default mode the atomics' summation order makes every repetition a different draw of a chaotic system).  Note that the
    deterministic mode swaps in other kernels for the reductions (ordered filter-gradient folds, two-launch group norm): it is the
    record, the default-mode test is the gate.  Asserted on the means over the last 50 steps: every run has come down > 10x from
    its first ELBO; the bf16 ELBO (mean of the two seeds) lies within max(3 %, 1.5 x the larger seed spread) of the fp32 one, the
    summed cross-entropy within max(12 %, ...) and every cross-entropy level within max(20 %, ...).  The record of this test is ONE
    draw of a chaotic system per build: every change of the engine's arithmetic moves it (round 4: summed cross-entropy 1.017; round 5
    after the fp32 pre-normalisation tensors 1.03, after the contraction-free batch-norm coefficients 1.087, level 3 1.136) -- the
    floors are set so that it documents the number without pretending to resolve what only the 32-seed study resolves (1.029 +- 0.012)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, PHX_DETERMINISTIC="1")
    import tempfile
    procs = []
    for dt in ("f32", "bf16"):
        for so in (0, 1):               # four runs side by side on the one GPU; output to files (nothing can block on a full pipe)
            log = tempfile.TemporaryFile(mode="w+")
            procs.append((dt, so, log, subprocess.Popen([sys.executable, os.path.join(root, "tests", "convergence_worker.py"), "200", "50",
                                                         dt, str(so)], env=env, cwd=root, stdout=log, stderr=subprocess.STDOUT, text=True)))
    rec = {"runs": {"f32": [], "bf16": []}}
    for dt, so, log, pr in procs:
        pr.wait(timeout=1500)
        log.seek(0)
        out = log.read()
        log.close()
        assert pr.returncode == 0, (dt, so, out[-4000:])
        run = json.loads([l for l in out.splitlines() if l.startswith("CONVERGENCE ")][-1][len("CONVERGENCE "):])
        rec["keys"] = run["keys"]
        rec["runs"][dt].append(run)
    keys = rec["keys"]
    it = keys.index("total_loss")
    for dt in ("f32", "bf16"):
        for run in rec["runs"][dt]:
            assert run["finite"]
            assert run["tail"][it] < 0.1 * run["first"][it], (dt, run["first"][it], run["tail"][it])      # it trained
    f = np.array([run["tail"] for run in rec["runs"]["f32"]])
    b = np.array([run["tail"] for run in rec["runs"]["bf16"]])
    ce = [i for i, k in enumerate(keys) if k.startswith("residual_multinoulli_loss")]
    failures = []

    def check(name, fv, bv, floor):
        fm, bm, spread = fv.mean(), bv.mean(), max(abs(fv[0] - fv[1]), abs(bv[0] - bv[1]))
        tol = max(floor * fm, 1.5 * spread)
        print("%-36s f32 %9.1f %9.1f   bf16 %9.1f %9.1f   bf16 / f32 = %.3f   (tolerance %.1f %%)" %
              (name, fv[0], fv[1], bv[0], bv[1], bm / fm, 100 * tol / fm))
        if not abs(bm - fm) <= tol:
            failures.append((name, float(fm), float(bm), float(tol)))
    check("ELBO", f[:, it], b[:, it], 0.03)
    check("cross-entropy, all levels", f[:, ce].sum(axis=1), b[:, ce].sum(axis=1), 0.12)
    for i in ce:
        check(keys[i], f[:, i], b[:, i], 0.20)
    assert not failures, failures


def test_bf16_shared_encoder_sampling_graph_16_samples_192x192_vs_oracle():
    """The graph `bench.py --workload generate` times (BASELINE config 5): ONE 192 x 192 image, 4 classes, 16 Monte-Carlo samples in
    one pass through model.sampling_graph(16) -- prior encoder once, features repeated (graph.tile_batch), latent path + likelihood
    at batch 16, inference-mode batch norm folded into the convolution epilogues (phx_conv3x3_mfma_bf16_affine), bf16, n0 = 32 --
    against the oracle's sampler on the reference's own batching, np.tile(x, [16, 1, 1, 1]) (phiseg_model.py:577-585), with the
    same Philox noise per (image, sample) row.  Bound: RMS deviation of the logits <= 3 % of their range (bf16 storage, see
    test_bf16_path_lidc); the 16 samples differ from one another."""
    from oracle import init as oinit
    from oracle import nets
    from phiseg_code_amd.phiseg import phiseg_model
    n = 16
    cfg = dict(arch="phiseg", norm="batch_norm", n0=32, zdim0=2, H=192, B=1, nlabels=4, latent_levels=5,
               resolution_levels=7, image_size=(192, 192, 1), KL_weight=1.0, CE_weight=1.0, exponential_weighting=True)
    model = phiseg_model.phiseg(make_config(cfg, "bf16"), rng_seed=42)
    var_order = [(nm, v.shape) for nm, v in model.graph.variables.items()]
    params = otrain.make_params(var_order, 0, torch.float64, perturbed=False)
    model.set_weights({k: v.detach().numpy() for k, v in params.items()})
    x_np, _ = oinit.synthetic_batch(1, 192, 4, 1234)
    s_multi, sm_multi = model.sampling_graph(n)
    s_out, sm = model.sess.run([s_multi, sm_multi], {model.x_inp: x_np, model.training_pl: False})
    assert s_out.shape == (n, 192, 192, 4)
    xt = np.repeat(x_np, n, axis=0)
    with torch.no_grad():
        ref = nets.sample(params, torch.as_tensor(xt, dtype=torch.float64), otrain.torch_eps_fn(42, 0, n), dict(cfg, B=n))
    r = ref["s_out_eval"].numpy()
    rms = np.sqrt(((s_out - r) ** 2).mean()) / np.abs(r).max()
    print("shared-encoder sampling graph, 16 samples: RMS deviation %.4f of the logit range" % rms)
    assert rms < 0.03, rms
    assert np.abs(sm.sum(axis=-1) - 1.0).max() < 1e-5
    assert (sm.argmax(-1) == ref["s_out_eval_sm"].numpy().argmax(-1)).mean() > 0.97
    assert np.abs(s_out[0] - s_out[1]).max() > 1e-3 * np.abs(r).max()
