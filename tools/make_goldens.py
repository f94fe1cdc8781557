#!/usr/bin/env python3
"""Write tests/golden/*.npz by EXECUTING THE REFERENCE'S OWN zoo + loss code (build container only).

    python tools/make_goldens.py            # needs /root/reference (read-only) -- not on the GPU box

How: ``tools/tf1_shim`` provides a numeric ``tensorflow`` stand-in whose primitives are
``oracle.tf1_ops``; the reference files
    phiseg/model_zoo/{posteriors,priors,likelihoods}.py, tfwrapper/{layers,normalisation,utils}.py,
    phiseg/phiseg_model.py (methods KL_two_gauss_with_diag_cov, multinoulli_loss_with_logits,
    add_residual_multinoulli_loss, add_hierarchical_KL_div_loss, _aggregate_output_list)
are imported UNMODIFIED from /root/reference and run in fp64.  Variables are supplied by
``oracle.init.variable_value`` (keyed by the TF variable name the reference asks for), noise by
``oracle.init.eps_fn_numpy`` -- both re-creatable from the seed on the GPU box, so the fixtures hold
only inputs-by-seed metadata + expected outputs.  Nothing of the reference's source is stored.
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "tf1_shim"))
sys.path.insert(0, REF)
os.environ.setdefault("SGE_GPU", "0")

import shim  # noqa: E402
from oracle import init as oinit  # noqa: E402

shim.install()
from phiseg.model_zoo import likelihoods, posteriors, priors  # noqa: E402  (reference files)
from tfwrapper import normalisation as tfnorm  # noqa: E402         (reference file)
from phiseg import phiseg_model as ref_model  # noqa: E402          (reference file)

NORMS = {"batch_norm": tfnorm.batch_norm, "group_norm": tfnorm.group_norm2D,
         "instance_norm": tfnorm.instance_norm2D}
ZOO = {"phiseg": (posteriors.phiseg, priors.phiseg, likelihoods.phiseg),
       "prob_unet2D": (posteriors.prob_unet2D, priors.prob_unet2D, likelihoods.prob_unet2D),
       "det_unet2D": (posteriors.dummy, priors.dummy, likelihoods.det_unet2D)}

CASES = {
    # name: cfg.  H must be a multiple of 2^(resolution_levels-1) = 64.  The "tiny" cases keep the LIDC geometry
    # (128x128, 7 resolution levels -> 2x2 at the top) with n0 = 4 channels: at H = 64 the top level is 1x1, where
    # instance / batch statistics over 1-2 values are degenerate (var = 0) and fp32 vs fp64 parity is meaningless.
    "tiny_phiseg_bn": dict(arch="phiseg", norm="batch_norm", n0=4, zdim0=2, H=128, B=3, nlabels=2),
    "tiny_phiseg_gn4": dict(arch="phiseg", norm="group_norm", n0=4, zdim0=2, H=128, B=2, nlabels=4),
    "tiny_phiseg_in": dict(arch="phiseg", norm="instance_norm", n0=4, zdim0=2, H=128, B=2, nlabels=2),
    "tiny_probunet_bn": dict(arch="prob_unet2D", norm="batch_norm", n0=4, zdim0=6, H=128, B=3, nlabels=2,
                             latent_levels=1),
    "tiny_phiseg71_bn": dict(arch="phiseg", norm="batch_norm", n0=4, zdim0=2, H=128, B=3, nlabels=2,
                             latent_levels=1),
    # prostate-shaped geometry of BASELINE config 5: 192x192 (maps 96, 48, 24, 12, 6, 3: not powers of two), 4 classes
    "tiny_phiseg_bn_192": dict(arch="phiseg", norm="batch_norm", n0=4, zdim0=2, H=192, B=2, nlabels=4),
    "lidc_phiseg_bn": dict(arch="phiseg", norm="batch_norm", n0=32, zdim0=2, H=128, B=2, nlabels=2,
                           full=False),
    # BASELINE config 1 / the reference's own batch size (phiseg/experiments/phiseg_7_5.py:40), SURVEY 8(c)
    "lidc_phiseg_bn_b12": dict(arch="phiseg", norm="batch_norm", n0=32, zdim0=2, H=128, B=12, nlabels=2,
                               full=False),
    # the deterministic U-Net baseline (experiments/detunet.py): dummy posterior / prior, no KL term
    "tiny_detunet_bn": dict(arch="det_unet2D", norm="batch_norm", n0=4, zdim0=6, H=128, B=3, nlabels=2, latent_levels=1,
                            KL_weight=None),
}


def full_cfg(c):
    c = dict(c)
    c.setdefault("latent_levels", 5)
    c.setdefault("resolution_levels", 7)
    c.setdefault("full", True)
    c.setdefault("KL_weight", 1.0)
    c.update(image_size=(c["H"], c["H"], 1), CE_weight=1.0, exponential_weighting=True,
             weight_seed=0, eps_seed=42, data_seed=1234)
    return c


def run_reference(cfg, training):
    """One eager pass of the reference graph-builder code -> dict of torch tensors (+ shim state)."""
    shim.reset(torch.float64)
    shim.S.training = training
    shim.S.provider = lambda name, shape, init: oinit.variable_value(name, shape, cfg["weight_seed"], True)
    L = cfg["latent_levels"]
    eps_np = oinit.eps_fn_numpy(cfg["eps_seed"], 0, cfg["B"])
    tag = {"prior": "prior"}

    def eps_provider(scope, k, shape):
        net = tag["prior"] if scope == "prior" else scope
        kk = k if scope != "prior" or tag["prior"] == "prior" else k - tag["base"]
        return eps_np(net, L - 1 - kk, tuple(shape))
    shim.S.eps_provider = eps_provider

    x_np, s_np = oinit.synthetic_batch(cfg["B"], cfg["H"], cfg["nlabels"], cfg["data_seed"])
    x = shim.TT(torch.as_tensor(x_np, dtype=torch.float64))
    s_oh = shim.one_hot(torch.as_tensor(s_np), cfg["nlabels"])
    post, prior, lik = ZOO[cfg["arch"]]
    norm = NORMS[cfg["norm"]]
    kw = dict(n0=cfg["n0"], resolution_levels=cfg["resolution_levels"], latent_levels=L, norm=norm)
    # mirrors phiseg/phiseg_model.py:37-98
    z, mu, sigma = post(x, s_oh, cfg["zdim0"], training=training, **kw)
    pz, pmu, psig = prior(z, x, zdim_0=cfg["zdim0"], n_classes=cfg["nlabels"], training=training,
                          generation_mode=False, **kw)
    tag["prior"], tag["base"] = "prior_gen", shim.S.eps_calls.get("prior", 0)
    pzg, pmug, psigg = prior(z, x, zdim_0=cfg["zdim0"], n_classes=cfg["nlabels"], training=training,
                             generation_mode=True, scope_reuse=True, **kw)
    s_list = lik(z, training, n_classes=cfg["nlabels"], image_size=cfg["image_size"], x=x, **kw)
    var_order_after_train_graph = list(shim.S.var_order)
    s_eval = lik(pzg, training, scope_reuse=True, n_classes=cfg["nlabels"], image_size=cfg["image_size"],
                 x=x, **kw)

    # losses: the reference's own methods on a stub `self` (phiseg_model.py:210-311)
    stub = types.SimpleNamespace()
    stub.exp_config = types.SimpleNamespace(latent_levels=L, nlabels=cfg["nlabels"],
                                            residual_multinoulli_loss_weight=cfg["CE_weight"],
                                            KL_divergence_loss_weight=cfg["KL_weight"],
                                            exponential_weighting=cfg["exponential_weighting"])
    for m in ("KL_two_gauss_with_diag_cov", "multinoulli_loss_with_logits",
              "add_residual_multinoulli_loss", "add_hierarchical_KL_div_loss", "_aggregate_output_list"):
        setattr(stub, m, types.MethodType(getattr(ref_model.phiseg, m), stub))
    stub.s_out_list, stub.s_inp_oh = s_list, s_oh
    stub.mu_list, stub.sigma_list = mu, sigma
    stub.prior_mu_list, stub.prior_sigma_list = pmu, psig
    stub.loss_dict, stub.loss_tot = {}, 0
    stub.add_residual_multinoulli_loss()
    if cfg["KL_weight"] is not None:                      # phiseg_model.py:122: hasattr / not None guard
        stub.add_hierarchical_KL_div_loss()
    s_out_eval = stub._aggregate_output_list(list(s_eval), use_softmax=False)
    return dict(x=x_np, s=s_np, z=z, mu=mu, sigma=sigma, prior_mu=pmu, prior_sigma=psig,
                prior_z_gen=pzg, prior_mu_gen=pmug, prior_sigma_gen=psigg, s_list=s_list,
                s_eval=s_eval, s_out_eval=s_out_eval, s_accum=stub.s_accum, loss_dict=stub.loss_dict,
                loss_tot=stub.loss_tot, var_order=var_order_after_train_graph,
                all_vars=dict(shim.S.variables), conv_log=list(shim.S.conv_log),
                moving_updates=dict(shim.S.moving_updates))


def t2n(t):
    return shim._v(t).detach().numpy()


def summarise(name, arr, out, full):
    """Full tensor for small cases; checksum + strided subsample for LIDC-sized ones."""
    if arr.size <= 8192 or (full and arr.size <= 8192):
        out[name] = arr
    else:
        out[name + "@sum"] = np.array(arr.sum())
        out[name + "@abssum"] = np.abs(arr).sum()
        out[name + "@sub8"] = arr[:, ::8, ::8, :].copy()


def main():
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    only = sys.argv[1:]
    for cname, c in CASES.items():
        if only and cname not in only:
            continue
        cfg = full_cfg(c)
        out = {}
        # ---- training-mode pass (ELBO + gradients) ------------------------------------------
        r = run_reference(cfg, training=True)
        L = cfg["latent_levels"]
        for key in ("z", "mu", "sigma", "prior_mu", "prior_sigma"):
            for l in range(L if cfg["arch"] != "det_unet2D" else 0):
                summarise("train/%s_%d" % (key, l), t2n(r[key][l]), out, cfg["full"])
        for l in range(L):
            summarise("train/s_%d" % l, t2n(r["s_list"][l]), out, cfg["full"])
            summarise("train/s_accum_%d" % l, t2n(r["s_accum"][l]), out, cfg["full"])
        for k, v in r["loss_dict"].items():
            out["train/loss/" + k] = t2n(v)
        out["train/loss/total_loss"] = t2n(r["loss_tot"])
        r["loss_tot"].v.backward()
        gn = {}
        for name, var in r["all_vars"].items():
            g = var.v.grad
            gn[name] = None if g is None else [float(g.norm()), float(g.sum())]
        out["train/grad_norm_sum_json"] = np.array(json.dumps(gn))
        if cfg["full"]:
            for name in list(r["all_vars"])[:]:
                g = r["all_vars"][name].v.grad
                if g is not None and g.numel() <= 512:
                    out["train/grad/" + name] = g.numpy()
        mu_keys = sorted(r["moving_updates"])
        out["train/moving_updates_json"] = np.array(json.dumps(
            {k: [float(r["moving_updates"][k].sum()), float(r["moving_updates"][k].abs().sum())] for k in mu_keys}))
        out["meta/var_order_json"] = np.array(json.dumps(r["var_order"]))
        out["meta/conv_log_json"] = np.array(json.dumps(r["conv_log"]))
        # ---- inference-mode pass (sampling path, phiseg_model.py:356-364) ---------------------
        r = run_reference(cfg, training=False)
        for l in range(L):
            if cfg["arch"] != "det_unet2D":
                summarise("infer/prior_z_gen_%d" % l, t2n(r["prior_z_gen"][l]), out, cfg["full"])
            summarise("infer/s_eval_%d" % l, t2n(r["s_eval"][l]), out, cfg["full"])
        summarise("infer/s_out_eval", t2n(r["s_out_eval"]), out, cfg["full"])
        meta = {k: v for k, v in cfg.items() if k not in ("image_size",)}
        out["meta/cfg_json"] = np.array(json.dumps(meta))
        path = os.path.join(outdir, cname + ".npz")
        np.savez_compressed(path, **out)
        print("wrote %s (%d arrays, %.1f KB) total_loss=%.6f" % (
            path, len(out), os.path.getsize(path) / 1024.0, float(out["train/loss/total_loss"])))


if __name__ == "__main__":
    main()
