"""Oracle restatement of the PHiSeg networks + ELBO (test infrastructure only).

Follows, for structure:
  phiseg/model_zoo/posteriors.py:56-132 (phiseg), 9-52 (prob_unet2D)
  phiseg/model_zoo/priors.py:51-128 (phiseg), 8-48 (prob_unet2D)
  phiseg/model_zoo/likelihoods.py:162-223 (phiseg), 81-159 (prob_unet2D)
  tfwrapper/layers.py:94-145 (conv2D: conv -> [bias] -> norm -> activation)
  phiseg/phiseg_model.py:26-130, 210-311 (wiring, losses)
Only the LIVE graph is evaluated (the never-consumed ``z{a}_ups_to_{b}_c_*`` branches with
b < a, SURVEY.md Q1, are skipped -- they cannot influence any output).

Parameters are a dict keyed by the TF variable names of SURVEY.md Appendix B.  Noise comes
from ``eps_fn(net, level, shape)`` with net in {'posterior', 'prior', 'prior_gen'}.
"""
import torch

from . import tf1_ops as T


def num_channels(n0):
    return [n0, 2 * n0, 4 * n0, 6 * n0, 6 * n0, 6 * n0, 6 * n0]


class Ctx:
    """Carries parameters, the normalisation mode and the training flag through one net."""

    def __init__(self, params, norm="batch_norm", training=True, num_groups=None, bf16_sim=False):
        self.p = params
        self.norm = norm
        self.training = training
        self.num_groups = num_groups
        self.moving_updates = {}     # name -> new value (training-mode batch norm)
        # bf16_sim: model of the engine's bf16 storage policy (NOT reference behaviour): round to bfloat16 wherever
        # the HIP path stores bf16 -- conv outputs before and after norm+activation, pooled / up-sampled maps, the
        # packed 3x3 filters and the (padded) inputs of the MFMA convolutions; heads and latents stay fp32.
        self.bf16_sim = bf16_sim

    def r(self, t):
        return t.to(torch.bfloat16).to(t.dtype) if self.bf16_sim else t

    # tfwrapper/layers.py:94-145
    def conv(self, x, scope, act="relu", normalise=True):
        """normalise=False reproduces call sites that do not pass ``normalisation=`` (the
        mu/sigma/y_lvl/pre_mu/pre_sigma/prediction heads): identity norm, bias kept."""
        p = self.p
        w = p[scope + "/W"]
        head = (not normalise) and act != "relu"
        if self.bf16_sim and not head and w.shape[0] == 3 and w.shape[3] % 32 == 0 and (w.shape[2] % 32 == 0 or w.shape[2] < 32):
            x, w = self.r(x), self.r(w)                # MFMA path: bf16 operands (narrow inputs are padded + rounded)
        y = T.conv2d_same(x, w)
        norm = self.norm if normalise else "identity"
        if norm != "batch_norm":                       # layers.py:126-132
            y = T.bias_add(y, p[scope + "/b"])
        if not head and norm != "identity":
            y = self.r(y)                              # stored pre-norm activation
        if norm == "batch_norm":                       # normalisation.py:145-163
            bn = scope + "/batch_norm/BatchNorm/"
            if self.training:
                y, mean, var_u = T.batch_norm_train(y, p[bn + "gamma"], p[bn + "beta"])
                self.moving_updates[bn + "moving_mean"] = T.batch_norm_moving_update(
                    p[bn + "moving_mean"], mean.detach())
                self.moving_updates[bn + "moving_variance"] = T.batch_norm_moving_update(
                    p[bn + "moving_variance"], var_u.detach())
            else:
                y = T.batch_norm_infer(y, p[bn + "gamma"], p[bn + "beta"],
                                       p[bn + "moving_mean"], p[bn + "moving_variance"])
        elif norm == "group_norm":
            gn = scope + "/group_norm/"
            y = T.group_norm(y, p[gn + "gamma"], p[gn + "beta"], self.num_groups)
        elif norm == "instance_norm":
            inn = scope + "/instance_norm/"
            y = T.instance_norm(y, p[inn + "scale"], p[inn + "offset"])
        if act == "relu":
            y = T.relu(y)
        elif act == "softplus":
            y = T.softplus(y)
        return y if head else self.r(y)

    def pool(self, x):
        return self.r(T.avg_pool_2x2_same(x))

    def up2(self, x, latent=False):
        y = T.resize_bilinear_legacy(x, 2 * x.shape[1], 2 * x.shape[2])
        return y if latent else self.r(y)


def up2(x):
    return T.resize_bilinear_legacy(x, 2 * x.shape[1], 2 * x.shape[2])


# ---------------------------------------------------------------------------------------------
def _phiseg_ladder(ctx, net, x_in, z_teacher, generation_mode, eps_fn, eps_net,
                   zdim_0, n0, resolution_levels, latent_levels):
    """Shared body of posteriors.phiseg (56-132) and priors.phiseg (51-128)."""
    nc = num_channels(n0)
    d = resolution_levels - latent_levels
    pre_z = []
    h = x_in
    for i in range(resolution_levels):
        if i > 0:
            h = ctx.pool(pre_z[i - 1])
        for t in (1, 2, 3):
            h = ctx.conv(h, "%s/z%d_pre_%d" % (net, i, t))
        pre_z.append(h)

    mu, sigma, z = [None] * latent_levels, [None] * latent_levels, [None] * latent_levels
    feed = [None] * latent_levels          # what is sent downward from level i (z_ups_mat[i][i])
    for i in reversed(range(latent_levels)):
        if i == latent_levels - 1:
            src = pre_z[i + d]
        else:
            # live branch only: z_ups_mat[i][i+1] = 2 convs on bilinear-x2 of z_ups_mat[i+1][i+1]
            u = ctx.up2(feed[i + 1], latent=True)
            u = ctx.conv(u, "%s/z%d_ups_to_%d_c_1" % (net, i + 1, i + 1))
            u = ctx.conv(u, "%s/z%d_ups_to_%d_c_2" % (net, i + 1, i + 1))
            src = torch.cat([pre_z[i + d], u], dim=3)
            src = ctx.conv(src, "%s/z%d_input_1" % (net, i))
            src = ctx.conv(src, "%s/z%d_input_2" % (net, i))
        # Q2: the top-level mu is a 3x3 conv, everything else 1x1 (kernel size is in W's shape)
        mu[i] = ctx.conv(src, "%s/z%d_mu" % (net, i), act="identity", normalise=False)
        sigma[i] = ctx.conv(src, "%s/z%d_sigma" % (net, i), act="softplus", normalise=False)
        z[i] = mu[i] + sigma[i] * eps_fn(eps_net, i, tuple(mu[i].shape))
        feed[i] = z[i] if (z_teacher is None or generation_mode) else z_teacher[i]
    return z, mu, sigma


def posterior_phiseg(ctx, x, s_oh, eps_fn, zdim_0=2, n0=32, resolution_levels=7, latent_levels=5):
    x_in = torch.cat([x, s_oh - 0.5], dim=-1)          # posteriors.py:87
    return _phiseg_ladder(ctx, "posterior", x_in, None, True, eps_fn, "posterior",
                          zdim_0, n0, resolution_levels, latent_levels)


def prior_phiseg(ctx, z_list, x, generation_mode, eps_fn, zdim_0=2, n0=32,
                 resolution_levels=7, latent_levels=5):
    return _phiseg_ladder(ctx, "prior", x, z_list, generation_mode, eps_fn,
                          "prior_gen" if generation_mode else "prior",
                          zdim_0, n0, resolution_levels, latent_levels)


def likelihood_phiseg(ctx, z_list, image_size, n_classes, n0=32, resolution_levels=7, latent_levels=5):
    nc = num_channels(n0)
    d = resolution_levels - latent_levels
    post_z = []
    for i in range(latent_levels):
        h = ctx.conv(z_list[i], "likelihood/z%d_post_1" % i)
        h = ctx.conv(h, "likelihood/z%d_post_2" % i)
        for t in range(d):                                  # increase_resolution (170-179)
            h = ctx.conv(ctx.up2(h), "likelihood/preups_%d/z%d_post" % (i, t))
        post_z.append(h)
    post_c = [None] * latent_levels
    post_c[latent_levels - 1] = post_z[latent_levels - 1]
    for i in reversed(range(latent_levels - 1)):
        u = ctx.conv(ctx.up2(post_c[i + 1]), "likelihood/post_z%d_ups_c" % (i + 1))
        h = torch.cat([post_z[i], u], dim=3)
        h = ctx.conv(h, "likelihood/post_c_%d_1" % i)
        post_c[i] = ctx.conv(h, "likelihood/post_c_%d_2" % i)
    s = []
    for i in range(latent_levels):
        s_in = ctx.conv(post_c[i], "likelihood/y_lvl%d" % i, act="identity", normalise=False)
        s.append(T.resize_nearest(s_in, image_size[0], image_size[1]))
    return s


# ---------------------------------------------------------------------------------------------
def _probunet_encoder(ctx, prefix, x_in, n0, resolution_levels):
    nc = num_channels(n0)
    enc = []
    h = x_in
    for ii in range(resolution_levels):
        if ii > 0:
            h = ctx.pool(enc[ii - 1])
        for t in (1, 2, 3):
            h = ctx.conv(h, "%s/conv_%d_%d" % (prefix, ii, t))
        enc.append(h)
    return enc


def _probunet_head(ctx, net, top, eps_fn, eps_net):
    mu = T.global_average_pool(ctx.conv(top, net + "/pre_mu", act="identity", normalise=False))
    sigma = T.global_average_pool(ctx.conv(top, net + "/pre_sigma", act="softplus", normalise=False))
    z = mu + sigma * eps_fn(eps_net, 0, tuple(mu.shape))
    return [z], [mu], [sigma]


def posterior_probunet(ctx, x, s_oh, eps_fn, n0=32, resolution_levels=7, **_):
    enc = _probunet_encoder(ctx, "posterior", torch.cat([x, s_oh - 0.5], dim=-1), n0, resolution_levels)
    return _probunet_head(ctx, "posterior", enc[-1], eps_fn, "posterior")


def prior_probunet(ctx, z_list, x, generation_mode, eps_fn, n0=32, resolution_levels=7, **_):
    # priors.py:8-48 -- ignores z_list / generation_mode, always samples its own z
    enc = _probunet_encoder(ctx, "prior", x, n0, resolution_levels)
    return _probunet_head(ctx, "prior", enc[-1], eps_fn, "prior_gen" if generation_mode else "prior")


def _unet_on_x(ctx, x, n0, resolution_levels):
    enc = _probunet_encoder(ctx, "likelihood/encoder", x, n0, resolution_levels)
    h = enc[-1]
    for jj in range(resolution_levels - 1):
        ii = resolution_levels - jj - 1
        h = torch.cat([ctx.up2(h), enc[ii - 1]], dim=3)      # crop_and_concat: equal sizes here
        for t in (1, 2, 3):
            h = ctx.conv(h, "likelihood/decoder/conv_%d_%d" % (jj, t))
    return h


def posterior_dummy(ctx, x, s_oh, eps_fn, latent_levels=5, **_):
    """posteriors.py:135-138 / priors.py:130-133: constant placeholders, never consumed."""
    zero = [torch.zeros((), dtype=x.dtype)] * latent_levels
    return zero, zero, zero


def prior_dummy(ctx, z_list, x, generation_mode, eps_fn, latent_levels=5, **_):
    zero = [torch.zeros((), dtype=x.dtype)] * latent_levels
    return zero, zero, zero


def likelihood_detunet(ctx, z_list, image_size, n_classes, x, n0=32, resolution_levels=7, **_):
    """likelihoods.py:10-79: the deterministic U-Net (z_list ignored)."""
    h = _unet_on_x(ctx, x, n0, resolution_levels)
    for t in range(3):
        h = ctx.conv(h, "likelihood/recomb_%d" % t)
    return [ctx.conv(h, "likelihood/prediction", act="identity", normalise=False)]


def likelihood_probunet(ctx, z_list, image_size, n_classes, x, n0=32, resolution_levels=7, **_):
    nc = num_channels(n0)
    z = z_list[0]
    h = _unet_on_x(ctx, x, n0, resolution_levels)
    bs, zdim = z.shape
    bz = z.reshape(bs, 1, 1, zdim).expand(bs, image_size[0], image_size[1], zdim)
    h = torch.cat([h, bz], dim=-1)
    for t in range(3):
        h = ctx.conv(h, "likelihood/recomb_%d" % t)
    return [ctx.conv(h, "likelihood/prediction", act="identity", normalise=False)]


ZOO = {
    "phiseg": (posterior_phiseg, prior_phiseg, likelihood_phiseg),
    "prob_unet2D": (posterior_probunet, prior_probunet, likelihood_probunet),
    "det_unet2D": (posterior_dummy, prior_dummy, likelihood_detunet),
}


# ---------------------------------------------------------------------------------------------
def elbo(params, x, s, eps_fn, cfg, training=True, bf16_sim=False):
    """phiseg_model.py:26-130: returns dict with every tensor the build must reproduce.

    cfg keys: arch ('phiseg'|'prob_unet2D'), norm, n0, zdim0, resolution_levels, latent_levels,
    nlabels, image_size, KL_weight, CE_weight, exponential_weighting, num_groups(optional)."""
    post_fn, prior_fn, lik_fn = ZOO[cfg["arch"]]
    L = cfg["latent_levels"]
    kw = dict(n0=cfg["n0"], resolution_levels=cfg["resolution_levels"], latent_levels=L)
    ctx = Ctx(params, cfg["norm"], training, cfg.get("num_groups"), bf16_sim=bf16_sim)
    s_oh = T.one_hot(s, cfg["nlabels"], x.dtype)
    z, mu, sigma = post_fn(ctx, x, s_oh, eps_fn, zdim_0=cfg["zdim0"], **kw)
    pz, pmu, psigma = prior_fn(ctx, z, x, False, eps_fn, zdim_0=cfg["zdim0"], **kw)
    if cfg["arch"] == "phiseg":
        s_list = lik_fn(ctx, z, cfg["image_size"], cfg["nlabels"], **kw)
    else:
        s_list = lik_fn(ctx, z, cfg["image_size"], cfg["nlabels"], x=x, **kw)

    out = dict(z=z, mu=mu, sigma=sigma, prior_mu=pmu, prior_sigma=psigma, prior_z=pz, s=s_list)
    loss_dict = {}
    loss_tot = 0.0
    s_accum = [None] * L
    if cfg.get("CE_weight") is not None:          # add_residual_multinoulli_loss (241-262)
        for ii in reversed(range(L)):
            s_accum[ii] = s_list[ii] if ii == L - 1 else s_accum[ii + 1] + s_list[ii]
            li = T.multinoulli_loss_with_logits(s_oh, s_accum[ii])
            loss_dict["residual_multinoulli_loss_lvl%d" % ii] = li
            loss_tot = loss_tot + cfg["CE_weight"] * li
    if cfg.get("KL_weight") is not None:          # add_hierarchical_KL_div_loss (265-287)
        for ii in reversed(range(L)):
            wl = 4 ** ii if cfg.get("exponential_weighting", True) else 1
            li = wl * T.kl_two_gauss_with_diag_cov(mu[ii], sigma[ii], pmu[ii], psigma[ii])
            loss_dict["KL_divergence_loss_lvl%d" % ii] = li
            loss_tot = loss_tot + cfg["KL_weight"] * li
    loss_dict["total_loss"] = loss_tot
    out.update(s_accum=s_accum, loss_dict=loss_dict, loss_tot=loss_tot,
               moving_updates=ctx.moving_updates)
    return out


def sample(params, x, eps_fn, cfg):
    """phiseg_model.py:61-73,89-111,356-364: prior(generation_mode) -> likelihood -> sum -> softmax,
    inference-mode normalisation."""
    post_fn, prior_fn, lik_fn = ZOO[cfg["arch"]]
    L = cfg["latent_levels"]
    kw = dict(n0=cfg["n0"], resolution_levels=cfg["resolution_levels"], latent_levels=L)
    ctx = Ctx(params, cfg["norm"], False, cfg.get("num_groups"))
    pz, pmu, psigma = prior_fn(ctx, None, x, True, eps_fn, zdim_0=cfg["zdim0"], **kw)
    if cfg["arch"] == "phiseg":
        s_list = lik_fn(ctx, pz, cfg["image_size"], cfg["nlabels"], **kw)
    else:
        s_list = lik_fn(ctx, pz, cfg["image_size"], cfg["nlabels"], x=x, **kw)
    s_out = s_list[-1]
    for i in range(len(s_list) - 1):              # _aggregate_output_list (304-311)
        s_out = s_out + s_list[i]
    return dict(prior_z=pz, prior_mu=pmu, prior_sigma=psigma, s_eval=s_list, s_out_eval=s_out,
                s_out_eval_sm=torch.softmax(s_out, dim=-1))
