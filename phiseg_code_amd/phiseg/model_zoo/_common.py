"""Pieces shared by the posterior and prior networks (they are the same ladder with different inputs:
reference phiseg/model_zoo/posteriors.py:56-132 and priors.py:51-128; prob_unet2D: posteriors.py:9-52,
priors.py:8-48)."""
from phiseg_code_amd import graph as G
from phiseg_code_amd.tfwrapper import activations as act
from phiseg_code_amd.tfwrapper import layers
from phiseg_code_amd.tfwrapper import normalisation as tfnorm

NET_STREAM = {"posterior": 0, "prior": 1, "prior_gen": 2}      # Philox stream = 16 * net + level


def channel_plan(n0):
    return [n0, 2 * n0, 4 * n0, 6 * n0, 6 * n0, 6 * n0, 6 * n0]


def sample(mu, sigma, net, level):
    """z = mu + sigma * N(0, 1)  (posteriors.py:108,128; priors.py:100,120)."""
    return mu + sigma * G.random_normal(mu, 16 * NET_STREAM[net] + level)


def encoder(net, prefix, widths, levels, norm, training, extra=None):
    """`levels` stages of (2x2 average pool from the previous stage, three 3x3 conv+norm+relu)."""
    feats = []
    extra = extra or {}
    for i in range(levels):
        if i > 0:
            net = layers.averagepool2D(feats[i - 1])
        for t in (1, 2, 3):
            net = layers.conv2D(net, prefix % (i, t), num_filters=widths[i], normalisation=norm, training=training,
                                **extra)
        feats.append(net)
    return feats


def hierarchical_ladder(scope_name, rng_net, inputs, teacher, zdim_0, training, scope_reuse, norm, kwargs):
    """Top-down latent ladder.  `teacher`: list of latents fed downward instead of the net's own samples
    (training-time prior is teacher-forced with posterior samples, priors.py:123-126), or None."""
    n0 = kwargs.get('n0', 32)
    latent_levels = kwargs.get('latent_levels', 5)
    resolution_levels = kwargs.get('resolution_levels', 7)
    widths = channel_plan(n0)
    gap = resolution_levels - latent_levels
    g = G.get_default_graph()
    with g.variable_scope(scope_name) as scope:
        if scope_reuse:
            scope.reuse_variables()
        pre_z = encoder(inputs, 'z%d_pre_%d', widths, resolution_levels, norm, training)
        # sampling path only: n Monte-Carlo samples per image share the encoder (it depends on x alone, priors.py:80-95);
        # its feature maps are repeated n times and everything below runs at batch B * n
        tile = int(kwargs.get('tile_samples', 1))
        if tile > 1:
            pre_z = [G.tile_batch(f, tile) if (i >= gap) else f for i, f in enumerate(pre_z)]
        mu, sigma, z = [None] * latent_levels, [None] * latent_levels, [None] * latent_levels
        # sent[a][b]: latent of level a brought to the resolution of level b (reference z_ups_mat[b][a])
        sent = [[None] * latent_levels for _ in range(latent_levels)]
        for i in reversed(range(latent_levels)):
            if i == latent_levels - 1:
                feat = pre_z[i + gap]
                # the top level's mu keeps conv2D's default 3x3 kernel in the reference (posteriors.py:105)
                mu[i] = layers.conv2D(feat, 'z%d_mu' % i, num_filters=zdim_0, activation=act.identity)
            else:
                for j in reversed(range(i + 1)):
                    u = layers.bilinear_upsample2D(sent[i + 1][j + 1], factor=2, name='ups')
                    for t in (1, 2):
                        u = layers.conv2D(u, name='z%d_ups_to_%d_c_%d' % (i + 1, j + 1, t), num_filters=zdim_0 * n0,
                                          normalisation=norm, training=training)
                    sent[i + 1][j] = u
                feat = G.concat([pre_z[i + gap], sent[i + 1][i]], axis=3, name='concat_%d' % i)
                for t in (1, 2):
                    feat = layers.conv2D(feat, 'z%d_input_%d' % (i, t), num_filters=widths[i], normalisation=norm,
                                         training=training)
                mu[i] = layers.conv2D(feat, 'z%d_mu' % i, num_filters=zdim_0, activation=act.identity,
                                      kernel_size=(1, 1))
            sigma[i] = layers.conv2D(feat, 'z%d_sigma' % i, num_filters=zdim_0, activation=act.softplus,
                                     kernel_size=(1, 1))
            z[i] = sample(mu[i], sigma[i], rng_net, i)
            sent[i][i] = z[i] if teacher is None else teacher[i]
    return z, mu, sigma


def probunet_encoder_head(scope_name, rng_net, inputs, zdim_0, training, scope_reuse, norm, kwargs):
    resolution_levels = kwargs.get('resolution_levels', 7)
    widths = channel_plan(kwargs.get('n0', 32))
    g = G.get_default_graph()
    with g.variable_scope(scope_name) as scope:
        if scope_reuse:
            scope.reuse_variables()
        add_bias = norm is not tfnorm.batch_norm
        enc = encoder(inputs, 'conv_%d_%d', widths, resolution_levels, norm, training, extra=dict(add_bias=add_bias))
        mu_p = layers.conv2D(enc[-1], 'pre_mu', num_filters=zdim_0, kernel_size=(1, 1), activation=act.identity)
        mu = [layers.global_averagepool2D(mu_p)]
        sigma_p = layers.conv2D(enc[-1], 'pre_sigma', num_filters=zdim_0, kernel_size=(1, 1), activation=act.softplus)
        sigma = [layers.global_averagepool2D(sigma_p)]
        z = [sample(mu[0], sigma[0], rng_net, 0)]
    return z, mu, sigma
