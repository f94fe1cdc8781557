"""Network zoo with the reference's callable surface (surface B2):
posterior(x, s_oh, zdim_0, training, scope_reuse, norm, **kw) -> (z, mu, sigma)
prior(z_list, x, zdim_0, n_classes, generation_mode, training, scope_reuse, norm, **kw) -> (z, mu, sigma)
likelihood(z_list, training, image_size, n_classes, scope_reuse, norm, **kw) -> [s_l]
Lists are indexed fine -> coarse (index 0 = highest-resolution latent level)."""
