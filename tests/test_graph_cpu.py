"""Host logic without a GPU: the symbolic graph built by our tfwrapper / model_zoo / phiseg_model must
create exactly the variables (names, shapes, creation order) the reference's code creates -- the fixture's
var_order was recorded while executing the reference zoo (tools/make_goldens.py)."""
import importlib
import types

import numpy as np
import pytest

from tests.helpers import load_golden

from phiseg_code_amd import graph as G
from phiseg_code_amd.phiseg import phiseg_model
from phiseg_code_amd.phiseg.model_zoo import likelihoods, posteriors, priors
from phiseg_code_amd.tfwrapper import normalisation as tfnorm

NORMS = {"batch_norm": tfnorm.batch_norm, "group_norm": tfnorm.group_norm2D, "instance_norm": tfnorm.instance_norm2D}


def make_config(cfg, compute_dtype="f32"):
    base = importlib.import_module("phiseg_code_amd.phiseg.experiments.phiseg_7_5")
    c = types.SimpleNamespace(**{k: getattr(base, k) for k in dir(base) if not k.startswith("_")})
    arch = cfg["arch"]
    if arch == "det_unet2D":              # the deterministic baseline: dummy latent nets (experiments/detunet.py)
        c.posterior, c.prior, c.likelihood = posteriors.dummy, priors.dummy, likelihoods.det_unet2D
    else:
        c.posterior, c.prior, c.likelihood = getattr(posteriors, arch), getattr(priors, arch), getattr(likelihoods, arch)
    c.layer_norm = NORMS[cfg["norm"]]
    c.latent_levels, c.resolution_levels = cfg["latent_levels"], cfg["resolution_levels"]
    c.n0, c.zdim0, c.nlabels = cfg["n0"], cfg["zdim0"], cfg["nlabels"]
    c.image_size = (cfg["H"], cfg["H"], 1)
    c.batch_size = cfg["B"]
    if cfg.get("KL_weight", 1.0) is None:
        c.KL_divergence_loss_weight = None
    c.compute_dtype = compute_dtype
    return c


@pytest.mark.parametrize("case", ["tiny_phiseg_bn", "tiny_phiseg_gn4", "tiny_phiseg_in", "tiny_probunet_bn",
                                  "tiny_phiseg71_bn", "tiny_phiseg_bn_192", "lidc_phiseg_bn", "tiny_detunet_bn"])
def test_variables_match_reference_trace(case):
    g, cfg, var_order = load_golden(case)
    model = phiseg_model.phiseg(make_config(cfg))
    ours = [(n, tuple(v.shape)) for n, v in model.graph.variables.items()]
    assert ours == [(n, tuple(s)) for n, s in var_order]
    # 11 loss_dict entries for 5 latent levels, same keys as the reference (phiseg_model.py:253,279,130)
    L = cfg["latent_levels"]
    want = {"total_loss"} | {"residual_multinoulli_loss_lvl%d" % i for i in range(L)} | \
           ({"KL_divergence_loss_lvl%d" % i for i in range(L)} if cfg.get("KL_weight", 1.0) is not None else set())
    assert set(model.loss_dict) == want


def test_phiseg_7_5_parameter_count_and_scope_reuse():
    base = importlib.import_module("phiseg_code_amd.phiseg.experiments.phiseg_7_5")
    model = phiseg_model.phiseg(base)
    n_train = sum(v.size for v in model.graph.variables.values() if v.trainable)
    assert n_train == 18706994           # 18 706 994 trainable parameters incl. never-used branches
    with pytest.raises(ValueError):      # re-creating an existing variable without scope_reuse must fail
        posteriors.phiseg(model.x_inp, model.s_inp_oh, 2, training=True)


def test_experiment_config_surface():
    want = ["experiment_name", "log_dir_name", "posterior", "likelihood", "prior", "layer_norm",
            "use_logistic_transform", "latent_levels", "resolution_levels", "n0", "zdim0", "max_channel_power",
            "data_identifier", "preproc_folder", "data_root", "dimensionality_mode", "image_size", "nlabels",
            "num_labels_per_subject", "augmentation_options", "optimizer", "lr_schedule_dict", "deep_supervision",
            "batch_size", "num_iter", "annotator_range", "KL_divergence_loss_weight", "exponential_weighting",
            "residual_multinoulli_loss_weight", "do_image_summaries", "rescale_RGB", "validation_frequency",
            "validation_samples", "num_validation_images", "tensorboard_update_frequency"]
    for name in ["phiseg_7_5", "phiseg_7_1", "phiseg_7_5_1annot", "phiseg_7_1_1annot", "probunet", "probunet_1annot", "detunet"]:
        m = importlib.import_module("phiseg_code_amd.phiseg.experiments." + name)
        for k in want:
            assert hasattr(m, k), (name, k)
    pu = importlib.import_module("phiseg_code_amd.phiseg.experiments.probunet")
    assert pu.zdim0 == 6 and pu.latent_levels == 1 and pu.posterior is posteriors.prob_unet2D
    du = importlib.import_module("phiseg_code_amd.phiseg.experiments.detunet")
    assert du.likelihood is likelihoods.det_unet2D and du.posterior is posteriors.dummy and du.KL_divergence_loss_weight is None


def test_c_abi_exports_every_declared_symbol():
    from phiseg_code_amd import runtime as rt
    protos = rt.parse_header()
    assert len(protos) >= 55
    lib = rt.lib()                       # binds every prototype; a missing symbol raises AttributeError
    assert lib.abi_version() == 1


def test_product_library_has_no_settable_kernel_policy():
    """libphx.so exports exactly what include/phx.h declares -- and none of the test build's policy setters (include/phx_debug.h):
    the kernel-selection policy is a compile-time constant there.  libphx_dbg.so (same sources, -DPHX_DEBUG_BUILD) exports both sets."""
    import ctypes
    from phiseg_code_amd import runtime as rt
    dbg = rt.parse_header(rt.DEBUG_HEADER)
    assert {"phx_debug_conv_policy", "phx_debug_pair_kernel_grid"} <= set(dbg) and all(n.startswith("phx_debug_") for n in dbg)
    assert not (set(dbg) & set(rt.parse_header()))
    dll = ctypes.CDLL(rt.LIB_PATH)
    for name in dbg:
        assert not hasattr(dll, name), name + " is exported by the product library"
    d = rt.debug_lib()                   # binds phx.h + phx_debug.h: a missing symbol raises
    assert d.abi_version() == 1 and callable(d.debug_conv_policy)


def test_dropin_aliases():
    import phiseg_code_amd
    phiseg_code_amd.install_dropin_aliases()
    import tfwrapper.layers as L2
    from phiseg_code_amd.tfwrapper import layers as L1
    assert L1 is L2
    # phiseg_train.py:7 `from data.data_switch import data_switch`, :11 `import utils`
    from data.data_switch import data_switch
    from phiseg_code_amd.data import data_switch as ds
    assert data_switch is ds.data_switch
    import utils
    assert hasattr(utils, "generalised_energy_distance") and hasattr(utils, "variance_ncc_dist")


def test_launch_plans_host_side():
    """Host-side plan queries of libphx (no GPU needed): which layers the one-launch kernels take at the benchmark's shapes, and
    that the concat-free filter gradient sizes its workspace for the tile split it will use."""
    import ctypes
    from phiseg_code_amd import runtime as rt
    L = rt.lib()
    # conv + group / instance norm in one launch: maps that fit one pixel tile, 16-channel groups or per-channel statistics
    assert L.conv3x3_fgn_supported(64, 16, 16, 192, 192, 12) in (32, 64)
    assert L.conv3x3_fgn_supported(64, 4, 4, 64, 192, 192) in (32, 64)             # instance norm
    assert L.conv3x3_fgn_supported(64, 32, 32, 128, 128, 8) == 0                   # a sample spans several tiles
    assert L.conv3x3_fgn_supported(64, 8, 8, 192, 192, 6) == 0                     # 32-channel groups
    assert L.conv3x3_fgn_supported(64, 12, 12, 192, 192, 12) == 0                  # not a power of two
    # concat-free filter gradient: K1 % 64 != 0 forces 32-channel input tiles -> more (smaller) partial filters, same total elements
    same = L.conv3x3_wgrad_ws_bytes_dual(64, 64, 64, 128, 192, 64)
    assert same == L.conv3x3_wgrad_ws_bytes(64, 64, 64, 128, 192)
    plan_a, plan_b = (ctypes.c_int * 6)(), (ctypes.c_int * 6)()
    L.conv3x3_wgrad_reduce_plan(64, 128, 128, 64, 128, plan_a)
    L.conv3x3_wgrad_reduce_plan_dual(64, 128, 128, 64, 128, 32, plan_b)
    assert plan_a[2] == 64 and plan_b[2] == 32 and plan_a[3] == plan_b[3] == 64    # tci 64 -> 32, tco unchanged
    assert L.conv3x3_wgrad_ws_bytes_dual(64, 128, 128, 64, 128, 32) > 0
    # the large-map kernels count 16 x 32-pixel tiles, the 256-pixel kernel its own; atomic statistics only with few tiles
    assert L.conv3x3_mfma_bf16_tiles(64, 128, 128, 128, 128) == 64 * 8 * 4
    assert L.conv3x3_mfma_bf16_tiles(64, 16, 16, 192, 192) == 64
    assert L.conv3x3_mfma_stats_atomic_supported(64, 16, 16, 192, 192) == 1 and L.conv3x3_mfma_stats_atomic_supported(64, 128, 128, 32, 32) == 0


@pytest.mark.parametrize("shape", [(64, 16, 16, 384, 384), (64, 16, 16, 768, 384), (64, 32, 32, 384, 384), (12, 16, 16, 512, 512),
                                   (64, 16, 16, 384, 192), (64, 64, 64, 192, 192), (64, 32, 32, 128, 128)])
def test_deferred_filter_gradient_job_fits_the_stand_alone_workspace(shape):
    """A deferred filter-gradient job (phx_conv3x3_wgrad_multi_job_dual with the engine's blocks_target) must accept the workspace
    phx_conv3x3_wgrad_ws_bytes sizes from the stand-alone plan -- with >= 33 64 x 64 channel blocks (n0 = 64 PHiSeg's 384 -> 384
    layers, any 512-wide layer) the XCD round-up of the slice count once exceeded the stand-alone split.  Host-side only."""
    import ctypes
    from phiseg_code_amd import engine
    from phiseg_code_amd import runtime as rt
    L = rt.lib()
    B, H, W, K, N = shape
    wsb = int(L.conv3x3_wgrad_ws_bytes(B, H, W, K, N))
    assert wsb > 0
    jb = ctypes.create_string_buffer(int(L.conv3x3_wgrad_multi_job_bytes()))
    info = (ctypes.c_int * 9)()
    fake = 1 << 20                                             # (pointers are only recorded, never dereferenced on the host)
    L.conv3x3_wgrad_multi_job_dual(fake, None, 0, fake, fake, fake, wsb, B, H, W, K, N, engine._WGRAD_DEFER_BLOCKS, 0, jb, info)
    assert info[0] != 0 and info[1] > 0 and info[3] == 1       # deferred, with blocks, through the workspace
    nslice, tci, tco = info[4], info[5], info[6]
    assert (K // tci) * (N // tco) * nslice * 9 * tci * tco * 4 <= wsb
