"""Validation metrics on the device: libphx against the oracle and the reference's own outputs."""
import os

import numpy as np
import pytest

from oracle import metrics as om
from tests.helpers import METRICS_CASES, metrics_case

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "metrics_cases.npz"), allow_pickle=True)


@pytest.mark.parametrize("k", range(len(METRICS_CASES)))
def test_device_metrics_match_reference_and_oracle(k):
    from phiseg_code_amd import utils as U
    seed, N, M, X, Y, C, mode = METRICS_CASES[k]
    sm, gts = metrics_case(seed, N, M, X, Y, C, mode)
    s_ref = gts[seed % M]
    ged, ncc, dice = U.validation_metrics(sm[None], gts[None], s_ref[None], C)
    o_ged, o_ncc, o_dice = om.validation_metrics(sm, gts, s_ref, C)
    # integer counts -> the GED is exact up to the final float32 store
    np.testing.assert_allclose(ged[0], float(GOLD["ged_%d" % k]), rtol=0, atol=2e-7)
    np.testing.assert_allclose(dice[0], o_dice, rtol=0, atol=2e-7)
    # float32 logs, double accumulation: 1e-5 on a correlation coefficient
    np.testing.assert_allclose(ncc[0], float(GOLD["ncc_%d" % k]), rtol=0, atol=1e-5)
    np.testing.assert_allclose(ncc[0], o_ncc, rtol=0, atol=1e-5)
    # the reference-signature wrappers
    assert abs(U.generalised_energy_distance(sm.argmax(axis=-1), gts, nlabels=C - 1, label_range=range(1, C)) - o_ged) < 2e-7
    assert abs(U.variance_ncc_dist(sm, np.eye(C)[gts]) - o_ncc) < 1e-5


def test_device_metrics_batch_of_images_lidc_shape():
    """100 soft-max samples, 4 annotators, 128 x 128, two labels (the LIDC validation shape), 3 images in one call."""
    from phiseg_code_amd import utils as U
    sms, gtss, refs = [], [], []
    for seed in (21, 22, 23):
        sm, gts = metrics_case(seed, 100, 4, 128, 128, 2, "plain" if seed != 22 else "empty_fg")
        sms.append(sm); gtss.append(gts); refs.append(gts[seed % 4])
    ged, ncc, dice = U.validation_metrics(np.stack(sms), np.stack(gtss), np.stack(refs), 2)
    for i in range(3):
        o_ged, o_ncc, o_dice = om.validation_metrics(sms[i], gtss[i], refs[i], 2)
        np.testing.assert_allclose(ged[i], o_ged, rtol=0, atol=5e-7)
        np.testing.assert_allclose(ncc[i], o_ncc, rtol=0, atol=2e-5)
        np.testing.assert_allclose(dice[i], o_dice, rtol=0, atol=2e-7)


def test_do_validation_matches_oracle_scoring():
    """phiseg._do_validation (phiseg_model.py:530-660) on synthetic validation data: the device-scored averages equal the
    oracle's scoring of the same samples (the sampling pass is re-run with the same noise step for the comparison)."""
    import importlib
    import types
    from phiseg_code_amd.data import synthetic
    from phiseg_code_amd.phiseg import phiseg_model
    base = importlib.import_module("phiseg_code_amd.phiseg.experiments.phiseg_7_5")
    cfg = types.SimpleNamespace(**{k: getattr(base, k) for k in dir(base) if not k.startswith("_")})
    cfg.compute_dtype, cfg.validation_samples, cfg.num_validation_images = "bf16", 8, 3
    data = synthetic.SyntheticLIDC(cfg, seed=5, n_validation=3)
    model = phiseg_model.phiseg(cfg, rng_seed=3)
    np.random.seed(0)
    res = model._do_validation(data)
    assert np.isfinite([res["loss"], res["dice"], res["ged"], res["ncc"]]).all()
    assert 0.0 <= res["ged"] <= 2.0 and -1.0 <= res["ncc"] <= 1.0 and res["per_structure_dice"].shape == (cfg.nlabels,)
    # same samples again (rewind the Philox noise step), scored by the oracle
    np.random.seed(0)
    model.sess.store.noise_step -= cfg.num_validation_images
    geds, nccs, dices = [], [], []
    for ii in range(3):
        s_gt = data.validation.labels[ii]
        s = s_gt[:, :, np.random.choice(cfg.annotator_range)]
        x_b = np.tile(data.validation.images[ii][None], [cfg.validation_samples, 1, 1, 1])
        sm = model.predict_segmentation_sample(x_b, return_softmax=True)
        g, n, d = om.validation_metrics(sm, np.ascontiguousarray(s_gt.transpose(2, 0, 1)), s, cfg.nlabels)
        geds.append(g); nccs.append(n); dices.append(d)
    np.testing.assert_allclose(res["ged"], np.mean(geds), atol=1e-6)
    np.testing.assert_allclose(res["ncc"], np.mean(nccs), atol=2e-5)
    np.testing.assert_allclose(res["per_structure_dice"], np.mean(dices, axis=0), atol=1e-6)
