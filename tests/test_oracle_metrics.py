"""The metrics oracle (oracle/metrics.py) against the fixtures produced by the reference's own utils functions."""
import os

import numpy as np
import pytest

from oracle import metrics as om
from tests.helpers import METRICS_CASES, metrics_case

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "metrics_cases.npz"), allow_pickle=True)


@pytest.mark.parametrize("k", range(len(METRICS_CASES)))
def test_ged_and_ncc_match_reference_outputs(k):
    seed, N, M, X, Y, C, mode = METRICS_CASES[k]
    sm, gts = metrics_case(seed, N, M, X, Y, C, mode)
    ged = om.generalised_energy_distance(sm.argmax(axis=-1), gts, range(1, C))
    np.testing.assert_allclose(ged, float(GOLD["ged_%d" % k]), rtol=1e-12, atol=1e-12)
    ncc = om.variance_ncc(sm, np.eye(C)[gts])
    if True:
        np.testing.assert_allclose(ncc, float(GOLD["ncc_%d" % k]), rtol=1e-7)   # the reference takes the log in float32


def test_distance_conventions():
    a = np.zeros((4, 4), dtype=np.uint8)
    b = a.copy(); b[0, 0] = 1
    assert om.label_iou_distance(a, a, [1]) == 0.0            # both empty -> IoU 1
    assert om.label_iou_distance(a, b, [1]) == 1.0            # exactly one empty -> IoU 0
    assert om.per_label_dice(a, b, 2) == [2.0 * 15 / 31, 0.0]
