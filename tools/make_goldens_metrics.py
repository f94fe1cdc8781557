#!/usr/bin/env python3
"""Write tests/golden/metrics_cases.npz by EXECUTING THE REFERENCE'S OWN utils.generalised_energy_distance and
utils.variance_ncc_dist (build container only; needs /root/reference).  The modules utils.py imports that this image lacks
are supplied as empty stand-ins, except medpy.metric.jc (MedPy 0.4.0 binary Jaccard coefficient, restated from its published
definition).  Only seeds + expected outputs are stored."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
for name in ("nibabel", "skimage", "skimage.measure", "skimage.transform", "medpy", "medpy.metric"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["skimage"].measure, sys.modules["skimage"].transform = sys.modules["skimage.measure"], sys.modules["skimage.transform"]


def jc(a, b):
    a, b = np.asarray(a).astype(bool), np.asarray(b).astype(bool)
    return float(np.count_nonzero(a & b)) / float(np.count_nonzero(a | b))


sys.modules["medpy.metric"].jc = jc
sys.modules["medpy"].metric = sys.modules["medpy.metric"]
sys.path.insert(0, "/root/reference")
import utils as ref_utils                     # noqa: E402  (the reference's utils.py, unmodified)


from tests.helpers import METRICS_CASES as CASES, metrics_case as make_case      # noqa: E402  (seeded inputs, shared with the tests)

if __name__ == "__main__":
    out = {"cases": np.array(CASES, dtype=object)}
    for k, (seed, N, M, X, Y, C, mode) in enumerate(CASES):
        sm, gts = make_case(seed, N, M, X, Y, C, mode)
        s_pred = sm.argmax(axis=-1)
        ged = ref_utils.generalised_energy_distance(s_pred, gts, nlabels=C - 1, label_range=range(1, C))
        onehot = np.eye(C)[gts]
        ncc = float(np.asarray(ref_utils.variance_ncc_dist(sm, onehot)).ravel()[0])
        out["ged_%d" % k], out["ncc_%d" % k] = np.float64(ged), np.float64(ncc)
        print(CASES[k], "GED %.6f NCC %.6f" % (ged, ncc))
    np.savez(os.path.join(ROOT, "tests", "golden", "metrics_cases.npz"), **out, allow_pickle=True)
