"""CPU restatement (numpy) of the reference's validation metrics -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows  utils.py:270-322 (generalised_energy_distance), utils.py:326-370 (variance_ncc_dist), utils.py:103-118 (ncc),
phiseg/phiseg_model.py:586-613 (how _do_validation calls them: labels 1..nlabels-1 for the GED, the Dice of the arg-max of
the mean soft-max against the annotation used for the ELBO).
Third-party pieces the reference imports and this image lacks: medpy.metric.binary.jc / dc (MedPy 0.4.0): Jaccard
|A & B| / |A | B| and Dice 2 |A & B| / (|A| + |B|) of two boolean arrays -- restated here from their published
definitions; the reference only calls them with both arrays non-empty.
Pinned by tests/golden/metrics_cases.npz, produced by tools/make_goldens_metrics.py running the reference's own
utils.generalised_energy_distance / variance_ncc_dist (MedPy's jc supplied as above) on the same seeded inputs."""
import numpy as np


def label_iou_distance(m1, m2, labels):
    """1 - mean over `labels` of the IoU of the two label maps (both empty -> 1, exactly one empty -> 0)."""
    tot = 0.0
    for lbl in labels:
        a, b = (m1 == lbl), (m2 == lbl)
        na, nb = int(a.sum()), int(b.sum())
        if na == 0 and nb == 0:
            tot += 1.0
        elif na == 0 or nb == 0:
            tot += 0.0
        else:
            inter = int((a & b).sum())
            tot += inter / float(na + nb - inter)
    return 1.0 - tot / len(labels)


def generalised_energy_distance(samples, gts, labels):
    """samples [N, X, Y], gts [M, X, Y] integer label maps -> 2 E d(S,Y) - E d(S,S') - E d(Y,Y')."""
    N, M = samples.shape[0], gts.shape[0]
    d_sy = sum(label_iou_distance(samples[i], gts[j], labels) for i in range(N) for j in range(M))
    d_ss = sum(label_iou_distance(samples[i], samples[j], labels) for i in range(N) for j in range(N))
    d_yy = sum(label_iou_distance(gts[i], gts[j], labels) for i in range(M) for j in range(M))
    return 2.0 / (N * M) * d_sy - d_ss / float(N * N) - d_yy / float(M * M)


def variance_ncc(samples_sm, gts_onehot, eps=1e-8):
    """samples_sm [N, X, Y, C] soft-max, gts_onehot [M, X, Y, C] -> mean_j corr(E_ss, E_sy[j]) (population std)."""
    N, M = samples_sm.shape[0], gts_onehot.shape[0]
    logs = np.log(samples_sm.astype(np.float64) + eps)
    mean_seg = samples_sm.astype(np.float64).mean(axis=0)
    e_ss = np.mean([-(mean_seg * logs[i]).sum(axis=-1) for i in range(N)], axis=0)
    out = 0.0
    for j in range(M):
        e_sy = np.mean([-(gts_onehot[j] * logs[i]).sum(axis=-1) for i in range(N)], axis=0)
        a, v = e_ss.ravel(), e_sy.ravel()
        a = (a - a.mean()) / (a.std() * a.size)
        v = (v - v.mean()) / v.std()
        out += float(np.dot(a, v))
    return out / M


def per_label_dice(pred, gt, nlabels):
    """Dice per label of two label maps (both empty -> 1, exactly one empty -> 0), phiseg_model.py:603-613."""
    out = []
    for lbl in range(nlabels):
        a, b = (pred == lbl), (gt == lbl)
        na, nb = int(a.sum()), int(b.sum())
        if na == 0 and nb == 0:
            out.append(1.0)
        elif na == 0 or nb == 0:
            out.append(0.0)
        else:
            out.append(2.0 * int((a & b).sum()) / float(na + nb))
    return out


def validation_metrics(samples_sm, gts, s_ref, nlabels):
    """One validation image as _do_validation scores it -> (GED, NCC, [dice per label])."""
    s_pred = samples_sm.argmax(axis=-1)
    onehot = np.eye(nlabels)[gts]
    ged = generalised_energy_distance(s_pred, gts, range(1, nlabels))
    ncc = variance_ncc(samples_sm, onehot)
    dice = per_label_dice(samples_sm.mean(axis=0).argmax(axis=-1), s_ref, nlabels)
    return ged, ncc, dice
