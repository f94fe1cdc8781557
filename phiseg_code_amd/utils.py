"""The two numpy helpers of the reference's top-level utils.py that the hot path uses."""


import numpy as np


def find_floor_in_list(keys, value):
    """Largest key <= value (learning-rate schedule lookup, reference utils.py:70-84) -> (key, index)."""
    best, best_i = None, None
    for i, k in enumerate(keys):
        if k <= value and (best is None or k > best):
            best, best_i = k, i
    if best is None:
        raise ValueError("no schedule key <= %s" % (value,))
    return best, best_i


def list_mean(lst):
    return sum(lst) / float(len(lst))


# ---- validation metrics (reference: utils.py:270-370, phiseg_model.py:586-613), computed by libphx on the device -----------
def _metrics_device(samples_sm, gts, s_ref, nlabels, label0):
    """samples_sm [I, N, X, Y, C] float32 soft-max, gts [I, M, X, Y] uint8, s_ref [I, X, Y] uint8 -> out [I, 2 + 8] float32
    (GED, NCC, Dice per label) on the GPU; host arrays in, host array out."""
    import torch
    from . import runtime as rt
    L = rt.lib()
    I, N, X, Y, C = samples_sm.shape
    M = gts.shape[1]
    dev = torch.device("cuda", torch.cuda.current_device())
    sm = torch.as_tensor(np.ascontiguousarray(samples_sm, dtype=np.float32)).to(dev)
    gt = torch.as_tensor(np.ascontiguousarray(gts, dtype=np.uint8)).to(dev)
    sr = torch.as_tensor(np.ascontiguousarray(s_ref, dtype=np.uint8)).to(dev)
    wsb = int(L.validation_metrics_ws_bytes(I, N, M, X * Y, C))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    out = torch.empty(I, 10, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    L.validation_metrics(sm.data_ptr(), gt.data_ptr(), sr.data_ptr(), ws.data_ptr(), wsb, I, N, M, X * Y, C, label0,
                         out.data_ptr(), st)
    return out.cpu().numpy()


def generalised_energy_distance(sample_arr, gt_arr, nlabels, **kwargs):
    """Reference signature (utils.py:270): sample_arr [N, X, Y], gt_arr [M, X, Y] label maps; `nlabels` = number of labels
    in `label_range` (default range(nlabels)); the reference calls it with nlabels - 1 and label_range = 1 .. nlabels-1."""
    label_range = list(kwargs.get('label_range', range(nlabels)))
    lo, hi = label_range[0], label_range[-1]
    if label_range != list(range(lo, hi + 1)) or len(label_range) != nlabels:
        raise NotImplementedError("label_range must be a contiguous range of nlabels labels")
    C = hi + 1
    if C < 2:
        C = 2
    onehot = np.eye(C, dtype=np.float32)[np.asarray(sample_arr).astype(np.int64)]
    gts = np.asarray(gt_arr).astype(np.uint8)
    out = _metrics_device(onehot[None], gts[None], gts[None, 0], C, lo)
    return float(out[0, 0])


def variance_ncc_dist(sample_arr, gt_arr):
    """Reference signature (utils.py:326): sample_arr [N, X, Y, C] soft-max, gt_arr [M, X, Y, C] one-hot annotations."""
    gts = np.asarray(gt_arr).argmax(axis=-1).astype(np.uint8)
    out = _metrics_device(np.asarray(sample_arr)[None], gts[None], gts[None, 0], sample_arr.shape[-1], 1)
    return float(out[0, 1])


def validation_metrics(samples_sm, gts, s_ref, nlabels):
    """Batched form of what _do_validation computes per image (phiseg_model.py:586-613): samples_sm [I, N, X, Y, C],
    gts [I, M, X, Y], s_ref [I, X, Y] -> (ged [I], ncc [I], dice [I, nlabels])."""
    out = _metrics_device(samples_sm, gts, s_ref, nlabels, 1)
    return out[:, 0], out[:, 1], out[:, 2:2 + nlabels]
