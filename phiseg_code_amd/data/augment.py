"""Device-side mini-batch provider -- counterpart of the reference's data/batch_provider.py (BatchProvider.next_batch,
_select_random_label, _augmentation_function) and data/lidc_data.py (the .train / .validation / .test surface).

The data set is uploaded to HBM once (LIDC train split: ~1 GB of images + ~1 GB of 4-annotator masks -- a fraction of a
per cent of 288 GB); a batch is then ONE kernel (phx_augment_batch: gather, random annotator, rotation, crop-scale, flips)
whose outputs can be the training plan's own input buffers -- no host round trip, where the reference spends the training
thread on numpy / OpenCV per step (phiseg_model.py:193).

Random decisions: the reference draws them from the unseeded global numpy RNG; here every decision of sample j of batch t is
a function of (seed, t, j) through the Philox contract (philox_host), so runs are reproducible and ranks can draw disjoint
shards.  Option names and defaults are the reference's (batch_provider.py:171-177); note that the shipped experiments ask for
'do_flip_lr' / 'do_flip_ud', keys the provider never reads -- flips stay off unless 'do_fliplr' / 'do_flipud'
are given, exactly as there."""
import ctypes
import math

import numpy as np

from phiseg_code_amd import philox_host

ROTATE, SCALE, FLIPLR, FLIPUD = 1, 2, 4, 8
PARAM_DTYPE = np.dtype([("src", "<i4"), ("annot", "<i4"), ("flags", "<i4"), ("r_y", "<i4"), ("p_x", "<i4"), ("p_y", "<i4"),
                        ("iM", "<f8", (6,))])


def rotation_inverse(cols, rows, angle_deg):
    """Inverse (2 x 3, row major) of cv2.getRotationMatrix2D((cols / 2, rows / 2), angle, 1) -- what cv2.warpAffine applies to
    destination coordinates (reference utils.py:18-22)."""
    cx, cy = cols / 2, rows / 2
    a, b = math.cos(math.radians(angle_deg)), math.sin(math.radians(angle_deg))
    m = [[a, b, (1 - a) * cx - b * cy], [-b, a, b * cx + (1 - a) * cy]]
    det = m[0][0] * m[1][1] - m[0][1] * m[1][0]
    d = 1.0 / det if det != 0 else 0.0
    i00, i01, i10, i11 = m[1][1] * d, -m[0][1] * d, -m[1][0] * d, m[0][0] * d
    return [i00, i01, -i00 * m[0][2] - i01 * m[1][2], i10, i11, -i10 * m[0][2] - i11 * m[1][2]]


def draw_decisions(seed, step, sample, X, Y, options, n_annot_choices):
    """The random decisions of one sample (batch_provider.py:131-137, 197-260) from the Philox stream (seed, step, 2000 + sample)."""
    u = philox_host.uniforms(seed, step, 2000 + sample, 8)
    opt = options or {}
    nth = int(opt.get("augment_every_nth", 2))
    d = dict(augment=int(u[0] * nth) == 0, angle=None, r_y=None, p_x=None, p_y=None, fliplr=False, flipud=False,
             annot=min(int(u[7] * n_annot_choices), n_annot_choices - 1))
    if d["augment"]:
        if opt.get("do_rotations", False):
            deg = float(opt.get("rot_degrees", 10.0))
            d["angle"] = -deg + 2.0 * deg * float(u[1])
        if opt.get("do_scaleaug", False):
            offset = int(opt.get("offset", 30))
            r = Y - offset + min(int(u[2] * (offset + 1)), offset)             # random_integers(n_y - offset, n_y)
            d["r_y"] = r
            d["p_x"] = min(int(u[3] * (X - r + 1)), X - r)                       # random_integers(0, n_x - r_y)
            d["p_y"] = min(int(u[4] * (Y - r + 1)), Y - r)
    flipn = max(2, nth)
    if opt.get("do_fliplr", False):
        d["fliplr"] = int(u[5] * flipn) == 0
    if opt.get("do_flipud", False):
        d["flipud"] = int(u[6] * flipn) == 0
    return d


def pack_params(decisions, src_indices, annotators, X, Y):
    rec = np.zeros(len(decisions), dtype=PARAM_DTYPE)
    for i, d in enumerate(decisions):
        flags = 0
        iM = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0]
        if d["angle"] is not None:
            flags |= ROTATE
            iM = rotation_inverse(Y, X, d["angle"])
        if d["r_y"] is not None:
            flags |= SCALE
        if d["fliplr"]:
            flags |= FLIPLR
        if d["flipud"]:
            flags |= FLIPUD
        rec[i] = (src_indices[i], annotators[i], flags, d["r_y"] or 0, d["p_x"] or 0, d["p_y"] or 0, iM)
    return rec


class DeviceBatchProvider:
    """BatchProvider (data/batch_provider.py:19-67) over arrays resident in HBM.  X [N, X, Y] float32 (images - 0.5, never
    re-normalised: lidc_data_loader.py:92), y [N, X, Y, A] uint8."""

    def __init__(self, X, y, indices=None, do_augmentations=False, augmentation_options=None, num_labels_per_subject=1,
                 annotator_range=None, seed=1234, nlabels=2, **kwargs):
        import torch
        from phiseg_code_amd import runtime as rt
        self.L = rt.lib()
        assert self.L.augment_param_bytes() == PARAM_DTYPE.itemsize
        X = np.asarray(X, dtype=np.float32)
        if X.ndim == 4:
            X = X[..., 0]
        y = np.asarray(y, dtype=np.uint8)
        if y.ndim == 3:
            y = y[..., None]
        self.shape = X.shape[1:3]
        self.n_annot = y.shape[3]
        dev = torch.device("cuda", torch.cuda.current_device())
        self.images_dev = torch.as_tensor(np.ascontiguousarray(X)).to(dev)
        self.labels_dev = torch.as_tensor(np.ascontiguousarray(y)).to(dev)
        self.images, self.labels = X[..., None], y                  # host views (validation reads .images / .labels)
        self.indices = np.arange(X.shape[0]) if indices is None else np.asarray(indices)
        self.unused_indices = self.indices.copy()
        self.do_augmentations = do_augmentations
        self.augmentation_options = dict(augmentation_options or {})
        self.nlabels = int(self.augmentation_options.get("nlabels", nlabels))
        self.annotator_range = list(annotator_range if annotator_range is not None else range(num_labels_per_subject))
        self.seed, self.step = int(seed), 0
        self._index_rng = np.random.default_rng([self.seed, 7])
        self._out = {}

    def _draw(self, batch_size):
        if len(self.unused_indices) < batch_size:                    # sampling without replacement across batches (51-55)
            self.unused_indices = self.indices
        idx = np.sort(self._index_rng.choice(self.unused_indices, batch_size, replace=False))
        self.unused_indices = np.setdiff1d(self.unused_indices, idx)
        Xs, Ys = self.shape
        opts = self.augmentation_options if self.do_augmentations else {"augment_every_nth": 1}
        dec = [draw_decisions(self.seed, self.step, j, Xs, Ys, opts if self.do_augmentations else None, len(self.annotator_range))
               for j in range(batch_size)]
        if not self.do_augmentations:
            for d in dec:
                d.update(angle=None, r_y=None, fliplr=False, flipud=False)
        annots = [self.annotator_range[d["annot"]] for d in dec]
        self.step += 1
        self.last_decisions, self.last_indices, self.last_annotators = dec, idx, annots
        return pack_params(dec, idx, annots, Xs, Ys)

    def next_batch_device(self, batch_size, x_ptr=None, s_ptr=None, stream=None):
        """One launch: the batch lands in (x_ptr, s_ptr) -- e.g. a training plan's input buffers -- or in buffers owned by the
        provider (returned as torch tensors)."""
        import torch
        rec = self._draw(batch_size)
        Xs, Ys = self.shape
        par = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(self.images_dev.device)
        if x_ptr is None:
            key = batch_size
            if key not in self._out:
                self._out[key] = (torch.empty(batch_size, Xs, Ys, 1, dtype=torch.float32, device=par.device),
                                  torch.empty(batch_size, Xs, Ys, dtype=torch.uint8, device=par.device))
            xo, so = self._out[key]
            x_ptr, s_ptr = xo.data_ptr(), so.data_ptr()
        else:
            xo = so = None
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        self.L.augment_batch(self.images_dev.data_ptr(), self.labels_dev.data_ptr(), par.data_ptr(), x_ptr, s_ptr, batch_size,
                             Xs, Ys, self.n_annot, self.nlabels, st)
        self._keep = par
        return xo, so

    def next_batch(self, batch_size):
        """The reference's call (phiseg_model.py:193): host arrays x [B, X, Y, 1] float32, s [B, X, Y] uint8."""
        import torch
        xo, so = self.next_batch_device(batch_size)
        torch.cuda.synchronize()
        return xo.cpu().numpy(), so.cpu().numpy()


class lidc_data:
    """data/lidc_data.py: .train (augmented, random annotator), .validation, .test providers built from the arrays the
    reference keeps in HDF5 (lidc_data_loader.py:92-104: <split>/images [N,128,128] float, <split>/labels [N,128,128,4] uint8).
    `source`: a dict {'train': {'images':..., 'labels':...}, 'val': ..., 'test': ...} of arrays, the path of an .npz with keys
    'train_images', 'train_labels', ..., or the reference's HDF5 file data_lidc.hdf5 (h5py when importable, else data/mini_hdf5.py)."""

    def __init__(self, exp_config, source, seed=1234):
        data = self._load(source)
        ar = getattr(exp_config, "annotator_range", range(exp_config.num_labels_per_subject))
        common = dict(num_labels_per_subject=exp_config.num_labels_per_subject, annotator_range=ar, nlabels=exp_config.nlabels)
        self.train = DeviceBatchProvider(data["train"]["images"], data["train"]["labels"], do_augmentations=True,
                                         augmentation_options=exp_config.augmentation_options, seed=seed, **common)
        self.validation = DeviceBatchProvider(data["val"]["images"], data["val"]["labels"], seed=seed + 1, **common)
        self.test = (DeviceBatchProvider(data["test"]["images"], data["test"]["labels"], seed=seed + 2, **common)
                     if "test" in data else None)

    @staticmethod
    def _load(source):
        if isinstance(source, dict):
            return source
        if str(source).endswith(".npz"):
            z = np.load(source)
            return {sp: dict(images=z[sp + "_images"], labels=z[sp + "_labels"]) for sp in ("train", "val", "test")
                    if sp + "_images" in z.files}
        try:
            import h5py as h5
        except ImportError:                  # not installed here: the package's own reader of the loader's HDF5 layout
            from phiseg_code_amd.data import mini_hdf5 as h5
        with h5.File(source, "r") as f:
            return {sp: dict(images=f[sp]["images"][()], labels=f[sp]["labels"][()]) for sp in ("train", "val", "test") if sp in f}
