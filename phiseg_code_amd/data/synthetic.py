"""Synthetic LIDC-shaped batches (there is no dataset on the benchmark box).

Shapes and value ranges follow the reference's data pipeline: images are stored as ``image - 0.5`` and never
re-normalised (data/lidc_data_loader.py:92, data/batch_provider.py:117-118), labels are uint8 in [0, nlabels).
``data.train.next_batch(B)`` mirrors the call the reference's train loop makes (phiseg_model.py:193)."""
import numpy as np


def make_batch(batch, size, nlabels, rng):
    h = w = size
    x = (rng.random((batch, h, w, 1), dtype=np.float32) - 0.5).astype(np.float32)
    s = np.zeros((batch, h, w), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    for b in range(batch):
        if rng.random() < 0.25:            # annotators often disagree on presence: 25 % empty masks
            continue
        cy, cx = rng.uniform(0.3125 * h, 0.6875 * h), rng.uniform(0.3125 * w, 0.6875 * w)
        ry, rx = rng.uniform(3, 20) * h / 128.0, rng.uniform(3, 20) * w / 128.0
        for k in range(1, nlabels):
            f = 1.0 - (k - 1) / float(max(nlabels - 1, 1))
            s[b][((yy - cy) / (ry * f)) ** 2 + ((xx - cx) / (rx * f)) ** 2 <= 1.0] = k
    return x, s


def philox_batch(batch, size, nlabels, seed=1234, step=0, sample_offset=0):
    """The same kind of batch from the build's Philox streams (philox_host): sample g = sample_offset + b of (seed, step)
    depends on nothing else, so every rank of a data-parallel job -- and the CPU baseline -- can draw exactly its shard.
    x [B,H,W,1] ~ U(-0.5, 0.5); s: nested filled ellipses (label k inside the k-th), 25 % empty masks."""
    from phiseg_code_amd import philox_host
    h = w = size
    yy, xx = np.mgrid[0:h, 0:w]
    xs, ss = [], []
    for b in range(batch):
        u = philox_host.uniforms(seed, step, 1000 + b + sample_offset, h * w + 8)
        xs.append((u[:h * w] - 0.5).astype(np.float32).reshape(h, w, 1))
        p = u[h * w:]
        s = np.zeros((h, w), dtype=np.uint8)
        if p[0] >= 0.25:
            cy, cx = (0.3125 + 0.375 * p[1]) * h, (0.3125 + 0.375 * p[2]) * w
            ry, rx = (3 + 17 * p[3]) * h / 128.0, (3 + 17 * p[4]) * w / 128.0
            for k in range(1, nlabels):
                f = 1.0 - (k - 1) / float(nlabels - 1) if nlabels > 2 else 1.0
                s[((yy - cy) / (ry * f)) ** 2 + ((xx - cx) / (rx * f)) ** 2 <= 1.0] = k
        ss.append(s)
    return np.stack(xs), np.stack(ss)


class _Split:
    def __init__(self, size, nlabels, seed):
        self.size, self.nlabels = size, nlabels
        self.rng = np.random.default_rng(seed)

    def next_batch(self, batch_size):
        return make_batch(batch_size, self.size, self.nlabels, self.rng)


class _ValidationSplit(_Split):
    """``.images`` [n, X, Y, 1] and ``.labels`` [n, X, Y, annotators] as phiseg_model._do_validation reads them
    (data/batch_provider.py keeps the validation set as arrays): every image with `annotators` differing masks."""

    def __init__(self, size, nlabels, seed, n_images=8, annotators=4):
        super().__init__(size, nlabels, seed)
        xs, ls = [], []
        for _ in range(n_images):
            x, _ = make_batch(1, size, nlabels, self.rng)
            _, s = make_batch(annotators, size, nlabels, self.rng)
            xs.append(x[0])
            ls.append(np.transpose(s, (1, 2, 0)))
        self.images, self.labels = np.stack(xs), np.stack(ls)


class SyntheticLIDC:
    """Object with the ``.train`` / ``.validation`` surface of the reference's ``lidc_data`` (data/lidc_data.py)."""

    def __init__(self, exp_config, seed=1234, n_validation=8):
        self.train = _Split(exp_config.image_size[0], exp_config.nlabels, seed)
        self.validation = _ValidationSplit(exp_config.image_size[0], exp_config.nlabels, seed + 1, n_validation,
                                           getattr(exp_config, "num_labels_per_subject", 4))
