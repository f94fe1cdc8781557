"""(f)2 device-side mini-batch producer vs the oracle restating data/batch_provider.py:140-272 + utils.py:18-38.

cv2 is not installed (parity unpinned at that level): the oracle restates OpenCV's published warpAffine / resize algorithms;
the CPU tests pin the restatement's self-consistency (identities, symmetries), the GPU tests compare the HIP kernel with it:
label maps bit-equal, images to 1e-6 (same operations in the same order; no FMA contraction on either side)."""
import numpy as np
import pytest

from oracle import augment as oa


def _data(n, X, nlabels, annot, seed):
    rng = np.random.default_rng(seed)
    img = (rng.random((n, X, X), dtype=np.float32) - 0.5).astype(np.float32)
    yy, xx = np.mgrid[0:X, 0:X]
    lab = np.zeros((n, X, X, annot), dtype=np.uint8)
    for i in range(n):
        for a in range(annot):
            for k in range(1, nlabels):
                cy, cx, r = rng.uniform(0.3, 0.7) * X, rng.uniform(0.3, 0.7) * X, rng.uniform(0.08, 0.3) * X / k
                lab[i, ..., a][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = k
    return img, lab


def test_oracle_identities():
    img, lab = _data(1, 32, 3, 1, 0)
    M = oa.rotation_matrix(32, 32, 0.0)
    np.testing.assert_array_equal(oa.warp_affine_linear(img[0], M), img[0])          # angle 0: identity, bit for bit
    np.testing.assert_array_equal(oa.resize_linear(img[0], 32, 32), img[0])          # same size: identity
    # 90 degrees about (cols/2, rows/2): dst(x, y) = src(y, cols - x) -- an exact pixel permutation wherever it stays inside
    r = oa.warp_affine_linear(img[0], oa.rotation_matrix(32, 32, 90.0))
    np.testing.assert_allclose(r[1:, :], np.rot90(img[0])[:-1, :], atol=1e-6)
    # label maps: one-hot interpolation + argmax keeps a constant map constant, pads with label 0 outside
    const = np.full((32, 32), 2, dtype=np.uint8)
    out = np.argmax(oa.warp_affine_linear(oa.onehot(const, 3), oa.rotation_matrix(32, 32, 7.0), np.float64), axis=-1)
    assert set(np.unique(out)) <= {0, 2} and out[16, 16] == 2
    up = oa.resize_linear(img[0][4:20, 6:22], 32, 32)
    assert up.shape == (32, 32) and up.min() >= img[0][4:20, 6:22].min() - 1e-6 and up.max() <= img[0][4:20, 6:22].max() + 1e-6


def test_product_parameter_draws_follow_reference_ranges():
    from phiseg_code_amd.data import augment as pa
    opts = dict(do_rotations=True, do_scaleaug=True, nlabels=2, do_flip_lr=True, do_flip_ud=True)   # the shipped experiment's dict
    n_aug = 0
    for j in range(400):
        d = pa.draw_decisions(1234, 3, j, 128, 128, opts, 4)
        assert not d["fliplr"] and not d["flipud"]                 # 'do_flip_lr' is not a key the provider reads
        assert 0 <= d["annot"] < 4
        if d["augment"]:
            n_aug += 1
            assert -10.0 <= d["angle"] <= 10.0
            assert 98 <= d["r_y"] <= 128 and 0 <= d["p_x"] <= 128 - d["r_y"] and 0 <= d["p_y"] <= 128 - d["r_y"]
        else:
            assert d["angle"] is None and d["r_y"] is None
    assert 140 < n_aug < 260                                       # augment_every_nth = 2: half of the images
    iM = pa.rotation_inverse(128, 128, 7.5)
    np.testing.assert_allclose(np.array(iM).reshape(2, 3), oa.invert_affine(oa.rotation_matrix(128, 128, 7.5)), rtol=0, atol=1e-15)


@pytest.mark.gpu
@pytest.mark.parametrize("X,nlabels", [(128, 2), (64, 4), (192, 3)])
def test_device_augmentation_matches_oracle(X, nlabels):
    import torch
    from phiseg_code_amd.data import augment as pa
    img, lab = _data(6, X, nlabels, 4, X)
    opts = dict(do_rotations=True, do_scaleaug=True, do_fliplr=True, do_flipud=True, nlabels=nlabels, augment_every_nth=2)
    prov = pa.DeviceBatchProvider(img, lab, do_augmentations=True, augmentation_options=opts, num_labels_per_subject=4,
                                  annotator_range=range(4), seed=99, nlabels=nlabels)
    seen_flags = set()
    for _ in range(6):
        x, s = prov.next_batch(5)
        assert x.shape == (5, X, X, 1) and s.shape == (5, X, X) and s.dtype == np.uint8
        for j, (d, src, an) in enumerate(zip(prov.last_decisions, prov.last_indices, prov.last_annotators)):
            ref_x, ref_s = oa.augment_pair(img[src], lab[src, ..., an], d, nlabels)
            assert np.array_equal(s[j], ref_s), (j, d)
            np.testing.assert_allclose(x[j, ..., 0], ref_x, rtol=0, atol=1e-6, err_msg=str(d))
            seen_flags.add((d["angle"] is not None, d["r_y"] is not None, d["fliplr"], d["flipud"]))
    assert len(seen_flags) >= 4                                      # augmented and untouched, flipped and not


@pytest.mark.gpu
def test_provider_feeds_the_training_plan_directly():
    """next_batch_device writes straight into a plan's input buffers; sampling without replacement covers the data set."""
    import torch
    from phiseg_code_amd.data import augment as pa
    img, lab = _data(8, 64, 2, 4, 5)
    prov = pa.DeviceBatchProvider(img, lab, do_augmentations=False, num_labels_per_subject=4, annotator_range=[0, 2], seed=1)
    seen = []
    for _ in range(2):
        x, s = prov.next_batch(4)
        seen += list(prov.last_indices)
        for j, (src, an) in enumerate(zip(prov.last_indices, prov.last_annotators)):
            assert an in (0, 2)
            assert np.array_equal(x[j, ..., 0], img[src]) and np.array_equal(s[j], lab[src, ..., an])
    assert sorted(seen) == list(range(8))
    xb = torch.zeros(4, 64, 64, 1, device="cuda")
    sb = torch.zeros(4, 64, 64, dtype=torch.uint8, device="cuda")
    prov.next_batch_device(4, xb.data_ptr(), sb.data_ptr())
    torch.cuda.synchronize()
    assert float(xb.abs().sum()) > 0


@pytest.mark.gpu
def test_lidc_data_from_the_loaders_hdf5_file():
    """data/lidc_data.py: the three providers built straight from the HDF5 file the reference's loader writes (read by
    data/mini_hdf5.py: tests/golden/lidc_like.hdf5 was written by libhdf5 as lidc_data_loader.py:92-104 does); un-augmented
    batches are rows of the file (float64 images cast to the fp32 feed), augmented training batches keep their shape and range."""
    import os
    import types
    from phiseg_code_amd.data import augment as pa
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    exp = np.load(os.path.join(gold, "lidc_like_expected.npz"))
    cfg = types.SimpleNamespace(num_labels_per_subject=4, annotator_range=range(4), nlabels=2,
                                augmentation_options={'do_flip_lr': True, 'do_flip_ud': True, 'do_rotations': True,
                                                      'do_scaleaug': True, 'nlabels': 2})
    data = pa.lidc_data(cfg, os.path.join(gold, "lidc_like.hdf5"), seed=3)
    x, s = data.validation.next_batch(2)
    for j, (src, an) in enumerate(zip(data.validation.last_indices, data.validation.last_annotators)):
        np.testing.assert_array_equal(x[j, ..., 0], exp["val_images"][src].astype(np.float32))
        np.testing.assert_array_equal(s[j], exp["val_labels"][src, ..., an])
    xt, st = data.train.next_batch(5)
    assert xt.shape == (5, 24, 24, 1) and st.shape == (5, 24, 24) and xt.dtype == np.float32 and st.dtype == np.uint8
    assert float(np.abs(xt).max()) <= 0.5 + 1e-6 and set(np.unique(st)) <= {0, 1}
    assert data.test is not None and data.test.next_batch(3)[0].shape == (3, 24, 24, 1)


@pytest.mark.gpu
def test_reference_data_modules_surface(tmp_path):
    """data/data_switch.py, data/lidc_data.py (exp_config.preproc_folder/data_lidc.hdf5), data/batch_provider.py,
    data/lidc_data_loader.py: same names and call signatures as the reference's data package."""
    import os
    import shutil
    import types
    from phiseg_code_amd.data import batch_provider, data_switch, lidc_data_loader
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    pre = tmp_path / "preproc"
    pre.mkdir()
    cfg = types.SimpleNamespace(num_labels_per_subject=4, nlabels=2, data_root="data_lidc.pickle", preproc_folder=str(pre),
                                augmentation_options={'do_rotations': True, 'do_scaleaug': True, 'nlabels': 2})
    cls = data_switch.data_switch('lidc')
    with pytest.raises(FileNotFoundError, match="data_lidc.hdf5"):
        cls(cfg)                                                # not pre-processed yet: says what to do
    shutil.copy(os.path.join(gold, "lidc_like.hdf5"), str(pre / "data_lidc.hdf5"))
    data = cls(cfg)
    assert list(cfg.annotator_range) == [0, 1, 2, 3]            # back-filled like lidc_data.py:31-33
    assert data.validation.images.shape == (2, 24, 24, 1) and data.validation.labels.shape == (2, 24, 24, 4)
    assert data.test.images.shape[0] == 3 and data.train.next_batch(4)[0].shape == (4, 24, 24, 1)
    h = lidc_data_loader.load_and_maybe_process_data("data_lidc.pickle", str(pre))
    assert sorted(h.keys()) == ["many", "misc", "test", "train", "val"]
    bp = batch_provider.BatchProvider(h["val"]["images"][()], h["val"]["labels"][()], np.arange(2), add_dummy_dimension=True,
                                      num_labels_per_subject=4, annotator_range=range(4))
    assert bp.next_batch(2)[1].shape == (2, 24, 24)
    with pytest.raises(ValueError):
        data_switch.data_switch('acdc')
    assert data_switch.data_switch('synthetic').__name__ == 'SyntheticLIDC'
