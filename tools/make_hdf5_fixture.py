"""Generates tests/golden/lidc_like.hdf5 (+ .npz of the same arrays): a small HDF5 file written by real libhdf5 exactly the way the
reference's data/lidc_data_loader.py:46-104 (prepare_data) writes data_lidc.hdf5 -- h5py.File(path, "w"), create_group per split,
create_dataset(name, data=...) for uids (int), labels (uint8 [N,X,Y,4]) and images (float [N,X,Y]) -- so that the pure-Python reader
(phiseg_code_amd/data/mini_hdf5.py) is pinned against the real format.  Run with an interpreter that has h5py; in this image:

    /opt/conda/bin/python3.9 tools/make_hdf5_fixture.py

(The many-entries group exercises B-tree nodes with several symbol-table nodes, which the three-dataset groups never reach.)"""
import os

import h5py
import numpy as np

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
rng = np.random.default_rng(20240928)
X = 24
arrays = {}
f = h5py.File(os.path.join(out, "lidc_like.hdf5"), "w")
groups = {}
for tt in ["train", "test", "val"]:                       # creation order of the reference
    groups[tt] = f.create_group(tt)
for tt, n in (("test", 3), ("train", 7), ("val", 2)):
    img = rng.random((n, X, X)) - 0.5
    lbl = (rng.random((n, X, X, 4)) < 0.1).astype(np.uint8)
    uid = rng.integers(-2 ** 62, 2 ** 62, n)
    groups[tt].create_dataset("uids", data=np.asarray(uid, dtype=int))
    groups[tt].create_dataset("labels", data=np.asarray(lbl, dtype=np.uint8))
    groups[tt].create_dataset("images", data=np.asarray(img, dtype=float))
    arrays.update({tt + "_uids": uid, tt + "_labels": lbl, tt + "_images": img})
many = f.create_group("many")
for i in range(150):
    a = np.arange(i % 5 + 1, dtype=np.float32) * i
    many.create_dataset("entry_%03d" % i, data=a)
    arrays["many_%03d" % i] = a
misc = f.create_group("misc")
misc.create_dataset("scalar", data=np.float64(3.25))
misc.create_dataset("i16", data=np.arange(-5, 5, dtype=np.int16).reshape(2, 5))
misc.create_dataset("f32_be", data=np.arange(6, dtype=">f4").reshape(3, 2))
arrays.update(misc_scalar=np.float64(3.25), misc_i16=np.arange(-5, 5, dtype=np.int16).reshape(2, 5),
              misc_f32_be=np.arange(6, dtype=np.float32).reshape(3, 2))
f.close()
np.savez_compressed(os.path.join(out, "lidc_like_expected.npz"), **arrays)
print("h5py", h5py.__version__, "hdf5", h5py.version.hdf5_version, os.path.getsize(os.path.join(out, "lidc_like.hdf5")), "bytes")
