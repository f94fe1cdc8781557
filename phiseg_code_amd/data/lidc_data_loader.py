"""data/lidc_data_loader.py of the reference, reading side: `load_and_maybe_process_data` returns the pre-processed data set
<preprocessing_folder>/data_lidc.hdf5 (lidc_data_loader.py:107-135) as an h5py-like handle -- h5py when it is importable, else
the package's own reader (data/mini_hdf5.py).

The pre-processing step itself (lidc_data_loader.py:46-104: un-pickle the Probabilistic-U-Net LIDC crops, split the subjects
80 / 16 / 4 with sklearn's train_test_split, write the HDF5 file) is NOT restated: it needs an HDF5 writer and is a one-off of the
reference's own tooling; a file it produced is read as it is."""
import logging
import os


def open_hdf5(path):
    try:
        import h5py
        return h5py.File(path, 'r')
    except ImportError:
        from phiseg_code_amd.data import mini_hdf5
        return mini_hdf5.File(path, 'r')


def load_and_maybe_process_data(input_file, preprocessing_folder, force_overwrite=False):
    data_file_path = os.path.join(preprocessing_folder, 'data_lidc.hdf5')
    if os.path.exists(data_file_path) and not force_overwrite:
        logging.info('Already preprocessed this configuration. Loading now!')
        return open_hdf5(data_file_path)
    raise FileNotFoundError(
        "%s does not exist: run the reference's data/lidc_data_loader.py once on %s (pre-processing is not part of this package), or "
        "pass arrays / an .npz to phiseg_code_amd.data.lidc_data.lidc_data" % (data_file_path, input_file))
