"""data/lidc_data.py of the reference: `lidc_data(exp_config)` with .train (augmented, random annotator), .validation and .test
providers (+ .validation.images / .labels, .test.images / .labels as the evaluation scripts read them), built over
exp_config.preproc_folder/data_lidc.hdf5.  The providers keep the data set in HBM and augment on the device
(data/augment.py); `source` overrides the file: a dict of arrays, an .npz or an HDF5 path."""
import os

from phiseg_code_amd.data import augment


class lidc_data(augment.lidc_data):

    def __init__(self, exp_config, source=None, seed=1234):
        if source is None:
            root = getattr(exp_config, 'data_root', '')
            if str(root).endswith(('.hdf5', '.h5', '.npz')):
                source = root
            else:
                source = os.path.join(exp_config.preproc_folder, 'data_lidc.hdf5')
                if not os.path.exists(source):
                    from phiseg_code_amd.data import lidc_data_loader
                    lidc_data_loader.load_and_maybe_process_data(root, exp_config.preproc_folder)    # raises with instructions
        if not hasattr(exp_config, 'annotator_range'):          # (lidc_data.py:31-33)
            exp_config.annotator_range = range(exp_config.num_labels_per_subject)
        super().__init__(exp_config, source, seed=seed)
