"""data/batch_provider.py of the reference: `BatchProvider(X, y, indices, add_dummy_dimension=..., do_augmentations=...,
augmentation_options=..., num_labels_per_subject=..., annotator_range=...)` with `next_batch(batch_size)` -- here the
device-resident provider (data/augment.py: the data set lives in HBM, one augmentation launch per batch, random decisions from
the Philox contract instead of the unseeded global numpy RNG)."""
from phiseg_code_amd.data.augment import DeviceBatchProvider


class BatchProvider(DeviceBatchProvider):

    def __init__(self, X, y, indices, add_dummy_dimension=True, do_augmentations=False, augmentation_options=None,
                 num_labels_per_subject=1, annotator_range=None, **kwargs):
        if not add_dummy_dimension:
            raise NotImplementedError("add_dummy_dimension=False: every call site of the reference passes True (lidc_data.py:36-50)")
        super().__init__(X, y, indices=indices, do_augmentations=do_augmentations, augmentation_options=augmentation_options,
                         num_labels_per_subject=num_labels_per_subject, annotator_range=annotator_range, **kwargs)
