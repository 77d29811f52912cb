from segan_pytorch_amd.datasets import (SEDataset, SyntheticSEDataset, collate_fn, de_emphasize,  # noqa
                                        normalize_wave_minmax, pre_emphasize, slice_signal_index)
