"""`src.datamodules.ucf101_datamodule.UCF101DataModule` (reference: src/datamodules/ucf101_datamodule.py:6-27, base.py:4-53) over
the two sources this build has: pre-decoded clip folders (clip_folder_datamodule.py) and synthetic clips
(synthetic_datamodule.py).  Accepts the reference's config keys (configs/datamodule/ucf101.yaml) and yields the reference's batch
dict.  `source=None` (the reference's own file has no such key) means "folder": real data, as there."""
from src.datamodules.clip_folder_datamodule import ClipFolderDataModule
from src.datamodules.synthetic_datamodule import SyntheticClipDataModule


class UCF101DataModule:
    caption = "class"             # text = the class directory's name (ucf101_dataset.py:99)

    def __init__(self, data_folder=None, sequence_length=16, resolution=128, batch_size=32, num_workers=0, collate_fn=None,
                 source=None, n_batches=2, seed=0, device="cuda", dataname=None, tiny=False, progress_bar=True, devices=None,
                 **kwargs):
        self.sequence_length, self.resolution, self.batch_size = sequence_length, resolution, batch_size
        self.collate_fn, self.dataname = collate_fn, dataname
        if (source or "folder") == "synthetic":
            self.impl = SyntheticClipDataModule(sequence_length=sequence_length, resolution=resolution, batch_size=batch_size,
                                                n_batches=n_batches, seed=seed, device=device)
        elif (source or "folder") == "folder":
            self.impl = ClipFolderDataModule(data_folder, sequence_length=sequence_length, resolution=resolution,
                                             batch_size=batch_size, device=device, shuffle_seed=seed, **kwargs)
        else:
            raise ValueError(f"source must be 'folder' or 'synthetic', got {source!r}")

    def set_epoch(self, epoch):
        if hasattr(self.impl, "set_epoch"):
            self.impl.set_epoch(epoch)

    def setup(self, stage=None):
        pass

    def train_dataloader(self):
        return self.impl.train_dataloader()

    def val_dataloader(self):
        return self.impl.val_dataloader()

    def test_dataloader(self):
        return self.impl.test_dataloader()
