"""Synthetic clip datamodule with the reference's batch-dict contract (src/datamodules/data_utils.py:16-36):
keys video (B,3,T,H,W) / text / length / label / frame / orig_length.  UCF101 clips are ImageNet-normalised
(ucf101_dataset.py:107-111), i.e. roughly zero-mean unit-scale: randn stands in for them."""
import torch


class SyntheticClipDataModule:
    def __init__(self, sequence_length=16, resolution=128, batch_size=16, n_batches=2, seed=0, device="cuda",
                 **kwargs):
        self.sequence_length, self.resolution, self.batch_size = sequence_length, resolution, batch_size
        self.n_batches, self.seed, self.device = n_batches, seed, device
        self.epoch = 0

    def set_epoch(self, epoch):
        """Fresh training clips every epoch (a function of (seed, epoch, batch index) only: resuming at epoch e replays exactly
        what an uninterrupted run would have seen)."""
        self.epoch = int(epoch)

    def _loader(self, offset):
        for i in range(self.n_batches):
            g = torch.Generator().manual_seed(self.seed + offset + i + (1_000_003 * self.epoch if offset == 0 else 0))
            B, T, R = self.batch_size, self.sequence_length, self.resolution
            yield {"video": torch.randn(B, 3, T, R, R, generator=g).to(self.device),
                   "text": [f"synthetic action {j % 101}" for j in range(B)], "length": [T] * B,
                   "label": torch.arange(B) % 101, "frame": torch.zeros(B, dtype=torch.long), "orig_length": [T] * B}

    def train_dataloader(self):
        return self._loader(0)

    def val_dataloader(self):
        return self._loader(10_000)

    def test_dataloader(self):
        return self._loader(20_000)
