"""Folder-of-clips datamodule: a working counterpart of the reference's UCF101 dataset (src/datamodules/datasets/
ucf101_dataset.py:20-99, which does not parse — SyntaxError at :88 — and needs torchvision's video reader and a pretrained
ResNet-50, neither available offline).  Same folder layout and batch contract:

    <data_folder>/<split>/<ClassName>/<clip>.npy        uint8 frames (T, H, W, 3), decoded ahead of time

* classes = sorted parent-directory names, label = index (:57-60), text = class name (:99);
* clips = windows of `sequence_length` consecutive frames, a new window every `frames_between_clips` frames (the reference's
  VideoClips(files, sequence_length, 100), :66);
* video = preprocess(frames, resolution) (:105-140) — here on the GPU (gsdd_amd.data.preprocess, uint8 in, fp32 CTHW out);
* batch dict keys video / text / length / label / frame / orig_length (src/datamodules/data_utils.py:16-36).  `length` is
  len(video) of the CTHW tensor, i.e. 3, exactly as the reference computes it (:99); `frame` (ResNet-50 features of the
  first frame in the reference, unused by the path) is a zero vector."""
import glob
import os

import numpy as np
import torch


class ShardedLoader:
    """The batches of ONE rank, in order (an iterator).  `gsdd_sharded` tells the runner that this loader has already dealt the
    batches to the ranks (src/tasks/runner.py::Trainer._batches deals any other iterable itself)."""
    gsdd_sharded = True

    def __init__(self, it, shard):
        self._it, self.shard = it, shard

    def __iter__(self):
        return self._it


class ClipFolderDataset:
    def __init__(self, data_folder, sequence_length, split="train", resolution=64, frames_between_clips=100, class_names=None, **kw):
        self.sequence_length, self.resolution = sequence_length, resolution
        folder = os.path.join(data_folder, split)
        files = sorted(glob.glob(os.path.join(folder, "**", "*.npy"), recursive=True))
        parent = lambda f: os.path.basename(os.path.dirname(f))
        if class_names is not None:
            files = [f for f in files if parent(f) in class_names]
        self.classes = sorted(set(parent(f) for f in files))
        self.class_to_label = {c: i for i, c in enumerate(self.classes)}
        self.clips = []                                   # (file, first frame)
        for f in files:
            n = np.load(f, mmap_mode="r").shape[0]
            for start in range(0, n - sequence_length + 1, frames_between_clips):
                self.clips.append((f, start))

    @property
    def n_classes(self):
        return len(self.classes)

    def __len__(self):
        return len(self.clips)

    def __getitem__(self, idx):
        f, start = self.clips[idx]
        frames = np.load(f, mmap_mode="r")[start:start + self.sequence_length]
        name = os.path.basename(os.path.dirname(f))
        return dict(frames=torch.from_numpy(np.array(frames)), label=self.class_to_label[name], text=name,
                    orig_length=int(frames.shape[0]))


class ClipFolderDataModule:
    def __init__(self, data_folder, sequence_length=16, resolution=128, batch_size=16, device="cuda", shuffle_seed=0, **kwargs):
        self.args = dict(data_folder=data_folder, sequence_length=sequence_length, resolution=resolution, **kwargs)
        self.batch_size, self.device, self.shuffle_seed = batch_size, device, shuffle_seed
        self.sequence_length, self.resolution = sequence_length, resolution
        self.epoch = 0
        self.shard = (0, 1)

    def set_shard(self, rank, world):
        """Data parallel: this process reads, decodes and preprocesses only batches rank, rank + world, ... of every split (whole
        rounds: a trailing round that cannot serve every rank is dropped, all ranks take the same number of steps) -- the role of
        the reference's DistributedSampler.  The order is the single-process order, so N ranks see the same global batches."""
        self.shard = (int(rank), int(world))

    def set_epoch(self, epoch):
        self.epoch = int(epoch)                          # the shuffle is a function of (shuffle_seed, epoch)

    def _loader(self, split, shuffle):
        return ShardedLoader(self._batches(split, shuffle), self.shard)

    def _batches(self, split, shuffle):
        from gsdd_amd.data import preprocess
        ds = ClipFolderDataset(split=split, **self.args)
        order = list(range(len(ds)))
        if shuffle:
            order = torch.randperm(len(ds), generator=torch.Generator().manual_seed(self.shuffle_seed + self.epoch)).tolist()
        rank, world = self.shard
        # data parallel: full batches only -- a short trailing batch would give one rank a different local batch size, i.e. overlapping
        # noise rows (row_offset = rank * local batch) and an unweighted gradient mean; one process keeps it, as the reference's loader does
        nbatch = (len(order) + self.batch_size - 1) // self.batch_size if world == 1 else len(order) // self.batch_size
        for b in range(rank, (nbatch // world) * world, world):
            i = b * self.batch_size
            items = [ds[j] for j in order[i:i + self.batch_size]]
            video = torch.stack([preprocess(it["frames"].to(self.device), self.resolution) for it in items])   # (B,3,T,R,R)
            yield {"video": video, "text": [it["text"] for it in items], "length": [video.shape[1]] * len(items),
                   "label": torch.tensor([it["label"] for it in items]), "frame": torch.zeros(len(items), 1000),
                   "orig_length": [it["orig_length"] for it in items]}

    def train_dataloader(self):
        return self._loader("train", True)

    def val_dataloader(self):
        return self._loader("test", False)

    def test_dataloader(self):
        return self._loader("test", False)
