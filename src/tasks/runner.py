"""Task runner used when pytorch_lightning / hydra are absent (reference: src/tasks/train_task.py:15-82, src/tasks/eval_task.py:14-62).
`Trainer` walks the LightningModule hooks in Lightning 1.6's order for the subset the reference uses:

  fit:  configure_optimizers -> per epoch [train(): training_step per batch (automatic optimisation: zero_grad/backward/step by
        the trainer; manual: the module does it, multistage_text_motion_model.py:186-200)] -> validation loop under no_grad in
        eval() -> validation_epoch_end -> training_epoch_end -> ModelCheckpoint (configs/callbacks/default.yaml:11-17: monitor
        `total/val`, mode min, save_top_k 1, filename "epoch_{epoch:03d}", save_last) ; `ckpt_path` resumes (train_task.py:64)
  test: test_step per batch -> test_epoch_end

Checkpoints are Lightning-layout dicts (`state_dict` with the module's attribute prefixes "generator." / "autoencoder.",
`optimizer_states`, `epoch`, `global_step`, `callbacks`) plus what `on_save_checkpoint` adds, written with torch.save and read
back with weights_only=True.  Multi-GPU: one process per GPU (torch.distributed.run); batches are dealt round-robin to ranks, the
gradient all-reduce lives in the trainers / the autograd bridges."""
import os
import time

import torch

from gsdd_amd.hydra_lite import instantiate
from gsdd_amd.parallel import init_distributed


class Trainer:
    def __init__(self, max_epochs=1, min_epochs=1, default_root_dir=None, callbacks=None, device="cuda", limit_batches=None, **kwargs):
        self.max_epochs, self.default_root_dir, self.device = int(max_epochs), default_root_dir, device
        ck = dict((callbacks or {}).get("model_checkpoint") or {})
        self.ckpt_cfg = {"dirpath": ck.get("dirpath") or (os.path.join(default_root_dir, "checkpoints") if default_root_dir else None),
                         "filename": ck.get("filename") or "epoch_{epoch:03d}", "monitor": ck.get("monitor"),
                         "mode": ck.get("mode", "min"), "save_last": bool(ck.get("save_last")),
                         "save_top_k": ck.get("save_top_k", 1)}
        self.limit_batches = limit_batches
        self.current_epoch, self.global_step = 0, 0
        self.callback_metrics, self.optimizers = {}, []
        self.best_model_path, self.best_score = "", None
        self.log_dir = default_root_dir
        self.datamodule = None
        self._dm_sharded = False
        self.rank, self.world = init_distributed()

    # ------------------------------------------------------------------ plumbing
    @staticmethod
    def _normalise_optimizers(ret):
        if isinstance(ret, dict):
            return [ret["optimizer"]]
        if isinstance(ret, (list, tuple)):
            if len(ret) == 2 and isinstance(ret[0], (list, tuple)):          # ([optimizers], [schedulers])
                return list(ret[0])
            return list(ret)
        return [ret]

    def _batches(self, loader):
        """Batches are dealt round-robin to the ranks, in whole rounds: a trailing round that cannot serve every rank is dropped,
        so all ranks take the same number of steps (every step holds collectives).  A datamodule that shards by itself
        (`set_shard(rank, world)`: this rank only reads and preprocesses its own batches, like a DistributedSampler) hands over
        exactly those batches and marks its loaders (`gsdd_sharded`); any other iterable -- also one handed over next to such a
        datamodule -- is dealt here, every rank walking all of it."""
        if self._dm_sharded and getattr(loader, "gsdd_sharded", False):
            for i, batch in enumerate(loader):
                if self.limit_batches is not None and i >= self.limit_batches:
                    break
                yield i, batch
            return
        mine = None
        for i, batch in enumerate(loader):
            if self.limit_batches is not None and i >= self.limit_batches * self.world:
                break
            if i % self.world == self.rank:
                mine = batch
            if i % self.world == self.world - 1:                              # the round is complete
                yield i // self.world, mine
                mine = None

    def _attach(self, model, datamodule):
        model.trainer = self
        self.datamodule = datamodule
        self._dm_sharded = False
        if hasattr(datamodule, "set_shard"):
            datamodule.set_shard(self.rank, self.world)
            self._dm_sharded = True
        if self.world > 1:
            from gsdd_amd.parallel import broadcast_module
            broadcast_module(model)                                           # (per-rank noise rows: parallel.set_rank_noise_rows, per step)

    # ------------------------------------------------------------------ checkpoints
    def _checkpoint(self, model):
        ck = {"epoch": self.current_epoch, "global_step": self.global_step, "pytorch-lightning_version": "1.6.5",
              "state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
              "optimizer_states": [o.state_dict() for o in self.optimizers],
              "callbacks": {"ModelCheckpoint": {"best_model_score": self.best_score, "best_model_path": self.best_model_path}}}
        model.on_save_checkpoint(ck)
        return ck

    def save_checkpoint(self, model, path):
        if self.rank != 0:
            return
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save(self._checkpoint(model), path)

    def _load_checkpoint(self, model, path, optimizers=True):
        ck = torch.load(path, map_location="cpu", weights_only=True)
        model.load_state_dict(ck["state_dict"])
        for m in model.modules():
            if hasattr(m, "_packed"):
                m._packed = None
        if optimizers:
            for o, st in zip(self.optimizers, ck.get("optimizer_states", [])):
                o.load_state_dict(st)
            self.current_epoch = int(ck["epoch"]) + 1
            self.global_step = int(ck["global_step"])
            cb = ck.get("callbacks", {}).get("ModelCheckpoint", {})
            self.best_score, self.best_model_path = cb.get("best_model_score"), cb.get("best_model_path", "")
        model.on_load_checkpoint(ck)
        return ck

    def _on_epoch_checkpoint(self, model):
        cfg = self.ckpt_cfg
        if not cfg["dirpath"]:
            return
        if cfg["monitor"] and cfg["monitor"] in self.callback_metrics and cfg["save_top_k"]:
            score = float(self.callback_metrics[cfg["monitor"]])
            better = self.best_score is None or (score < self.best_score if cfg["mode"] == "min" else score > self.best_score)
            if better:
                old = self.best_model_path
                self.best_score = score
                self.best_model_path = os.path.join(cfg["dirpath"], cfg["filename"].format(epoch=self.current_epoch) + ".ckpt")
                self.save_checkpoint(model, self.best_model_path)
                if old and old != self.best_model_path and self.rank == 0 and os.path.exists(old):
                    os.remove(old)                                            # save_top_k = 1
        if cfg["save_last"]:
            self.save_checkpoint(model, os.path.join(cfg["dirpath"], "last.ckpt"))

    # ------------------------------------------------------------------ loops
    def _eval_loop(self, model, loader, step, epoch_end):
        was_training = model.training
        model.eval()
        outs = []
        with torch.no_grad():
            for i, batch in self._batches(loader):
                outs.append(step(batch, i))
            epoch_end(outs)
        model.train(was_training)
        return outs

    def fit(self, model, datamodule, ckpt_path=None):
        self._attach(model, datamodule)
        self.optimizers = self._normalise_optimizers(model.configure_optimizers())
        if ckpt_path:
            self._load_checkpoint(model, ckpt_path)
        while self.current_epoch < self.max_epochs:
            model.train()
            if hasattr(datamodule, "set_epoch"):
                datamodule.set_epoch(self.current_epoch)
            outs = []
            for i, batch in self._batches(datamodule.train_dataloader()):
                out = model.training_step(batch, i)
                if getattr(model, "automatic_optimization", True):
                    for o in self.optimizers:
                        o.zero_grad()
                    out.backward()
                    for o in self.optimizers:
                        o.step()
                self.global_step += 1
                outs.append(out.detach() if torch.is_tensor(out) else {k: v.detach() for k, v in out.items()})
            self._eval_loop(model, datamodule.val_dataloader(), model.validation_step, model.validation_epoch_end)
            model.training_epoch_end(outs)
            if self.world > 1:                                                # measured cost of the gradient exchange (last step)
                for m in model.modules():
                    red = getattr(getattr(m, "_hip_trainer", None), "reducer", None)
                    if red is not None:
                        self.callback_metrics.update({f"comm/{k}": float(v) for k, v in red.stats().items()})
                native = getattr(getattr(model, "_native", None), "reducer", None)
                if native is not None:
                    self.callback_metrics.update({f"comm/{k}": float(v) for k, v in native.stats().items()})
            self._on_epoch_checkpoint(model)
            if self.rank == 0:
                print(f"epoch {self.current_epoch}: " + ", ".join(f"{k} {v:.5g}" for k, v in sorted(self.callback_metrics.items())
                                                                   if isinstance(v, float)))
            self.current_epoch += 1
        return self.callback_metrics

    def test(self, model, datamodule, ckpt_path=None):
        self._attach(model, datamodule)
        if ckpt_path:
            self._load_checkpoint(model, ckpt_path, optimizers=False)
        return self._eval_loop(model, datamodule.test_dataloader(), model.test_step, model.test_epoch_end)


def build(cfg):
    init_distributed()                                  # binds cuda:LOCAL_RANK before anything touches the GPU
    if cfg.get("seed") is not None:
        torch.manual_seed(cfg.seed)
    datamodule = instantiate(cfg.datamodule)
    model = instantiate(cfg.model, _recursive_=False)
    tcfg = {k: v for k, v in dict(cfg.trainer).items() if k != "_target_"}
    dev = tcfg.pop("device", "cuda")
    if dev == "cuda" and torch.cuda.is_available():
        dev = f"cuda:{torch.cuda.current_device()}"
    trainer = Trainer(callbacks=cfg.get("callbacks"), device=dev, **tcfg)
    return datamodule, model.to(dev), trainer


def evaluate(cfg):
    datamodule, model, trainer = build(cfg)
    t0 = time.perf_counter()
    outs = trainer.test(model, datamodule, ckpt_path=cfg.get("ckpt_path"))
    n = sum(o["pred_data"].shape[0] for o in outs if hasattr(o, "get") and torch.is_tensor(o.get("pred_data")))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"test: {len(outs)} batches ({n} sampled clips) in {dt:.2f} s", trainer.callback_metrics)
    return dict(trainer.callback_metrics)


def train(cfg):
    datamodule, model, trainer = build(cfg)
    metrics = {}
    if cfg.get("train", True):
        metrics.update(trainer.fit(model, datamodule, ckpt_path=cfg.get("ckpt_path")))
    if cfg.get("test"):
        trainer.test(model, datamodule, ckpt_path=trainer.best_model_path or None)
        metrics.update(trainer.callback_metrics)
    return metrics, {"cfg": cfg, "datamodule": datamodule, "model": model, "trainer": trainer}
