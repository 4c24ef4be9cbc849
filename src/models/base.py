"""Hook plumbing of the reference's BaseModel (src/models/base.py:4-63).  With pytorch_lightning installed this IS a
LightningModule; without it (this image) a small stand-in offers the attributes the hooks use — `trainer`, `current_epoch`,
`optimizers()`, `manual_backward`, `log_dict`, `save_hyperparameters`, `on_save/on_load_checkpoint` — and
src/tasks/runner.py::Trainer drives them in Lightning 1.6's order."""
import torch
import torch.nn as nn

try:                                            # use Lightning when the environment has it
    from pytorch_lightning import LightningModule as _Base
except Exception:                               # pragma: no cover - not installed in this image
    class _Base(nn.Module):
        automatic_optimization = True
        trainer = None

        def save_hyperparameters(self, *a, **k):
            pass

        def log_dict(self, d, **k):
            self._logged = dict(d)
            if self.trainer is not None:
                self.trainer.callback_metrics.update(d)

        def log(self, name, value, **k):
            self.log_dict({name: value})

        @property
        def device(self):
            return next(self.parameters()).device

        @property
        def current_epoch(self):
            return self.trainer.current_epoch if self.trainer is not None else 0

        def optimizers(self):
            opts = self.trainer.optimizers
            return opts[0] if len(opts) == 1 else opts

        def manual_backward(self, loss, *a, **k):
            loss.backward(*a, **k)

        def on_save_checkpoint(self, checkpoint):
            pass

        def on_load_checkpoint(self, checkpoint):
            pass


class BaseModel(_Base):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.save_hyperparameters(logger=False)

    def training_step(self, batch, batch_idx):
        return self.allsplit_step("train", batch, batch_idx)

    def validation_step(self, batch, batch_idx):
        return self.allsplit_step("val", batch, batch_idx)

    def test_step(self, batch, batch_idx):
        return self.allsplit_step("test", batch, batch_idx)

    def _epoch_dico(self, losses, split):
        """base.py:37-40: the accumulated means under their log names; the accumulator starts over for the next epoch
        (torchmetrics resets a Metric the trainer logs at epoch end; the reference relies on that)."""
        dico = {losses.loss2logname(name, split): float(value) for name, value in losses.compute().items()}
        losses.reset()
        return dico

    def allsplit_epoch_end(self, split, outputs):                         # base.py:36-54
        dico = self._epoch_dico(self.losses[split], split)
        dico.update({"epoch": float(self.trainer.current_epoch), "step": float(self.trainer.current_epoch)})
        if split == "val" and self.current_epoch % 10 == 0:
            self.render_sample_results()
        self.log_dict(dico)

    def training_epoch_end(self, outputs):
        return self.allsplit_epoch_end("train", outputs)

    def validation_epoch_end(self, outputs):
        return self.allsplit_epoch_end("val", outputs)

    def test_epoch_end(self, outputs):
        return self.allsplit_epoch_end("test", outputs)

    def render_sample_results(self):
        """The reference writes .mp4 files through its OpenGL renderer (multistage_text_motion_model.py:254-281); rendering is
        out of scope.  The sampled clips of one validation item are kept on `self.last_sample`."""
        if not getattr(self, "render_animations", False) or self.trainer is None:
            return
        loader = self.trainer.datamodule.val_dataloader()
        batch = next(iter(loader), None)
        if batch is None:
            return
        one = {k: (v[:1] if torch.is_tensor(v) or isinstance(v, list) else v) for k, v in batch.items()}
        was_training = self.generator.training
        self.generator.eval()
        with torch.no_grad():
            self.last_sample = self.sample_generator_step(one)
        self.generator.train(was_training)

    # ---- state that is not in state_dict but decides the next training step (exact resume)
    def _diffusion_models(self):
        from gsdd_amd.d3pm import DiffusionTransformer
        return [m for m in self.modules() if isinstance(m, DiffusionTransformer)]

    def on_save_checkpoint(self, checkpoint):
        dms = self._diffusion_models()
        checkpoint["gsdd"] = {"noise": [(m.noise_seed, m.noise_stream, m.row_offset) for m in dms],
                              "rng_cpu": torch.get_rng_state(),
                              "rng_cuda": torch.cuda.get_rng_state() if torch.cuda.is_available() else None,
                              "need_init": [bool(getattr(m, "_need_init")) for m in self.modules() if hasattr(m, "_need_init")]}

    def on_load_checkpoint(self, checkpoint):
        st = checkpoint.get("gsdd")
        if not st:
            return
        for m, (seed, stream, row) in zip(self._diffusion_models(), st["noise"]):
            m.set_noise(seed, stream, row)
        torch.set_rng_state(st["rng_cpu"].cpu())
        if st.get("rng_cuda") is not None and torch.cuda.is_available():
            torch.cuda.set_rng_state(st["rng_cuda"].cpu())
        for m, flag in zip([m for m in self.modules() if hasattr(m, "_need_init")], st["need_init"]):
            m._need_init = flag
