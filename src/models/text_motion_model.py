"""Stage-1 wrapper (VQ-VAE): the reference's TextMotionModel (src/models/text_motion_model.py:22-144): same constructor keys,
generator_step / sample_generator_step, ComputeLosses per split, automatic optimisation with
Adam(lr=gen_lr, betas=(0.5, 0.999)), epoch-end log keys `l/dummy/<split>`, `total/<split>`, `epoch`, `step`
(+ `recon/<split>`, `commitment/<split>`, which the reference does not log)."""
import torch

from gsdd_amd.hydra_lite import instantiate
from src.models.base import BaseModel
from src.models.metrics.loss import ComputeLosses


def _make_losses(cfg):
    """{split: ComputeLosses} from a `_target_` config (text_motion_model.py:66-70) or the default l_dummy."""
    def one():
        if isinstance(cfg, dict) and "_target_" in cfg:
            return instantiate(cfg, _recursive_=False)
        return ComputeLosses(loss_dict=dict((cfg or {}).get("loss_dict", {"l_dummy": 1.0})))
    return {split: one() for split in ("train", "test", "val")}


class TextMotionModel(BaseModel):
    def __init__(self, generator, losses=None, checkpoint_paths=None, evaluator=None, lr_args={}, render_animations=True,
                 do_evaluation=False, devices="cpu", **kwargs):
        super().__init__()
        self.gpu_device = devices if devices == "cpu" else "cuda:" + str(devices[0])
        self.generator = instantiate(generator, device=self.gpu_device, _recursive_=False) \
            if isinstance(generator, dict) else generator
        self.losses = _make_losses(losses)
        self.lr_args = dict(lr_args)
        self.render_animations = render_animations
        self.do_evaluation = do_evaluation
        self.evaluator = instantiate(evaluator, device=self.gpu_device, _recursive_=False) if do_evaluation else None
        self._extra = {s: {"recon": 0.0, "commitment": 0.0, "n": 0} for s in ("train", "test", "val")}

    def generator_step(self, batch):                                    # :76-82
        outputs = dict(self.generator(batch))
        outputs["length"] = batch["length"]
        return outputs

    def sample_generator_step(self, batch):                             # :85-91
        outputs = dict(self.generator(batch, do_inference=True))
        outputs["length"] = batch["length"]
        return outputs

    def allsplit_step(self, split, batch, batch_idx):                   # :93-107
        outputs = self.generator_step(batch)
        if self.do_evaluation and split != "train" and len(batch["length"]) != 1:
            eval_outputs = self.sample_generator_step(batch)
            self.evaluator.push_vals(batch, batch_idx, eval_outputs["pred_data"])
        loss = self.losses[split].update(outputs)
        ex = self._extra[split]
        ex["recon"] = ex["recon"] + outputs["losses"]["recon_loss"].detach()
        ex["commitment"] = ex["commitment"] + outputs["losses"]["commitment_loss"].detach()
        ex["n"] += 1
        return loss

    def allsplit_epoch_end(self, split, outputs):                       # :109-130
        dico = self._epoch_dico(self.losses[split], split)
        ex = self._extra[split]
        if ex["n"]:
            dico.update({f"recon/{split}": float(ex["recon"]) / ex["n"], f"commitment/{split}": float(ex["commitment"]) / ex["n"]})
        self._extra[split] = {"recon": 0.0, "commitment": 0.0, "n": 0}
        if self.do_evaluation and split != "train":
            metrics = self.evaluator.evaluate_metrics(self.trainer.datamodule, self.generator)
            dico.update({f"Metrics/{m}-{split}": v for m, v in metrics.items()})
            self.evaluator.reset()
        dico.update({"epoch": float(self.trainer.current_epoch), "step": float(self.trainer.current_epoch)})
        if split == "val" and self.current_epoch % 5 == 0:
            self.render_sample_results()
        self.log_dict(dico)

    def configure_optimizers(self):                                     # :132-144
        opt_g = torch.optim.Adam(self.generator.parameters(), lr=self.lr_args.get("gen_lr", 4e-4), betas=(0.5, 0.999))
        return [opt_g], []
