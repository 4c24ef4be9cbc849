"""Stage-1 wrapper (VQ-VAE): same constructor keys / hooks as the reference's TextMotionModel
(src/models/text_motion_model.py:22-144): generator_step -> VQVAE.forward, losses `recon|commitment|total/<split>`,
Adam(lr=gen_lr, betas=(0.5, 0.999))."""
import torch

from gsdd_amd.hydra_lite import instantiate
from src.models.base import BaseModel


class TextMotionModel(BaseModel):
    def __init__(self, generator, losses=None, checkpoint_paths=None, evaluator=None, lr_args={}, render_animations=True,
                 do_evaluation=False, devices="cpu", **kwargs):
        super().__init__()
        self.gpu_device = devices if devices == "cpu" else "cuda:" + str(devices[0])
        self.generator = instantiate(generator, device=self.gpu_device, _recursive_=False) \
            if isinstance(generator, dict) else generator
        self.lr_args = dict(lr_args)
        self.do_evaluation = do_evaluation

    def generator_step(self, batch):
        return dict(self.generator(batch))

    def allsplit_step(self, split, batch, batch_idx):
        out = self.generator_step(batch)
        losses = out["losses"]
        total = torch.mean(losses["commitment_loss"] + losses["recon_loss"])          # compute_dummy, loss_func.py:10-14
        self.log_dict({f"recon/{split}": float(losses["recon_loss"].detach()), f"commitment/{split}": float(losses["commitment_loss"].detach()),
                       f"total/{split}": float(total.detach())})
        return total

    def configure_optimizers(self):
        return torch.optim.Adam(self.generator.parameters(), lr=self.lr_args.get("gen_lr", 4e-4), betas=(0.5, 0.999))
