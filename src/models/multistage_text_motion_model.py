"""Stage-2 wrapper (D3PM over VQ-VAE tokens): same constructor keys / hooks as the reference's
MultistageTextMotionModel (src/models/multistage_text_motion_model.py:31-252).  The autoencoder runs in eval()
(the reference leaves it in train mode by accident)."""
import torch

from gsdd_amd.hydra_lite import instantiate
from src.models.base import BaseModel


class MultistageTextMotionModel(BaseModel):
    def __init__(self, generator, autoencoder, generator_losses=None, checkpoint_paths=None, evaluator=None,
                 freeze_models_dict=None, lr_args={}, render_animations=True, do_evaluation=False, devices="cpu",
                 **kwargs):
        super().__init__()
        self.gpu_device = devices if devices == "cpu" else "cuda:" + str(devices[0])
        self.generator = instantiate(generator) if isinstance(generator, dict) else generator
        self.autoencoder = instantiate(autoencoder, device=self.gpu_device, _recursive_=False) \
            if isinstance(autoencoder, dict) else autoencoder
        ckpt = (checkpoint_paths or {}).get("autoencoder") if isinstance(checkpoint_paths, dict) else checkpoint_paths
        if ckpt and ckpt != "__None__":                 # load_checkpoints (multistage_text_motion_model.py:113-122)
            from gsdd_amd.checkpoint import load_reference_checkpoint
            load_reference_checkpoint(self.autoencoder, ckpt)
        gen_ckpt = (checkpoint_paths or {}).get("generator") if isinstance(checkpoint_paths, dict) else None
        if gen_ckpt and gen_ckpt != "__None__":
            from gsdd_amd.checkpoint import load_reference_checkpoint
            load_reference_checkpoint(self.generator, gen_ckpt)
        self.autoencoder.eval()
        self.lr_args = dict(lr_args)
        self.do_evaluation = do_evaluation
        self.automatic_optimization = False

    def generator_step(self, batch):
        return dict(self.generator(batch, self.autoencoder, None))

    @torch.no_grad()
    def sample_generator_step(self, batch):
        clips = self.generator.sample_videos(batch["text"], self.autoencoder)
        return {"pred_data": clips, "gt_data": batch.get("video")}

    def allsplit_step(self, split, batch, batch_idx):
        """train: one optimiser step of the D3PM generator on the HIP path (zero_grad -> backward -> Adam in the reference,
        multistage_text_motion_model.py:186-197; here gsdd_amd.d3pm_train.D3PMTrainer.step, which also averages
        gradients over the data-parallel group).  The autoencoder stays frozen in eval mode (auto_lr 1e-6 in the
        reference is effectively a no-op and its backward is not built)."""
        if split == "train":
            from gsdd_amd.d3pm_train import D3PMTrainer
            if getattr(self, "_trainer", None) is None:
                self._trainer = D3PMTrainer(self.generator.diffusion_model, lr=self.lr_args.get("gen_lr", 1e-4))
            with torch.no_grad():
                x = batch["video"].to(self.autoencoder.device)
                tokens = self.autoencoder.encode(x).view(x.shape[0], -1)
                text_emb = torch.zeros_like(self.generator.textencoder(batch["text"]).unsqueeze(1).to(tokens.device))
            loss = self._trainer.step(tokens, text_emb)[0]
            self.log_dict({f"total/{split}": float(loss)})
            return loss
        return self.sample_generator_step(batch)

    def configure_optimizers(self):
        return [torch.optim.Adam(self.generator.parameters(), lr=self.lr_args.get("gen_lr", 1e-4), betas=(0.5, 0.999)),
                torch.optim.Adam(self.autoencoder.parameters(), lr=self.lr_args.get("auto_lr", 1e-6))]
