"""FVD evaluator with the reference's interface (src/utils/evaluator.py:10-116): push_vals(batch, batch_idx, outputs) ->
evaluate_metrics(...) -> {'fvd': ...} -> reset().  The statistic is gsdd_amd.metrics.frechet_distance (pinned to values of the
reference's function, tests/golden/frechet.npz).  The feature extractor is the reference's Kinetics-400 I3D
(src/models/motionencoder/pytorch_i3d.py -> gsdd_amd/i3d.py: the same architecture and state_dict keys on the HIP conv path,
pinned to the reference module on seeded weights) fed with clips de-normalised, resized to 224 and scaled to [-1, 1]; its
pretrained weights cannot be obtained offline, so `checkpoint_paths` must point at a supplied state_dict (or `videoencoder` at
any other module mapping (B,3,T,H,W) clips to features).  Unlike the reference (which never leaves train mode here:
evaluator.py:14-29), the extractor runs in eval mode."""
import torch

from gsdd_amd.hydra_lite import instantiate
from gsdd_amd.metrics import frechet_distance

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


class MeanPoolEncoder(torch.nn.Module):
    """A weight-free stand-in feature extractor (NOT I3D: the numbers it yields are no FVD): per-channel means over a `grid`^3 partition
    of every clip, (B, 3, T, H, W) -> (B, 3 * grid^3).  Lets the evaluation plumbing (`model.do_evaluation=true`,
    `model.evaluator.videoencoder._target_=src.utils.evaluator.MeanPoolEncoder`) run where no I3D weights exist."""

    def __init__(self, grid=2, **kwargs):
        super().__init__()
        self.grid = grid
        self.register_buffer("_anchor", torch.zeros(1))

    def forward(self, clips):
        return torch.nn.functional.adaptive_avg_pool3d(clips.float(), self.grid).flatten(1)


class Evaluator:
    def __init__(self, device, videoencoder, checkpoint_paths=None, target_resolution=224):
        self.device, self.target_resolution = device, target_resolution
        self.videoencoder = instantiate(videoencoder, _recursive_=False) if isinstance(videoencoder, dict) else videoencoder
        if checkpoint_paths and checkpoint_paths != "__None__":
            self.videoencoder.load_state_dict(torch.load(checkpoint_paths, map_location="cpu", weights_only=True))
        elif any(True for _ in self.videoencoder.parameters()):
            import warnings
            warnings.warn("Evaluator: no checkpoint for the video encoder (model.evaluator.checkpoint_paths / eval_ckpt): it runs on its "
                          "random initialisation, and the distance it yields is not an FVD", UserWarning)
        self.videoencoder.to(device).eval()
        self.reset()

    def reset(self):
        self.all_video_embeds_generated, self.all_video_embeds_gt = [], []

    def _prepare(self, clips):
        """evaluator.py:42-70: undo the ImageNet normalisation, quantise to uint8, preprocess at 224 (x2: [-1, 1]-ish range),
        repeat frames of 4- / 8-frame clips to 16."""
        from gsdd_amd.data import preprocess
        x = clips.detach().to(self.device).float().permute(0, 2, 3, 4, 1)                 # (B,T,H,W,3)
        x = x * torch.tensor(STD, device=x.device) + torch.tensor(MEAN, device=x.device)
        x = (x * 255).to(torch.uint8)                                                     # astype('uint8') truncation, as there
        out = torch.stack([preprocess(v.contiguous(), self.target_resolution) for v in x]) * 2
        if out.shape[2] in (4, 8):
            out = torch.repeat_interleave(out, 16 // out.shape[2], dim=2)
        return out

    @torch.no_grad()
    def push_vals(self, batch, batch_idx, outputs):
        self.all_video_embeds_generated.append(self.videoencoder(self._prepare(outputs)).float().cpu())
        self.all_video_embeds_gt.append(self.videoencoder(self._prepare(batch["video"])).float().cpu())

    def evaluate_metrics(self, val_dataset=None, generator=None):
        gen = torch.cat(self.all_video_embeds_generated).flatten(1)
        gt = torch.cat(self.all_video_embeds_gt).flatten(1)
        return {"fvd": float(frechet_distance(gen, gt))}
