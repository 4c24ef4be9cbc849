"""Loss accumulator with the reference's interface (src/models/metrics/loss.py:6-62, a torchmetrics.Metric there; torchmetrics
is not in this image, and the only features the reference uses are sum-states, update/compute/reset and loss2logname).

`loss_dict` maps a loss name to its weight; "total" (weight 1) is added.  `update(output)` returns the weighted sum WITH its
autograd graph (that is what manual_backward receives, multistage_text_motion_model.py:193-197) and accumulates detached values;
`compute()` -> per-loss means over the updates since `reset()`.  With torch.distributed initialised the sums are all-reduced in
compute() (dist_reduce_fx="sum", :15-16)."""
import torch
import torch.distributed as dist


def compute_dummy(output, loss_opts=None):
    """loss_func.py:10-14: stage 1 hands a dict of two losses, stage 2 a scalar tensor."""
    losses = output["losses"]
    if isinstance(losses, dict):
        return torch.mean(losses["commitment_loss"] + losses["recon_loss"])
    return torch.mean(losses)


_matching_ = {"l_dummy": compute_dummy}


def get_loss_function(ltype):
    if ltype not in _matching_:
        raise KeyError(f"loss '{ltype}': the path uses l_dummy only (configs/model/*.yaml); the pose losses l_codebook / "
                       "l_entropy / l_perplexity of loss_func.py:17-27 read keys no model on this path produces")
    return _matching_[ltype]


class ComputeLosses:
    def __init__(self, loss_dict=None, loss_opts=None, dist_sync_on_step=False, **kwargs):
        losses = dict(loss_dict or {})
        losses["total"] = 1.0
        self.losses, self.loss_opts, self._params = losses, dict(loss_opts or {}), losses
        self._losses_func = {name: get_loss_function(name) for name in losses if name != "total"}
        self.reset()

    def reset(self):
        self._sums = {name: 0.0 for name in self.losses}         # python floats or 0-dim device tensors (no sync per update)
        self.count = 0

    def update(self, output):
        total = 0.0
        for name in self.losses:
            if name == "total":
                continue
            val = self._losses_func[name](output, self.loss_opts)
            self._sums[name] = self._sums[name] + val.detach()
            total = total + self._params[name] * val
        self._sums["total"] = self._sums["total"] + total.detach()
        self.count += 1
        return total

    def compute(self):
        sums = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in self._sums.items()}
        count = torch.tensor(float(self.count))
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dev = next((v.device for v in sums.values() if v.is_cuda), torch.device("cpu"))
            if dist.get_backend() == "nccl" and dev.type != "cuda":
                dev = torch.device("cuda", torch.cuda.current_device())
            flat = torch.stack([sums[k].to(dev) for k in self.losses] + [count.to(dev)])
            dist.all_reduce(flat)
            sums, count = {k: flat[i] for i, k in enumerate(self.losses)}, flat[-1]
        return {k: sums[k] / count for k in self.losses}

    @staticmethod
    def loss2logname(loss, split):
        if loss == "total":
            return f"{loss}/{split}"
        loss_type, name = loss.split("_")
        return f"{loss_type}/{name}/{split}"
