"""Fallback task runner used when pytorch_lightning / hydra are absent (reference: src/tasks/train_task.py:15-82,
src/tasks/eval_task.py:14-62): instantiate datamodule + model from the composed config and drive the hooks."""
import time

import torch

from gsdd_amd.hydra_lite import instantiate


def build(cfg):
    if cfg.get("seed") is not None:
        torch.manual_seed(cfg.seed)
    datamodule = instantiate(cfg.datamodule)
    model = instantiate(cfg.model, _recursive_=False)
    dev = cfg.trainer.get("device", "cuda")
    return datamodule, model.to(dev)


def evaluate(cfg):
    datamodule, model = build(cfg)
    model.eval()
    n, t0 = 0, time.perf_counter()
    for i, batch in enumerate(datamodule.test_dataloader()):
        out = model.test_step(batch, i)
        if isinstance(out, dict) and out.get("pred_data") is not None:
            n += out["pred_data"].shape[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"test: {n} clips in {dt:.2f} s ({n / dt:.3f} clips/s)", getattr(model, "_logged", ""))
    return {"clips": n, "seconds": dt}


def train(cfg):
    datamodule, model = build(cfg)
    model.train()
    opts = model.configure_optimizers()
    opts = opts if isinstance(opts, (list, tuple)) else [opts]
    for epoch in range(cfg.trainer.get("max_epochs", 1)):
        for i, batch in enumerate(datamodule.train_dataloader()):
            loss = model.training_step(batch, i)        # stage 2: a full optimiser step on the HIP path
            if loss.requires_grad:                      # stage 1: VQVAE.forward hands back losses whose grad_fn is the HIP backward
                for o in opts:
                    o.zero_grad()
                loss.backward()
                for o in opts:
                    o.step()
            print(f"epoch {epoch} step {i} loss {float(loss.detach()):.5f}")
    return {}
