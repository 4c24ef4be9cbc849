"""The reference's command lines on the HIP path, at toy sizes: `python src/train.py model=videogpt_vq_vae`,
`python src/train.py model=discrete_diffusion`, `python src/eval.py` (reference: src/train.py:17-34, src/eval.py,
src/tasks/train_task.py:15-82, src/tasks/eval_task.py:14-62) through the Hydra-style composition of configs/: training +
validation loops, epoch-end log keys, ModelCheckpoint files (configs/callbacks/default.yaml:11-17) and `ckpt_path` resume
(train_task.py:64)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

SMALL_DATA = ["datamodule.resolution=32", "datamodule.sequence_length=4", "batch_size=2", "datamodule.n_batches=3"]
SMALL_VQ = ["{p}.n_hiddens=16", "{p}.n_codes=32", "{p}.embedding_dim=8", "{p}.n_res_layers=1"]
SMALL_D3PM = ["model.generator.diffusion_model.content_seq_len=64", "model.generator.diffusion_model.diffusion_step=20",
              "model.generator.diffusion_model.transformer.content_seq_len=64",
              "model.generator.diffusion_model.transformer.n_layer=2",
              "model.generator.diffusion_model.transformer.content_spatial_size=[8,8]",
              "model.generator.diffusion_model.transformer.dalle.num_embed=32",
              "model.generator.diffusion_model.transformer.dalle.spatial_size=[8,8]"]
STAGE1 = SMALL_DATA + [s.format(p="model.generator") for s in SMALL_VQ] + ["model=videogpt_vq_vae"]
STAGE2 = ["model=discrete_diffusion"] + SMALL_DATA + [s.format(p="model.autoencoder") for s in SMALL_VQ] + SMALL_D3PM


def run_train(tmp, extra):
    from src.train import main
    return main(list(extra) + [f"paths.output_dir={tmp}", f"callbacks.model_checkpoint.dirpath={tmp}/checkpoints"])


def finite(metrics, keys):
    assert set(keys) <= set(metrics), sorted(metrics)
    assert all(torch.isfinite(torch.tensor(float(metrics[k]))) for k in keys)


def test_stage1_vqvae_training_entry_point(tmp_path):
    metrics = run_train(tmp_path, STAGE1 + ["trainer.max_epochs=2"])
    finite(metrics, ["total/train", "total/val", "l/dummy/train", "l/dummy/val", "recon/train", "commitment/val", "epoch", "step"])
    assert metrics["epoch"] == 1.0
    files = sorted(os.listdir(tmp_path / "checkpoints"))
    assert "last.ckpt" in files and sum(f.startswith("epoch_") for f in files) == 1                 # save_last + save_top_k = 1
    ck = torch.load(tmp_path / "checkpoints" / "last.ckpt", weights_only=True)
    assert ck["epoch"] == 1 and ck["global_step"] == 6 and len(ck["optimizer_states"]) == 1
    assert all(k.startswith("generator.") for k in ck["state_dict"])                               # the reference's stage-1 layout
    # ... which stage 2 loads as its autoencoder (checkpoint_paths.autoencoder, multistage_text_motion_model.py:113-122)
    m2 = run_train(tmp_path / "s2", STAGE2 + ["trainer.max_epochs=1", f"model.checkpoint_paths.autoencoder={tmp_path}/checkpoints/last.ckpt"])
    finite(m2, ["total/train", "total/val", "l/dummy/train"])


def _params(objs):
    return {k: v.detach().clone() for k, v in objs["model"].state_dict().items()}


@pytest.mark.parametrize("native", [False, True])
def test_stage2_resume_equals_uninterrupted_run(tmp_path, native):
    """train 2 epochs straight == train 1 epoch, stop, resume from last.ckpt for the 2nd (parameters, Lt_history, optimiser
    state): the checkpoint carries the Philox stream position and the torch RNG state next to Lightning's own fields."""
    from src.tasks.runner import train
    from gsdd_amd.hydra_lite import compose
    from src.train import ROOT
    flags = STAGE2 + [f"model.native_step={str(native).lower()}"]

    def go(tmp, extra):
        cfg = compose(os.path.join(ROOT, "configs"), "train.yaml", flags + list(extra) +
                      [f"paths.output_dir={tmp}", f"callbacks.model_checkpoint.dirpath={tmp}/checkpoints"])
        return train(cfg)
    m_full, o_full = go(tmp_path / "full", ["trainer.max_epochs=2"])
    _, o_half = go(tmp_path / "half", ["trainer.max_epochs=1"])
    m_res, o_res = go(tmp_path / "res", ["trainer.max_epochs=2", f"ckpt_path={tmp_path}/half/checkpoints/last.ckpt"])
    assert o_res["trainer"].current_epoch == 2 and o_res["trainer"].global_step == 6
    full, res, half = _params(o_full), _params(o_res), _params(o_half)
    dm_f, dm_r = o_full["model"].generator.diffusion_model, o_res["model"].generator.diffusion_model
    assert dm_f.noise_stream == dm_r.noise_stream and dm_f.noise_stream > 0
    # The weight gradients accumulate with float atomics, so two runs agree to rounding, not bit for bit -- and Adam turns the
    # rounding noise of a mathematically zero gradient (attn1.key.bias, the single-key cross-attention's q/k, ...) into +-lr steps.
    # Compare where the gradient is real: a probe gradient at the final weights says where that is.
    from gsdd_amd.d3pm_train import D3PMTrainer
    gen = o_full["model"].generator
    batch = next(iter(o_full["datamodule"].train_dataloader()))
    with torch.no_grad():
        tokens = o_full["model"].autoencoder.encode(batch["video"]).view(len(batch["text"]), -1)
    _, g0 = D3PMTrainer(gen.diffusion_model).loss_and_grads(tokens, gen._text(batch["text"], tokens.device))
    gmax = max(v.abs().max().item() for v in g0.values())
    pre = "generator.diffusion_model.transformer."
    moved = 0
    for k in full:
        if not full[k].is_floating_point() or "Lt_" in k:
            continue
        if k.startswith(pre):
            g = g0[k[len(pre):]]
            if g.abs().max().item() < 1e-4 * gmax:
                continue
            big = g.abs() > 1e-2 * g.abs().max()
            assert torch.allclose(res[k][big], full[k][big], atol=5e-6, rtol=0), k           # one Adam step moves an entry by ~1e-4
            moved += int((full[k][big] != half[k][big]).sum())
        else:
            assert torch.equal(res[k], full[k]), k       # autoencoder, text tower, schedule buffers: untouched by stage 2
    assert moved > 1000                                  # the second epoch did train
    for k in ("generator.diffusion_model.Lt_history", "generator.diffusion_model.Lt_count"):
        torch.testing.assert_close(res[k], full[k], rtol=1e-4, atol=0)
    assert abs(m_res["total/train"] - m_full["total/train"]) < 1e-4 * abs(m_full["total/train"])


def test_eval_entry_point(tmp_path, capsys):
    from src.eval import main
    metrics = main(STAGE2 + [f"paths.output_dir={tmp_path}"])
    finite(metrics, ["total/test", "l/dummy/test"])
    assert "3 batches" in capsys.readouterr().out


def test_evaluation_hooks_with_a_stand_in_encoder(tmp_path):
    """`do_evaluation: true` (text_motion_model.py:98-101, :118-121): validation batches are sampled, pushed through the evaluator
    (de-normalise, 224 resize on the preprocessing kernel, features) and the Frechet statistic is logged as `Metrics/fvd-val` at epoch
    end.  The I3D encoder cannot exist offline; the stand-in pooling encoder exercises the same plumbing."""
    metrics = run_train(tmp_path, STAGE1 + ["trainer.max_epochs=1", "model.do_evaluation=true",
                                            "model.evaluator.videoencoder._target_=src.utils.evaluator.MeanPoolEncoder"])
    assert "Metrics/fvd-val" in metrics and torch.isfinite(torch.tensor(metrics["Metrics/fvd-val"]))
    from src.utils.evaluator import Evaluator, MeanPoolEncoder
    ev = Evaluator("cuda", MeanPoolEncoder(), target_resolution=32)
    g = torch.Generator().manual_seed(0)
    clips = torch.randn(6, 3, 4, 16, 16, generator=g).cuda()
    ev.push_vals({"video": clips}, 0, clips)
    assert abs(ev.evaluate_metrics()["fvd"]) < 1e-3                     # identical sets: distance 0
    ev.reset()
    ev.push_vals({"video": clips}, 0, clips + 1.0)
    assert ev.evaluate_metrics()["fvd"] > 1e-2


def test_bench_two_ranks_rehearsal(tmp_path):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), rehearsed on this box's single GPU
    with the gloo backend (RCCL refuses two ranks on one device) at a toy size: the N > 1 code path -- device binding, process group,
    barrier, per-rank times gathered, rank 0's JSON line, weak scaling of `value` -- must run and parse."""
    import json
    import subprocess
    import sys
    from src.train import ROOT
    env = dict(os.environ, GSDD_DIST_BACKEND="gloo", GSDD_FORCE_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29800 + os.getpid() % 100), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--no-cpu-baseline", "--batch", "4", "--grid", "2", "4", "4", "--codes", "64", "--layers", "2", "--diffusion-steps", "8"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 8 and line["scaling"] == "weak"
    assert len(line["ranks"]["videos_per_s_per_rank"]) == 2 and line["ranks"]["seconds_max"] >= line["ranks"]["seconds_min"] > 0
    assert abs(line["value"] - 8 / line["ranks"]["seconds_max"]) < 0.05 * line["value"]          # (the seconds are rounded to 1e-4)
