"""The reference's command lines on the HIP path, at toy sizes: `python src/train.py model=videogpt_vq_vae`,
`python src/train.py model=discrete_diffusion`, `python src/eval.py` (reference: src/train.py:17-34, src/eval.py,
src/tasks/train_task.py:15-82, src/tasks/eval_task.py:14-62) through the Hydra-style composition of configs/."""
import re

import pytest
import torch

pytestmark = pytest.mark.gpu

SMALL_DATA = ["datamodule.resolution=32", "datamodule.sequence_length=4", "batch_size=2", "datamodule.n_batches=3"]
SMALL_VQ = ["{p}.n_hiddens=16", "{p}.n_codes=32", "{p}.embedding_dim=8", "{p}.n_res_layers=1"]
SMALL_D3PM = ["model.generator.diffusion_model.content_seq_len=64",
              "model.generator.diffusion_model.transformer.content_seq_len=64",
              "model.generator.diffusion_model.transformer.n_layer=2",
              "model.generator.diffusion_model.transformer.content_spatial_size=[8,8]",
              "model.generator.diffusion_model.transformer.dalle.num_embed=32",
              "model.generator.diffusion_model.transformer.dalle.spatial_size=[8,8]"]


def losses(text):
    return [float(m) for m in re.findall(r"loss ([-+0-9.eE]+|nan|inf)", text)]


def test_stage1_vqvae_training_entry_point(capsys):
    from src.train import main
    main(SMALL_DATA + [s.format(p="model.generator") for s in SMALL_VQ] + ["model=videogpt_vq_vae"])
    got = losses(capsys.readouterr().out)
    assert len(got) == 3 and all(torch.isfinite(torch.tensor(got)))


def test_stage2_d3pm_training_entry_point(capsys):
    from src.train import main
    main(["model=discrete_diffusion"] + SMALL_DATA + [s.format(p="model.autoencoder") for s in SMALL_VQ] + SMALL_D3PM)
    got = losses(capsys.readouterr().out)
    assert len(got) == 3 and all(torch.isfinite(torch.tensor(got)))


def test_eval_entry_point(capsys):
    from src.eval import main
    out = main(SMALL_DATA + [s.format(p="model.autoencoder") for s in SMALL_VQ] + SMALL_D3PM)
    assert out["clips"] == 6
    assert "clips/s" in capsys.readouterr().out
