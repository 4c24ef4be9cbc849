"""`python src/train.py model=videogpt_vq_vae ...` / `model=discrete_diffusion ...` (reference: src/train.py:17-34).
Multi-GPU: `python -m torch.distributed.run --nproc-per-node N src/train.py ...` (one process per GPU, RCCL)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PROJECT_ROOT", ROOT)       # configs/paths/default.yaml (the reference sets it through pyrootutils)
from gsdd_amd.hydra_lite import compose  # noqa: E402  (importing gsdd_amd happens in src/__init__)
import src  # noqa: E402,F401


def main(argv=None):
    cfg = compose(os.path.join(ROOT, "configs"), "train.yaml", list(argv if argv is not None else sys.argv[1:]))
    from src.tasks.runner import train
    metric_dict, _ = train(cfg)
    return metric_dict


if __name__ == "__main__":
    main()
