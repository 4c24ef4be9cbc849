"""`python src/train.py model=videogpt_vq_vae ...` (reference: src/train.py:17-34)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsdd_amd.hydra_lite import compose  # noqa: E402  (importing gsdd_amd happens in src/__init__)
import src  # noqa: E402,F401


def main(argv=None):
    cfg = compose(os.path.join(ROOT, "configs"), "train.yaml", list(argv if argv is not None else sys.argv[1:]))
    from src.tasks.runner import train
    return train(cfg)


if __name__ == "__main__":
    main()
