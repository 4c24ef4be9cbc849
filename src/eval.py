"""`python src/eval.py model=discrete_diffusion ckpt_path=...` (reference: src/eval.py:8-13, src/tasks/eval_task.py:14-62)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PROJECT_ROOT", ROOT)
import src  # noqa: E402,F401
from gsdd_amd.hydra_lite import compose  # noqa: E402


def main(argv=None):
    cfg = compose(os.path.join(ROOT, "configs"), "eval.yaml", list(argv if argv is not None else sys.argv[1:]))
    from src.tasks.runner import evaluate
    return evaluate(cfg)


if __name__ == "__main__":
    main()
