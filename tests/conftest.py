import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """-> (state_dict of torch tensors, dict of numpy arrays, cfg dict) from tests/golden/<name>.npz"""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    cfg = {k[4:]: (z[k].tolist() if z[k].ndim else z[k].item()) for k in z.files if k.startswith("cfg_")}
    arrs = {k: z[k] for k in z.files if not k.startswith("sd/") and not k.startswith("cfg_")}
    return sd, arrs, cfg


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get


def parity_report(name, record):
    """Append one measured-parity record (mismatch counts, minimum margins, achieved errors) to the JSON the GPU tests leave
    behind: $GSDD_PARITY_REPORT, default gpurun_out/parity_report.json (the directory gpurun copies back; the file is then
    committed as profiles/rN_parity_report.json).  Also printed, so `pytest -s` / the captured log shows the numbers."""
    import json
    path = os.environ.get("GSDD_PARITY_REPORT", os.path.join(REPO, "gpurun_out", "parity_report.json"))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = {}
    if os.path.exists(path):
        try:
            with open(path) as f:
                data = json.load(f)
        except Exception:
            data = {}

    def plain(v):
        if isinstance(v, (np.floating, np.integer)):
            return v.item()
        if torch.is_tensor(v) or isinstance(v, np.ndarray):
            return plain(v.tolist()) if getattr(v, "ndim", 1) else v.item()
        if isinstance(v, (list, tuple)):
            return [plain(x) for x in v]
        return v
    data[name] = {k: plain(v) for k, v in record.items()}
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print(f"[parity] {name}: " + ", ".join(f"{k}={data[name][k]}" for k in sorted(data[name])))
