#!/usr/bin/env python3
"""Generate tests/golden/preprocess.npz by running the REFERENCE's clip preprocessing
(src/datamodules/datasets/ucf101_dataset.py:105-140, function `preprocess`).

The reference module itself does not import (SyntaxError at ucf101_dataset.py:88 and torchvision is absent), so the
function's text is cut out of the file at generation time and executed here, in the build container only; nothing of it is
stored in the repository — the fixture holds inputs (uint8 THWC clips) and the outputs the reference produced.

Usage:  python tests/golden/make_golden_preprocess.py
"""
import math
import os
import re
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/src/datamodules/datasets/ucf101_dataset.py"


def reference_preprocess():
    text = open(REF).read()
    m = re.search(r"^def preprocess\(.*?(?=^\S)", text, flags=re.S | re.M)
    if m is None:
        m = re.search(r"^def preprocess\(.*", text, flags=re.S | re.M)
    ns = {"torch": torch, "F": F, "math": math}
    exec(compile(m.group(0), REF, "exec"), ns)
    return ns["preprocess"]


def main():
    pre = reference_preprocess()
    rng = np.random.default_rng(7)
    res = {}
    # (T, H, W) -> (resolution, sequence_length): landscape / portrait / square, up- and down-scaling, odd sizes
    cases = [((6, 37, 53), 16, 4), ((5, 61, 40), 24, None), ((3, 20, 20), 32, 2), ((2, 120, 160), 64, 2), ((2, 17, 45), 17, None)]
    for i, (shape, r, sl) in enumerate(cases):
        video = rng.integers(0, 256, size=shape + (3,), dtype=np.uint8)
        with torch.no_grad():
            out = pre(torch.from_numpy(video), r, sl)
        res[f"in{i}"] = video
        res[f"out{i}"] = out.numpy().astype(np.float32)
        res[f"cfg{i}"] = np.array([r, -1 if sl is None else sl])
        print(i, shape, "->", tuple(out.shape))
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "preprocess.npz"), **res)


if __name__ == "__main__":
    main()
