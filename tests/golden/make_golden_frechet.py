#!/usr/bin/env python3
"""Generate tests/golden/frechet.npz with the REFERENCE's frechet_distance (src/utils/evaluator.py:118-179).  The module does
not import here (it pulls in the I3D extractor), so the four self-contained functions are cut out of the file at generation time
and executed in the build container only; the fixture holds the feature sets and the values the reference returned."""
import os
import re
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/src/utils/evaluator.py"


def reference_functions():
    text = open(REF).read()
    start = text.index("def _symmetric_matrix_square_root")
    m = re.search(r"^def frechet_distance\(.*?\n    return fd\n", text[start:], flags=re.S | re.M)
    end = start + m.end()
    ns = {"torch": torch}
    exec(compile(text[start:end], REF, "exec"), ns)
    return ns["frechet_distance"]


def main():
    fd = reference_functions()
    rng = np.random.default_rng(11)
    res = {}
    for i, (n1, n2, dim, shift) in enumerate([(64, 48, 12, 0.0), (200, 200, 32, 0.5), (40, 40, 40, 1.0)]):
        a = rng.standard_normal((n1, dim)).astype(np.float32) @ rng.standard_normal((dim, dim)).astype(np.float32)
        b = rng.standard_normal((n2, dim)).astype(np.float32) @ rng.standard_normal((dim, dim)).astype(np.float32) + np.float32(shift)
        res[f"a{i}"], res[f"b{i}"] = a, b
        res[f"fd{i}"] = np.float64(fd(a.copy(), b.copy()).item())
        print(i, res[f"fd{i}"])
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "frechet.npz"), **res)


if __name__ == "__main__":
    main()
