#!/usr/bin/env python3
"""Which near-tie Gumbel vector (tests/golden/neartie.npz) a build of the step kernel decides differently from another build, and
by what.  Runs the fixture's 64 guided (conditional + unconditional logits) positions through gsdd_d3pm_step under each library
given (GSDD_LIB_PATH), prints every vector on which the builds, the reference (fp32 torch CPU) or fp64 disagree, with the fp64
top-2 margin, and the posterior log-probabilities of the two contenders under each build (post_dbg hook).
usage: neartie_diff.py <tag>=<libgsdd.so> [<tag>=<libgsdd.so> ...]      (one child process per library)"""
import json
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def child(out_path):
    import torch
    import gsdd_amd
    from tests.conftest import load_golden
    sd, a, cfg = load_golden("neartie")
    B, L, K, T = cfg["B"], cfg["L"], cfg["K"], cfg["T"]
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    rows = lambda x: torch.from_numpy(x).permute(0, 2, 1).reshape(B * L, K).contiguous().cuda()
    sched = [sd[n].cuda() for n in gsdd_amd.d3pm.SCHED_ORDER]
    tok_in = dev(a["gum/xt"])
    tok_out = torch.empty_like(tok_in)
    t2 = torch.cat([dev(a["gum/t"]), dev(a["gum/t"])]).contiguous()
    sid = torch.tensor([int(a["gum/stream"])], dtype=torch.int64, device="cuda")
    post = torch.empty((B, K + 1, L), dtype=torch.float32, device="cuda")
    lc, lu = rows(a["gum/logits_c"]), rows(a["gum/logits_u"])
    kw = dict(K=K, T=T, guidance=float(cfg["guidance"]), seed=cfg["noise_seed"])
    gsdd_amd.ops.d3pm_step(lc, lu, tok_in, tok_out, sched, t2, sid, **kw)                    # the production instantiation (no hooks)
    tok_dbg = torch.empty_like(tok_in)
    gsdd_amd.ops.d3pm_step(lc, lu, tok_in, tok_dbg, sched, t2, sid, post_dbg=post, **kw)     # the same step with the posterior hook
    assert torch.equal(tok_out, tok_dbg), "the hooked instantiation decides differently from the production one"
    np.savez(out_path, tok=tok_out.cpu().numpy(), post=post.cpu().numpy())


def main():
    if sys.argv[1] == "--child":
        return child(sys.argv[2])
    from tests.conftest import load_golden
    _, a, cfg = load_golden("neartie")
    res = {}
    for spec in sys.argv[1:]:
        tag, lib = spec.split("=", 1)
        out = os.path.join(REPO, "gpurun_out", f"neartie_{tag}.npz")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", out], env=dict(os.environ, GSDD_LIB_PATH=os.path.abspath(lib)))
        res[tag] = np.load(out)
    m64, w64, s64, ref = np.abs(a["gum/margin64"]).reshape(-1), a["gum/winner64"].reshape(-1), a["gum/second64"].reshape(-1), a["gum/ref_tok"].reshape(-1)
    L = cfg["L"]
    report = {"vectors": int(m64.size)}
    for tag, r in res.items():
        tok = r["tok"].reshape(-1)
        report[tag] = {"eq_reference": int((tok == ref).sum()), "eq_fp64": int((tok == w64).sum())}
    tags = list(res)
    rows = []
    for i in range(m64.size):
        toks = {t: int(res[t]["tok"].reshape(-1)[i]) for t in tags}
        if len(set(toks.values())) > 1 or any(v != int(ref[i]) for v in toks.values()) or int(ref[i]) != int(w64[i]):
            b, l = divmod(i, L)
            row = {"vector": i, "fp64_margin": float(m64[i]), "fp64_winner": int(w64[i]), "fp64_second": int(s64[i]), "reference": int(ref[i]), "tokens": toks}
            for t in tags:
                p = res[t]["post"]
                row[f"post_{t}"] = [float(p[b, int(w64[i]), l]), float(p[b, int(s64[i]), l])]
            if len(tags) == 2:
                pa, pb = res[tags[0]]["post"][b, :, l], res[tags[1]]["post"][b, :, l]
                row["post_rows_differ_in"] = int((pa != pb).sum())
                row["post_max_abs_diff"] = float(np.abs(pa - pb).max())
            rows.append(row)
    report["disagreements"] = rows
    print(json.dumps(report, indent=1))
    with open(os.path.join(REPO, "gpurun_out", "neartie_diff.json"), "w") as f:
        json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
