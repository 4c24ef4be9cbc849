#!/usr/bin/env python3
"""How many (16-query x 32-key) tiles of the denoiser's self-attention actually hold a probability above 2^-PM of the row sum, per
layer, under the reference's init and under the trained-like weights of bench.py -- i.e. what ANY test (bound or measured) could let the
adaptive attention arithmetic skip.  B = 1, L = 4096, a mid-chain x_t."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsdd_amd  # noqa: E402
from gsdd_amd.d3pm_train import D3PMTrainer  # noqa: E402
import bench  # noqa: E402


def main():
    L, K, H = 4096, 4096, 16
    torch.manual_seed(0)
    d = gsdd_amd.DalleMaskImageEmbedding(num_embed=K, spatial_size=[64, 64], embed_dim=64)
    tr = gsdd_amd.Text2ImageTransformer(dalle=d, n_layer=19, n_embd=64, n_head=16, content_seq_len=L, block_activate="GELU2",
                                        content_spatial_size=[64, 64], condition_dim=512, diffusion_step=100)
    dm = gsdd_amd.DiffusionTransformer(transformer=tr, diffusion_step=100, alpha_init_type="alpha1", auxiliary_loss_weight=5e-4,
                                       adaptive_auxiliary_loss=True, guidance_scale=2, content_seq_len=L).cuda()
    g = torch.Generator().manual_seed(1)
    x0 = torch.randint(0, K, (1, L), generator=g).cuda()
    xt = torch.where(torch.rand((1, L), generator=g).cuda() < 0.5, torch.full_like(x0, K), x0)      # half masked
    cond = (torch.randn((1, 1, 512), generator=g) * 0.5).cuda()
    t = torch.tensor([50], device="cuda")
    for regime in ("init", "trained_like"):
        if regime == "trained_like":
            bench.trained_like_weights(dm)
        trainer = D3PMTrainer(dm, lr=1e-4)
        sv = trainer._forward(xt, cond, t)
        for li in (0, 1, 5, 10, 18):
            qkv = sv["layers"][li]["qkv"]                       # [3H][M][4]
            q, k = qkv[0:H].double(), qkv[H:2 * H].double()
            fr = {8: 0.0, 6: 0.0, 10: 0.0}
            nq = int(os.environ.get("GSDD_TILE_STATS_NQ", "512"))            # a sample of query rows per head
            s = (q[:, :nq] @ k.transpose(1, 2)) * 0.5 * 1.4426950408889634          # log2 domain, [H][nq][L]
            lse = torch.logsumexp(s * 0.6931471805599453, dim=-1, keepdim=True) * 1.4426950408889634
            rel = s - lse                                       # log2(p / rowsum)
            tile = rel.reshape(H, nq // 16, 16, L // 32, 32).amax(dim=(2, 4))        # max over the tile
            for pm in fr:
                fr[pm] = float((tile > -pm).double().mean())
            # what an A PRIORI bound could clear (a tile is cleared for a 16-query sub-tile when the bound proves every probability
            # of the tile below 2^-8 of the final row sum for all 16 queries; Jensen: log2 rowsum >= log2 L + q'.kbar; budget = log2 L - 8):
            #   A  the shipped form: ||q'|| max_tile ||k|| - q'.kbar < budget
            #   B  keys centred on the head's mean key: ||q'|| max_tile ||k - kbar|| < budget
            #   C  keys centred on their tile's mean c_u: q'.(c_u - kbar) + ||q'|| max_tile ||k - c_u|| < budget
            #   H  per head (one launch-time flag): max_q ||q'|| max_k ||k - kbar|| < budget
            #   T  the truth with the EXACT row sum (what no bound can beat): tile maximum of log2(p / rowsum) < -8
            import math
            qs = q[:, :nq] * (0.5 * 1.4426950408889634)                          # q' [H][nq][4]
            kbar = k.mean(dim=1, keepdim=True)                                  # [H][1][4]
            budget = math.log2(L) - 8.0
            qn_ = qs.norm(dim=-1)                                                # [H][nq]
            kt = k.reshape(H, L // 32, 32, 4)
            kn_t = kt.norm(dim=-1).amax(dim=-1)                                  # [H][tiles]
            kc_t = (kt - kbar[:, :, None, :]).norm(dim=-1).amax(dim=-1)
            cu = kt.mean(dim=2)                                                  # [H][tiles][4]
            ru = (kt - cu[:, :, None, :]).norm(dim=-1).amax(dim=-1)
            qk = (qs * kbar).sum(-1)                                             # q'.kbar [H][nq]
            bA = qn_[:, :, None] * kn_t[:, None, :] - qk[:, :, None]
            bB = qn_[:, :, None] * kc_t[:, None, :]
            bC = torch.einsum("hqd,hud->hqu", qs, cu - kbar) + qn_[:, :, None] * ru[:, None, :]
            #   D  what a producer-side statistic would give: the tile's radius about an anchor a_u the producer knows (the mean of the
            #      tile's first 16 keys) plus the anchor's distance from the mean key: ||q'|| (max_tile ||k - a_u|| + ||a_u - kbar||) < budget
            au = kt[:, :, :16].mean(dim=2)
            rau = (kt - au[:, :, None, :]).norm(dim=-1).amax(dim=-1)
            bD = qn_[:, :, None] * (rau + (au - kbar).norm(dim=-1))[:, None, :]
            sub = lambda z: z.reshape(H, nq // 16, 16, -1).amax(dim=2)           # worst query of the sub-tile
            clr = {n_: float((sub(z) < budget).double().mean()) for n_, z in (("A", bA), ("B", bB), ("C", bC), ("D", bD))}
            chunk = lambda z: float((sub(z).reshape(H, nq // 64, 4, -1)[..., :120].reshape(H, nq // 64, 4, 10, 12).amax(dim=(2, 4)) < budget).double().mean())
            clr["Bchunk"], clr["Dchunk"] = chunk(bB), chunk(bD)
            clr["H"] = float(((qn_.amax(dim=1) * (k - kbar).norm(dim=-1).amax(dim=1)) < budget).double().mean())
            clr["T"] = float((tile <= -8).double().mean())
            print(f"{regime:12s} layer {li:2d}: (sub-tile, tile) pairs cleared a priori: shipped bound {clr['A']:.3f}, keys centred on the head mean "
                  f"{clr['B']:.3f}, on the tile mean {clr['C']:.3f}; whole heads {clr['H']:.3f}; truth {clr['T']:.3f}; producer-side radius {clr['D']:.3f}; "
                  f"(wave, 384-key chunk) blocks cleared whole: head mean {clr['Bchunk']:.3f}, producer-side {clr['Dchunk']:.3f}", flush=True)
            # the kernel itself on exactly these q, k, v: adaptive (a8) against hi only (11) and hi + lo (22) -- what the bounds buy
            from gsdd_amd import ops
            RB = 16                                            # the row replicated: a launch of the bench's size per lane
            qf, kf, vf = (qkv[i * H:(i + 1) * H].repeat(1, RB, 1).contiguous() for i in range(3))
            outb = torch.empty((RB * L, H * 4), device="cuda")
            wsb = ops.d3pm_attention_workspace(RB, L, H, "cuda")
            tms = {}
            for mode in ("a8", "11", "22"):
                for _ in range(3):
                    ops.d3pm_attention(qf, kf, vf, RB, L, H, outb, ws=wsb, mode=mode)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    ops.d3pm_attention(qf, kf, vf, RB, L, H, outb, ws=wsb, mode=mode)
                e1.record()
                torch.cuda.synchronize()
                tms[mode] = e0.elapsed_time(e1) / 20 * 1e3
            print(f"{regime:12s} layer {li:2d}: kernel (the row x 16: prep + attention, us) adaptive {tms['a8']:.1f}  hi only {tms['11']:.1f}  hi + lo {tms['22']:.1f}", flush=True)
            qn, kn = q.norm(dim=-1), k.norm(dim=-1)
            print(f"{regime:12s} layer {li:2d}: |q| mean {float(qn.mean()):.2f} max {float(qn.max()):.2f}  |k| mean {float(kn.mean()):.2f} max "
                  f"{float(kn.max()):.2f}  score std {float((s * 0.693).std()):.2f}  tiles with p > 2^-6 / 2^-8 / 2^-10 of the row sum: "
                  f"{fr[6]:.3f} / {fr[8]:.3f} / {fr[10]:.3f}", flush=True)


if __name__ == "__main__":
    main()
