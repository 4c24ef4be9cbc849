#!/usr/bin/env python3
"""Headline benchmark: 16-frame 128x128 videos / second for a 100-step D3PM sample (BASELINE.json).

A "step" = one pass of the hot path over one batch: the full 100-step classifier-free-guided reverse
loop over a 16x16x16 token grid for `--batch` clips (config C3: bs 16, guidance 2, 19-layer denoiser,
K = 4096) followed by the VQ-VAE decode of those clips to (3,16,128,128).  Synthetic data, random-init
weights of the reference architecture.  One process per GPU; ranks sample independent clips (weak scaling,
no data-path collective).

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
import gsdd_amd  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense f32 MFMA peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 / f16 MFMA peak
# GSDD_ATTN_P -> (kernel template argument, what the softmax probabilities are carried as into the P.V product)
P_MODES = {"a8": (8, "f16 hi (11 bits), + f16 lo (22 bits) in every (16-query, 32-key) tile that can hold a probability above 2^-8 of "
                     "its row's sum -- cleared by a ||q|| ||k|| bound where that proves it cannot, measured otherwise (logits error vs "
                     "the fp32 oracle at full size: 8e-6 init weights, 1.3e-5 trained-like; all-tiles hi+lo = GSDD_ATTN_P=22: 1.3e-6 / "
                     "8.5e-6)"),
           "a12": (12, "f16 hi, + lo where a probability exceeds 2^-12 of the row's running sum"),
           "22": (1, "f16 hi + lo (22 bits) everywhere"), "11": (0, "f16 hi only (11 bits)")}
HBM_PEAK_GBS = 8000.0


def attention_mode_name(dm):
    """'a8' | '22' | '11' | 'a12': what the sampler's attention launches run as at L >= 2048 (module attribute, else GSDD_ATTN_P, else a8)."""
    return str(dm.transformer.attention_mode or os.environ.get("GSDD_ATTN_P", "a8"))


def build_models(args, device):
    torch.manual_seed(0)
    L = args.grid[0] * args.grid[1] * args.grid[2]
    side = 1
    while side * side < L:
        side *= 2
    dalle = gsdd_amd.DalleMaskImageEmbedding(num_embed=args.codes, spatial_size=[side, side], embed_dim=64)
    tr = gsdd_amd.Text2ImageTransformer(dalle=dalle, condition_seq_len=77, n_layer=args.layers, n_embd=64, n_head=16,
                                        content_seq_len=L, mlp_hidden_times=4, block_activate="GELU2",
                                        attn_type="selfcross", content_spatial_size=[side, side], condition_dim=512,
                                        diffusion_step=args.diffusion_steps, timestep_type="adalayernorm")
    dm = gsdd_amd.DiffusionTransformer(transformer=tr, diffusion_step=args.diffusion_steps, alpha_init_type="alpha1",
                                       auxiliary_loss_weight=5.0e-4, adaptive_auxiliary_loss=True, mask_weight=[1, 1],
                                       learnable_cf=False, guidance_scale=2, content_seq_len=L)
    frames, res = args.grid[0], args.grid[1] * 8
    vq = gsdd_amd.VQVAE(None, 128, args.codes, 256, 3, [1, 8, 8], frames, res)
    return dm.to(device).eval(), vq.to(device).eval(), L


def profile_traffic(kernel, grid):
    """HBM bytes per launch of `kernel` at `grid` workgroups from the newest profiles/r*_pmc_traffic.csv that lists it (rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE passes, committed with the round, one row per kernel and grid size -- tools/make_traffic_csv.py;
    bytes = (2*FETCH_SIZE + WRITE_SIZE) KB, the gfx950 correction of MI355X_MICROARCH.md).  -> (bytes, file name) or (None, None):
    the number is read, never typed in."""
    import csv
    import glob
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.csv")), reverse=True):
        with open(f) as fh:
            rows = [r for r in csv.DictReader(l for l in fh if not l.startswith("#"))]
        for r in rows:
            if r["kernel"].replace(" ", "") == kernel.replace(", ", ";").replace(" ", "") and int(r.get("grid_workgroups", grid)) == grid:
                return (2 * float(r["FETCH_SIZE_KB"]) + float(r["WRITE_SIZE_KB"])) * 1024, os.path.basename(f)
    return None, None


def profile_counters(kernel, grid=None, regime="flat"):
    """Issue-port / matrix-pipe occupancy of `kernel` from the newest profiles/r*_pmc_sq_counters.csv that lists it in the per-kernel
    layout of tools/make_sq_csv.py (round 4 on): one row per (regime, kernel, grid), the means per dispatch of a rocprofv3 --pmc pass
    over THIS program (`python3 bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-extra`, started directly after `--`;
    tools/profile_round.sh).  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over the dispatch's SQ_BUSY_CYCLES / 32 shader engines;
    valu_issue_busy = 4 * SQ_ACTIVE_INST_VALU (quad-cycles) / 1024 over the same.  Read, never typed in.  -> dict or None."""
    import csv
    import glob
    want = kernel.replace(", ", ";").replace(" ", "")
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_sq_counters.csv")), reverse=True):
        with open(f) as fh:
            rows = [r for r in csv.DictReader(l for l in fh if not l.startswith("#"))]
        if not rows or "valu_issue_busy" not in rows[0]:
            continue
        for r in rows:
            if (r["kernel"].replace(" ", "").startswith(want) and r["regime"] == regime
                    and (grid is None or int(r["grid_workgroups"]) == grid)):
                out = {"mfma_busy": float(r["mfma_busy"]), "valu_issue_busy": float(r["valu_issue_busy"]),
                       "valu_insts_per_dispatch": float(r["SQ_INSTS_VALU"]), "dispatches": int(r["dispatches"]),
                       "grid_workgroups": int(r["grid_workgroups"]), "source": os.path.basename(f)}
                return out
    return None


def rooflines(dm, vq, cond, cf_cond, B, L, H, device, K, grid, reps=3, regime="flat"):
    """Per-launch timing of the kernels of a reverse step, live, with HIP events on the launch stream, in situ: right after the
    timed region `reps` more guided denoiser passes run eagerly as ONE full batch of 2B rows on one stream (the timed region
    itself may run two concurrent lanes, where a launch overlaps the other lane's kernels and its duration stops measuring the
    kernel), with an event pair around every launch of interest: the operands, cache state and clocks are the loop's.
    Dominant kernel = self-attention (head dim 4): algorithmic FLOPs = 16*L^2 per (sample, head) (QK^T 2*4 + PV 2*4 per score);
    from block 1 on K and V arrive pre-split from the fused layer kernel, so the event pair brackets the attention kernel alone.
    -> (roofline of the dominant kernel, roofline_families: fused layer (GEMM family), logits, posterior step, decode)."""
    from gsdd_amd import ops
    tr = dm.transformer
    B2 = 2 * B
    st = dm._streams[0]
    T = dm.num_timesteps
    with torch.cuda.stream(st):
        conds = torch.cat([cond, cf_cond], 0).contiguous()
        condv = tr.cond_vectors(conds)
        ws = tr.workspace(B2, L, device, rep=2)
        tok = torch.randint(0, K, (B, L), device=device)
        tok[torch.rand((B, L), device=device) < 0.5] = K               # half [MASK], like the middle of a chain
        t2 = torch.full((B2,), T // 2, dtype=torch.int64, device=device)
        sid = torch.zeros((1,), dtype=torch.int64, device=device)
        tok_out = torch.empty_like(tok)
        M = B * L

        def one():
            logits = tr.run(tok, condv, 1, t2, ws, rep=2, stream=st)
            ev = ws.get("events")
            if ev is not None:
                ev.setdefault("step", []).append((ops.Event(), ops.Event()))
                ev["step"][-1][0].record(st)
            ops.d3pm_step(logits[:M], logits[M:], tok, tok_out, dm._sched(), t2, sid, K=K, T=T, guidance=float(dm.guidance_scale),
                          seed=1, row0=0, stream=st)
            if ev is not None:
                ev["step"][-1][1].record(st)

        one()                                                           # warm
        events = {}
        ws["events"] = events
        for _ in range(reps):
            one()
        ws.pop("events")
    st.synchronize()

    def mean_ms(name):
        ev = events[name]
        return sum(e0.elapsed_ms(e1) for e0, e1 in ev) / len(ev), len(ev)

    pm = P_MODES[attention_mode_name(dm)][0]
    kernel = f"d3pm_attention_v4_kernel<384, {pm}>"
    ms, n = mean_ms("attention")
    flops = 16.0 * L * L * H * B2
    tf = flops / (ms * 1e-3) / 1e12
    traffic, src = profile_traffic(kernel, B2 * H * ((L + 255) // 256)) if (B2, L, H) == (32, 4096, 16) else (None, None)
    full = (B2, L, H) == (32, 4096, 16)
    ctr = profile_counters(kernel, B2 * H * ((L + 255) // 256), regime) if full else None
    if ctr is not None:                           # one score = one (query, key) pair: 64 of them per wave-instruction slot
        ctr["valu_insts_per_64_scores"] = round(ctr["valu_insts_per_dispatch"] / (float(L) * L * H * B2 / 64.0), 3)
    # what the hardware says (the kernel issues bf16 / f16 MFMAs and is bound by the vector issue port of the f32 datapath: one v_exp_f32
    # and the f32 -> f16 conversions per score): `achieved / peak / frac` stay the algorithmic f32-equivalent yardstick (16 L^2 FLOP per
    # (sample, head) against the dense f32 matrix peak); `counters` carry the measured matrix-pipe and vector-issue occupancy, and
    # `frac_of_bf16_mfma_peak` prices the same FLOPs against the 2.5 PFLOP/s of the pipe the MFMAs actually run on
    roof = {"bound": "valu-issue (f32 datapath); matrix work on bf16/f16 MFMA", "kernel": kernel, "achieved": round(tf, 2),
            "peak": PEAK_F32_MFMA_TFLOPS, "peak_kind": "dense f32 matrix peak (f32-equivalent yardstick)", "unit": "TFLOP/s",
            "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4), "frac_of_bf16_mfma_peak": round(tf / PEAK_BF16_MFMA_TFLOPS, 4),
            "counters": ctr, "traffic": traffic, "traffic_source": src, "ms_per_launch": round(ms, 4),
            "launches_timed": n, "flops_per_launch": flops,
            "ms_by_block": [round(sum(e0.elapsed_ms(e1) for e0, e1 in events["attention"][i::n // reps]) / reps, 3) for i in range(n // reps)],
            "timed_as": f"one full batch of {B2} rows on one stream, after the timed region"}

    fam = {}
    Mrows = B2 * L
    ms, n = mean_ms("layer")                      # proj + MLP + next q|k|v (blocks 0..n-2: the full chain of four GEMMs)
    fl = 2.0 * (64 * 64 + 2 * 64 * 256 + 64 * 192) * Mrows
    fam["gemm_family_fused_layer"] = {"bound": "mfma (f16 hi + lo operands: 3 f16 MFMA products per f32-equivalent product)",
                                      "kernel": "d3pm_layer_h2_kernel<true, false>", "ms_per_launch": round(ms, 4),
                                      "launches_timed": n, "achieved": round(fl / ms / 1e9, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                                      "unit": "TFLOP/s", "frac": round(fl / ms / 1e9 / PEAK_F32_MFMA_TFLOPS, 4),
                                      "frac_of_bf16_mfma_peak": round(3 * fl / ms / 1e9 / PEAK_BF16_MFMA_TFLOPS, 4),
                                      "counters": profile_counters("d3pm_layer_h2_kernel<true;false>", 256, regime) if full else None}
    ms, n = mean_ms("logits")
    fl = 2.0 * 64 * K * Mrows
    fam["logits"] = {"bound": "mfma (exact f32: v_mfma_f32_32x32x2_f32, the f32 datapath's own peak) / hbm write",
                     "kernel": "d3pm_logits_kernel", "ms_per_launch": round(ms, 4), "launches_timed": n,
                     "achieved": round(fl / ms / 1e9, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(fl / ms / 1e9 / PEAK_F32_MFMA_TFLOPS, 4),
                     "hbm_write_GBs": round(Mrows * K * 4.0 / ms / 1e6, 1), "hbm_frac": round(Mrows * K * 4.0 / ms / 1e6 / HBM_PEAK_GBS, 4),
                     "counters": profile_counters("d3pm_logits_kernel", 256, regime) if full else None}
    ms, n = mean_ms("step")
    by = 2.0 * M * K * 4
    fam["posterior_step"] = {"bound": "hbm (streams both logits copies once); the f32 vector issue port is what limits it",
                             "kernel": "d3pm_step_kernel", "ms_per_launch": round(ms, 4), "launches_timed": n,
                             "achieved": round(by / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(by / ms / 1e6 / HBM_PEAK_GBS, 4),
                             "counters": profile_counters("d3pm_step_kernel", (M + 3) // 4, regime) if full else None}
    # decode: the whole VQ-VAE decoder (implicit-GEMM convs), 221.0 GFLOP per 16x128x128 clip
    codes = torch.randint(0, K, (B,) + tuple(grid), device=device)
    vq.decode(codes)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        vq.decode(codes)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    if tuple(grid) == (16, 16, 16):
        fl = 221.0e9 * B
        fam["decode"] = {"bound": "mfma (bf16x3 operands: 6 bf16 MFMA products per f32-equivalent product)",
                         "kernel": "gemm_kernel (VQ-VAE decoder, all launches)", "ms_per_call": round(ms, 3),
                         "achieved": round(fl / ms / 1e9, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(fl / ms / 1e9 / PEAK_F32_MFMA_TFLOPS, 4),
                         "frac_of_bf16_mfma_peak": round(6 * fl / ms / 1e9 / PEAK_BF16_MFMA_TFLOPS, 4),
                         # the decoder's dominant launch (the 256 -> 256 transposed-conv phases at 64x64 / 32x32)
                         "counters": profile_counters("gemm_kernel<128;true;256;512>", 2048, regime)}
    return roof, fam


def cpu_baseline(args, dm, vq, L):
    """The CPU oracle (our restatement of the reference path, torch-CPU ops) on a bounded sample of the same workload, as SURVEY.md
    section 8(d) prescribes: 3 guided reverse steps at B=1 (2 denoiser passes + posterior + Gumbel each) extrapolated x T/3, plus
    one decode; next to it the reference's shipped-default length L=1024 (same weights, position table cut to 1024)."""
    from oracle import d3pm as od, vqvae as ov
    sd = {k: v.detach().cpu() for k, v in dm.state_dict().items()}
    vsd = {k: v.detach().cpu() for k, v in vq.state_dict().items()}
    cfg = dict(downsample=[1, 8, 8], n_res_layers=3)
    K = args.codes
    T = args.diffusion_steps
    g = torch.Generator().manual_seed(1)

    def steps(Lx, n):
        tok = torch.randint(0, K, (1, Lx), generator=g)
        tok[torch.rand(1, Lx, generator=g) < 0.5] = K
        cond = torch.randn(1, 1, 512, generator=g)
        t0 = time.perf_counter()
        for i in range(n):
            t = torch.full((1,), T // 2 - i, dtype=torch.long)
            tok, _ = od.p_sample_step(tok, cond, torch.zeros_like(cond), t, sd, 2.0, None, i)
        return (time.perf_counter() - t0) / n

    with torch.no_grad():
        n = 3
        t_step = steps(L, n)
        codes = torch.randint(0, K, (1,) + tuple(args.grid), generator=g)
        t0 = time.perf_counter()
        ov.decode(codes, vsd, cfg)
        t_dec = time.perf_counter() - t0
        t_1024 = steps(1024, n) if L > 1024 else None
    per_video = T * t_step + t_dec
    out = {"value": 1.0 / per_video, "unit": "videos/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{n} guided reverse steps at B=1 L={L} ({t_step:.2f} s each) x{T}/{n} + 1 decode ({t_dec:.2f} s), "
                     "torch-CPU oracle"}
    if t_1024 is not None:
        out["shipped_default_L1024"] = {"value": 1.0 / (T * t_1024 + t_dec), "unit": "videos/s",
                                        "sample": f"{n} guided reverse steps at B=1 L=1024 ({t_1024:.2f} s each) x{T}/{n} + the same decode"}
    return out


def other_configs(args, dm, vq, device):
    """BASELINE.json's other single-GPU configurations, outside the timed headline, each one warm-up + timed repetitions in this
    process so that the driver's run carries their numbers: C2 = VQ-VAE training step (bs 64, 16x128x128 clips, 256 channels, 3
    residual blocks, 4096 codes; forward with batch statistics + codebook EMA, full backward, Adam; 88.5 TFLOP per step, SURVEY.md
    section 8(d)), C4 = D3PM training step at its per-GPU shape (bs 16 of the global 128; q_sample, 19-layer denoiser forward +
    backward, loss, Adam; 16 x 278 GFLOP = 4.45 TFLOP), C5 = text-conditioned sampling at its per-GPU shape (bs 8 of the global 64:
    captions -> text provider -> DiscreteDiffusion(zero_text_emb=False).sample_videos -> decode).  Synthetic data, random-init weights."""
    import statistics
    out = {}

    def timeit(fn, warm, n):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return ts

    try:                                                             # ---- C5 per rank (before anything touches dm's weights)
        sys.path.insert(0, REPO)
        from src.models.text_models.clip_text_embedding import CLIPTextEmbedding
        dd = gsdd_amd.DiscreteDiffusion(CLIPTextEmbedding(clip_dim=512), dm, zero_text_emb=False).to(device).eval()
        rank5, b5 = 5, 8
        texts = [f"a person is doing activity number {rank5 * b5 + i}" for i in range(b5)]
        keep = (dm.noise_seed, dm.noise_stream, dm.row_offset)
        dm.set_noise(1234, 0, row_offset=rank5 * b5)
        ts = timeit(lambda: dd.sample_videos(texts, vq), 1, 2)
        dm.set_noise(*keep)
        out["c5_rank"] = {"workload": "C5 per GPU: MSR-VTT-shaped text-conditioned 100-step sample, bs 8 (64 sharded x8), guidance 2, "
                                      "L=4096, K=4096, 19 layers + decode; text provider = deterministic stand-in (CLIP ViT-B/32 is not "
                                      "obtainable offline)",
                          "value": round(b5 / statistics.mean(ts), 4), "unit": "videos/s", "s_per_pass": [round(t, 4) for t in ts],
                          "sampler_lanes": dm._last_lanes, "guidance_copies": 1 if dm._last_cfg_dedupe else 2,
                          "frac_f32_matrix_peak": round(b5 * 18.76 / statistics.mean(ts) / PEAK_F32_MFMA_TFLOPS, 4)}
    except Exception as e:                                           # noqa: BLE001
        out["c5_rank"] = {"error": f"{type(e).__name__}: {e}"}

    try:                                                             # ---- C4 per rank
        from gsdd_amd.d3pm_train import D3PMTrainer
        targs = argparse.Namespace(**vars(args))
        dmt, _, L = build_models(targs, device)
        dmt.train()
        trainer = D3PMTrainer(dmt, lr=1e-4)
        b4 = 16
        g = torch.Generator().manual_seed(1)
        tok = torch.randint(0, args.codes, (b4, L), generator=g).to(device)
        cond4 = torch.zeros(b4, 1, 512, device=device)               # the reference's zeroed text embedding (discrete_diffusion.py:25)
        losses = []
        ts = timeit(lambda: losses.append(trainer.step(tok, cond4)), 3, 3)          # (two eager steps, then the capture: replays from the fourth)
        ms = statistics.median(ts) * 1e3
        tflop = b4 * 0.278
        out["c4"] = {"workload": "C4 per GPU: D3PM training step, bs 16 (global 128 on 8 GPUs), 16x16x16 tokens, K=4096, 19 layers, "
                                 "T=100: q_sample + forward + loss + backward + Adam", "ms_per_step": round(ms, 2),
                     "ms_each": [round(t * 1e3, 2) for t in ts], "samples_per_s": round(b4 / ms * 1e3, 2), "tflop_per_step": tflop,
                     "achieved_tflops": round(tflop / ms * 1e3, 2), "frac_f32_matrix_peak": round(tflop / ms * 1e3 / PEAK_F32_MFMA_TFLOPS, 4),
                     "loss_first_last": [round(float(losses[0][0]), 4), round(float(losses[-1][0]), 4)],
                     "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
        del trainer, dmt
    except Exception as e:                                           # noqa: BLE001
        out["c4"] = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()

    try:                                                             # ---- C2
        from gsdd_amd.vqvae_trainer import VQVAETrainer
        torch.manual_seed(0)
        vqt = gsdd_amd.VQVAE(None, 128, args.codes, 256, 3, [1, 8, 8], 16, 128).to(device).train()
        trainer = VQVAETrainer(vqt, lr=4e-4)
        b2 = 64
        g = torch.Generator().manual_seed(1)
        x = torch.randn((b2, 3, 16, 128, 128), generator=g).to(device)
        recon = []
        torch.cuda.reset_peak_memory_stats()
        ts = timeit(lambda: recon.append(trainer.step(x)["recon_loss"]), 1, 2)
        sec = statistics.mean(ts)
        out["c2"] = {"workload": "C2: VQ-VAE training step, bs 64, UCF101-shaped synthetic 16x128x128 clips, n_hiddens 256, "
                                 "n_res_layers 3, 4096 codes: forward (batch statistics, codebook EMA) + backward + Adam",
                     "s_per_step": round(sec, 4), "s_each": [round(t, 4) for t in ts], "clips_per_s": round(b2 / sec, 2),
                     "tflop_per_step": 88.5, "achieved_tflops": round(88.5 / sec, 2),
                     "frac_f32_matrix_peak": round(88.5 / sec / PEAK_F32_MFMA_TFLOPS, 4),
                     "recon_loss": [round(float(r), 4) for r in recon],
                     "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
        del trainer, vqt, x
    except Exception as e:                                           # noqa: BLE001
        out["c2"] = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()

    try:                                                             # ---- the reference's shipped default length, L = 1024
        sargs = argparse.Namespace(**vars(args))
        sargs.grid = [4, 16, 16]                                     # content_seq_len 1024 (diffusion_transformer.yaml:8 there), clips of 4 x 128 x 128
        dms, vqs, Ls = build_models(sargs, device)
        bs = args.batch
        gs = torch.Generator().manual_seed(7)
        conds = torch.randn(bs, 1, 512, generator=gs).to(device)
        cfs = torch.zeros_like(conds)
        dms.set_noise(1234, 0)

        def pass_s():
            o = dms.sample(["synthetic"] * bs, None, conds, cfs, content_token=None, filter_ratio=0)
            return vqs.decode(o["content_token"].view(bs, *sargs.grid))
        ts = timeit(pass_s, 1, 2)
        out["shipped_default_L1024"] = {"workload": "the reference's shipped sequence length: 100 guided steps over 1024 tokens (4x16x16 grid), bs "
                                                    f"{bs}, 19 layers, K={args.codes}, + decode to 3x4x128x128 (cpu_baseline.shipped_default_L1024 is the "
                                                    "oracle at this length)",
                                        "value": round(bs / statistics.mean(ts), 3), "unit": "videos/s", "s_per_pass": [round(t, 4) for t in ts],
                                        "sampler_lanes": dms._last_lanes}
        del dms, vqs
    except Exception as e:                                           # noqa: BLE001
        out["shipped_default_L1024"] = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()
    return out


LAYER_ARITH = {
    "h2": "per-block GEMMs (proj, MLP, next q|k|v): operands as f16 hi + lo (22 bits, every product exact in the f32 accumulator; "
          "as accurate as an f32 GEMM: DESIGN.md section 4; GSDD_LAYER=x3p selects the bf16x3 kernel)",
    "x3p": "per-block GEMMs as 3-way bf16 splits too (GSDD_LAYER=x3p)",
}


def trained_like_weights(dm, seed=1):
    """The `scale_weights` recipe of tests/test_gpu_fullsize.py::full_d3pm: weights of a trained-like magnitude (Linear ~
    N(0, 1/fan_in), biases 0.1 N(0,1), embeddings 0.5 N(0,1)) instead of the reference's N(0, 0.02) init, under which the softmax
    rows are nearly flat.  The attention kernel's cost is data-dependent only through its overflow-redo branch."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in dm.transformer.modules():
            if isinstance(mod, torch.nn.Linear):
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=g) * (1.0 / mod.in_features ** 0.5))
                mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
            elif isinstance(mod, torch.nn.Embedding):
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=g) * 0.5)
    dm.transformer._packed = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16, help="clips per GPU per step (config C3: 16)")
    ap.add_argument("--grid", type=int, nargs=3, default=[16, 16, 16], help="latent token grid t h w")
    ap.add_argument("--codes", type=int, default=4096)
    ap.add_argument("--layers", type=int, default=19)
    ap.add_argument("--diffusion-steps", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--lanes", type=int, default=2, help="concurrent sub-batches of the sampler on separate HIP streams (the sampler's "
                                                        "default: two when the batch allows it; tokens do not depend on it)")
    ap.add_argument("--no-extra", action="store_true", help="skip the side regimes (trained-like weights, zero cond) and the other configs")
    ap.add_argument("--attention-mode", default=None, choices=["a8", "22", "11", "a12"],
                    help="softmax P format of the sampler's attention (default: the library's, a8 at this length); DESIGN.md section 4")
    ap.add_argument("--trained-like", action="store_true", help="profiling passes only: run the whole bench on weights of a trained-like "
                                                                "magnitude (peaky softmax rows) instead of the reference init")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("GSDD_DIST_BACKEND", "nccl")          # "gloo" only to rehearse N>1 on a 1-GPU box
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch N > 1 as `python -m torch.distributed.run "
              f"--nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...` (one process per GPU)", file=sys.stderr)
    if os.environ.get("GSDD_FORCE_DEVICE") is not None:            # rehearsal: several ranks on one card
        local = int(os.environ["GSDD_FORCE_DEVICE"])
        os.environ["LOCAL_RANK"] = str(local)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device: the hot path has no CPU fallback")
    if local >= torch.cuda.device_count():
        raise SystemExit(f"LOCAL_RANK {local} but {torch.cuda.device_count()} GPU(s) visible: one process per GPU is the layout")
    torch.cuda.set_device(local)                                   # before any allocation and before the process group exists
    if world > 1:
        from gsdd_amd.parallel import init_distributed
        init_distributed(backend)                                  # raises with the rendezvous it tried if the backend cannot start
    device = torch.device("cuda", local)

    dm, vq, L = build_models(args, device)
    if args.trained_like:
        trained_like_weights(dm)
    dm.transformer.attention_mode = args.attention_mode
    B = args.batch
    texts = ["synthetic"] * B
    g = torch.Generator().manual_seed(100 + rank)
    cond = torch.randn(B, 1, 512, generator=g).to(device)          # general conditioning (not the zeroed case)
    cf_cond = torch.zeros(B, 1, 512, device=device)
    dm.set_noise(1234, 0, row_offset=rank * B)
    dm.sample_lanes = args.lanes

    dm_last_tokens = [None]

    def one_pass():
        out = dm.sample(texts, None, cond, cf_cond, content_token=None, filter_ratio=0, use_graph=not args.no_graph)
        dm_last_tokens[0] = out["content_token"]
        clips = vq.decode(out["content_token"].view(B, *args.grid))
        return clips

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        clips = one_pass()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        clips = one_pass()
    barrier()
    dt = time.perf_counter() - t0
    headline_redo = dm.attention_redo_events()                     # of the last timed pass (device counters of its workspaces)
    lanes_used = dm._last_lanes
    tok = dm_last_tokens[0]
    assert tuple(clips.shape) == (B, 3, args.grid[0], args.grid[1] * 8, args.grid[2] * 8)
    assert torch.isfinite(clips).all()
    assert tok.dtype == torch.int64 and int(tok.min()) >= 0 and int(tok.max()) < args.codes, "a [MASK] or out-of-range token survived"
    rank_s = [dt]
    if world > 1:
        import torch.distributed as dist
        tt = torch.zeros(world, device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        tt[rank] = dt
        dist.all_reduce(tt)                                         # every rank's own time (sum of one-hot rows)
        rank_s = tt.tolist()
        dt = max(rank_s)

    if rank == 0:
        value = B * world * args.steps / dt
        line = {
            "metric": "16-frame 128x128 videos/sec, 100-step D3PM sample", "value": round(value, 4),
            "unit": "videos/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C3: D3PM {args.diffusion_steps}-step p_sample, guidance 2, "
                                   f"{args.grid[0]}x{args.grid[1]}x{args.grid[2]} token grid, K={args.codes}, "
                                   f"{args.layers} layers, bs {B}/GPU + VQ-VAE decode to 3x{args.grid[0]}x"
                                   f"{args.grid[1] * 8}x{args.grid[2] * 8}",
                       "global_batch": B * world, "parallelism": f"replicas x{world} (batch-sharded, no collective)",
                       "hipgraph": not args.no_graph, "sampler_lanes": lanes_used,
                       "weights": "trained-like magnitude (--trained-like)" if args.trained_like else "reference init N(0, 0.02)",
                       "arith": "f32 results; QK^T and VQ-VAE GEMM operands as error-free 3-way bf16 splits on the matrix pipe (dropped "
                                "terms < 2^-24); to_logits on the exact-f32 matrix instruction; " +
                                LAYER_ARITH[os.environ.get("GSDD_LAYER", "h2")] + "; softmax P: " + P_MODES[attention_mode_name(dm)][1]},
            "ranks": {"seconds_max": round(max(rank_s), 4), "seconds_min": round(min(rank_s), 4),
                      "videos_per_s_per_rank": [round(B * args.steps / s_, 4) for s_ in rank_s]},
        }
        # the two side measurements must never cost the run its JSON line
        try:
            line["roofline"], families = rooflines(dm, vq, cond, cf_cond, B, L, 16, device, args.codes, args.grid,
                                                   regime="trained_like" if args.trained_like else "flat")
        except Exception as e:                                   # noqa: BLE001
            line["roofline"], families = {"error": f"{type(e).__name__}: {e}"}, None
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args, dm, vq, L)
            except Exception as e:                               # noqa: BLE001
                line["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_extra:
            # regimes the dominant kernel could be sensitive to, outside the timed headline (same process, one warm + one timed
            # pass each): the reference-faithful zero conditioning (DiscreteDiffusion.forward zeroes BOTH embeddings,
            # discrete_diffusion.py:25, :49: the two guidance copies are then the same computation and sample() runs one of them --
            # zero_cond_guidance_copies says how many ran; the headline's condition differs from the unconditional one and runs
            # two), one lane instead of two, and weights of a trained-like magnitude (peaky softmax rows); redo_chunks =
            # overflow-redo events of the attention kernel per timed pass
            try:
                def timed(cond_):
                    nonlocal cond
                    keep, cond = cond, cond_
                    one_pass()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    one_pass()
                    torch.cuda.synchronize()
                    d = time.perf_counter() - t1
                    cond = keep
                    return round(B / d, 4), dm.attention_redo_events()
                extra = {"unit": "videos/s", "redo_chunks_headline": headline_redo}
                if families is not None:
                    extra["roofline_families"] = families
                extra["configs"] = other_configs(args, dm, vq, device)
                extra["zero_cond"], extra["redo_chunks_zero_cond"] = timed(torch.zeros_like(cond))
                extra["zero_cond_guidance_copies"] = 1 if getattr(dm, "_last_cfg_dedupe", False) else 2
                other = 1 if lanes_used > 1 else 2
                dm.sample_lanes = other                        # same tokens either way: the noise key is the global row
                extra["one_lane" if other == 1 else "two_lanes"], _ = timed(cond)
                dm.sample_lanes = args.lanes
                trained_like_weights(dm)
                extra["trained_like"], extra["redo_chunks_trained_like"] = timed(cond)
                try:
                    roof_t, _ = rooflines(dm, vq, cond, cf_cond, B, L, 16, device, args.codes, args.grid, reps=1, regime="trained_like")
                    extra["trained_like_attention_ms_per_launch"] = roof_t["ms_per_launch"]
                    extra["trained_like_attention_counters"] = roof_t["counters"]
                    extra["trained_like_attention_ms_by_block"] = roof_t["ms_by_block"]       # blocks 1..18 (block 0 is a half batch)
                except Exception:                                # noqa: BLE001
                    pass
                # the same trained-like weights with P = f16 hi only in every tile (attention_mode '11': data-independent cost, the
                # documented fast mode whose error is pinned by tests/test_gpu_fullsize.py::test_attention_mode_p11_contract)
                if dm.transformer.attention_mode is None:
                    dm.transformer.attention_mode = "11"
                    extra["trained_like_p11"], _ = timed(cond)
                    dm.transformer.attention_mode = None
                line["extra"] = extra
            except Exception as e:                               # noqa: BLE001
                line["extra"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
