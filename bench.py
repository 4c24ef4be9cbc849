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
# GSDD_ATTN_P -> (kernel template argument, what the softmax probabilities are carried as into the P.V product)
P_MODES = {"a8": (8, "f16 hi (11 bits), + f16 lo (22 bits) in every (16-query, 32-key) tile that holds a probability above 2^-8 of "
                     "its row's running sum (measured logits error vs the fp32 oracle at full size: 8e-6 init weights, 1.5e-5 "
                     "trained-like; all-tiles hi+lo = GSDD_ATTN_P=22: 1.3e-6 / 8.5e-6)"),
           "a12": (12, "f16 hi, + lo where a probability exceeds 2^-12 of the row's running sum"),
           "22": (1, "f16 hi + lo (22 bits) everywhere"), "11": (0, "f16 hi only (11 bits)")}
HBM_PEAK_GBS = 8000.0


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


def attention_roofline(dm, B, L, H, device, K, reps=3):
    """Dominant kernel (self-attention, head dim 4) timed live with HIP events on its launch stream, in situ: after the timed
    region `reps` more denoiser passes run eagerly on the sampler's own stream and workspace, with an event pair around every
    full-batch attention launch (the ABI call; from block 1 on K and V arrive pre-split from the fused layer kernel, so this is
    the attention kernel alone), so the kernel sees the operands, cache
    state and clocks of the real loop (a back-to-back loop of attention launches alone clocks ~6 % lower).  Algorithmic
    FLOPs = 16*L^2 per (sample, head) (QK^T 2*4 + PV 2*4 per score).  `traffic` = HBM bytes per launch from the rocprofv3
    PMC passes committed under profiles/ (2*FETCH_SIZE + WRITE_SIZE, gfx950 correction)."""
    ws, (condv, Te, rep, B) = dm._last_ws, dm._last_run            # one lane of the sampler (B = its sub-batch)
    B2 = rep * B
    st = dm._stream
    tok = torch.randint(0, K, (B, L), device=device)
    tok[torch.rand((B, L), device=device) < 0.5] = K                  # half [MASK], like the middle of a chain
    t2 = torch.full((B2,), dm.num_timesteps // 2, dtype=torch.int64, device=device)
    events = []
    with torch.cuda.stream(st):
        dm.transformer.run(tok, condv, Te, t2, ws, rep=rep, stream=st)              # warm
        ws["attn_events"] = events
        for _ in range(reps):
            dm.transformer.run(tok, condv, Te, t2, ws, rep=rep, stream=st)
        ws.pop("attn_events")
    st.synchronize()
    ms = sum(e0.elapsed_ms(e1) for e0, e1 in events) / len(events)
    flops = 16.0 * L * L * H * B2
    tf = flops / (ms * 1e-3) / 1e12
    traffic = None
    if (B2, L, H) == (32, 4096, 16):
        traffic = (2 * 82043.7 + 33846.8) * 1024      # profiles/r2_pmc_traffic.csv, d3pm_attention_v4_kernel<384;8>
    return {"bound": "mfma", "kernel": f"d3pm_attention_v4_kernel<384, {P_MODES[os.environ.get('GSDD_ATTN_P', 'a8')][0]}>",
            "achieved": round(tf, 2),
            "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4),
            "traffic": traffic, "ms_per_launch": round(ms, 4), "launches_timed": len(events), "flops_per_launch": flops}


def cpu_baseline(args, dm, vq, L):
    """The CPU oracle (our restatement of the reference path, torch-CPU ops) on a bounded sample:
    one guided reverse step at B=1 (2 denoiser passes + posterior + Gumbel) and one decode, extrapolated
    to a whole video = diffusion_steps reverse steps + 1 decode."""
    from oracle import d3pm as od, vqvae as ov
    sd = {k: v.detach().cpu() for k, v in dm.state_dict().items()}
    vsd = {k: v.detach().cpu() for k, v in vq.state_dict().items()}
    cfg = dict(downsample=[1, 8, 8], n_res_layers=3)
    K = args.codes
    T = args.diffusion_steps
    g = torch.Generator().manual_seed(1)
    tok = torch.randint(0, K, (1, L), generator=g)
    tok[torch.rand(1, L, generator=g) < 0.5] = K
    cond = torch.randn(1, 1, 512, generator=g)
    t = torch.full((1,), T // 2, dtype=torch.long)
    with torch.no_grad():
        t0 = time.perf_counter()
        od.p_sample_step(tok, cond, torch.zeros_like(cond), t, sd, 2.0, None, 0)
        t_step = time.perf_counter() - t0
        codes = torch.randint(0, K, (1,) + tuple(args.grid), generator=g)
        t0 = time.perf_counter()
        ov.decode(codes, vsd, cfg)
        t_dec = time.perf_counter() - t0
    per_video = T * t_step + t_dec
    return {"value": 1.0 / per_video, "unit": "videos/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 guided reverse step at B=1 L={L} ({t_step:.2f} s) x{T} + 1 decode ({t_dec:.2f} s), "
                      "torch-CPU oracle"}


LAYER_ARITH = {
    "h2": "per-block GEMMs (proj, MLP, next q|k|v): operands as f16 hi + lo (22 bits, every product exact in the f32 accumulator; "
          "as accurate as an f32 GEMM: DESIGN.md section 4; GSDD_LAYER=x3p selects the bf16x3 kernel)",
    "x3p": "per-block GEMMs as 3-way bf16 splits too (GSDD_LAYER=x3p)",
    "x3": "per-block GEMMs as 3-way bf16 splits, split on the fly (GSDD_LAYER=x3)",
    "f32": "per-block GEMMs on v_mfma_f32_32x32x2_f32 (GSDD_LAYER=f32)",
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
    ap.add_argument("--lanes", type=int, default=1, help="concurrent sub-batches of the sampler (separate HIP streams)")
    ap.add_argument("--no-extra", action="store_true", help="skip the side regimes (trained-like weights, zero cond)")
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
    B = args.batch
    texts = ["synthetic"] * B
    g = torch.Generator().manual_seed(100 + rank)
    cond = torch.randn(B, 1, 512, generator=g).to(device)          # general conditioning (not the zeroed case)
    cf_cond = torch.zeros(B, 1, 512, device=device)
    dm.set_noise(1234, 0, row_offset=rank * B)
    dm.sample_lanes = args.lanes

    def one_pass():
        out = dm.sample(texts, None, cond, cf_cond, content_token=None, filter_ratio=0, use_graph=not args.no_graph)
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
    from gsdd_amd import ops as _ops
    _ops.d3pm_attention_redo_count(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        clips = one_pass()
    barrier()
    dt = time.perf_counter() - t0
    headline_redo = _ops.d3pm_attention_redo_count(reset=True) / max(args.steps, 1)
    assert tuple(clips.shape) == (B, 3, args.grid[0], args.grid[1] * 8, args.grid[2] * 8)
    assert torch.isfinite(clips).all()
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
                       "hipgraph": not args.no_graph,
                       "arith": "f32 results; QK^T, to_logits and VQ-VAE GEMM operands as error-free 3-way bf16 splits on the matrix "
                                "pipe (dropped terms < 2^-24); " + LAYER_ARITH[os.environ.get("GSDD_LAYER", "h2")] +
                                "; softmax P: " + P_MODES[os.environ.get("GSDD_ATTN_P", "a8")][1]},
            "ranks": {"seconds_max": round(max(rank_s), 4), "seconds_min": round(min(rank_s), 4),
                      "videos_per_s_per_rank": [round(B * args.steps / s_, 4) for s_ in rank_s]},
        }
        # the two side measurements must never cost the run its JSON line
        try:
            line["roofline"] = attention_roofline(dm, B, L, 16, device, args.codes)
        except Exception as e:                                   # noqa: BLE001
            line["roofline"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args, dm, vq, L)
            except Exception as e:                               # noqa: BLE001
                line["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_extra:
            # regimes the dominant kernel could be sensitive to, outside the timed headline (same process, one warm + one timed
            # pass each): the reference-faithful zero conditioning, and weights of a trained-like magnitude (peaky softmax rows);
            # redo_chunks = overflow-redo events of the attention kernel per timed pass
            try:
                from gsdd_amd import ops

                def timed(cond_):
                    nonlocal cond
                    keep, cond = cond, cond_
                    one_pass()
                    torch.cuda.synchronize()
                    ops.d3pm_attention_redo_count(reset=True)
                    t1 = time.perf_counter()
                    one_pass()
                    torch.cuda.synchronize()
                    d = time.perf_counter() - t1
                    cond = keep
                    return round(B / d, 4), ops.d3pm_attention_redo_count(reset=True)
                extra = {}
                extra["zero_cond"], extra["redo_chunks_zero_cond"] = timed(torch.zeros_like(cond))
                if args.lanes == 1 and B % 2 == 0 and B // 2 >= 4:
                    dm.sample_lanes = 2                        # two half-batches on two HIP streams (same tokens: the noise key is the
                    extra["two_lanes"], _ = timed(cond)        # global row); not the headline: per-launch timing wants one lane
                    dm.sample_lanes = 1
                trained_like_weights(dm)
                extra["trained_like"], extra["redo_chunks_trained_like"] = timed(cond)
                extra["unit"] = "videos/s"
                extra["redo_chunks_headline"] = headline_redo
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
