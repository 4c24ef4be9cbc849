#!/usr/bin/env python3
"""Copy the summaries tools/profile_round.sh left under gpurun_out/prof/ into profiles/ with the round's prefix and their header comments.
usage: collect_profiles.py r3"""
import csv
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(R, "gpurun_out", "prof")
tag = sys.argv[1]


def copy(src, dst, header=()):
    with open(os.path.join(P, src)) as f:
        body = f.read()
    with open(os.path.join(R, "profiles", f"{tag}_{dst}"), "w") as f:
        for h in header:
            f.write("# " + h + "\n")
        f.write(body)


copy("bench.summary.csv", "bench_kernel_stats.csv")
copy("bench.bygrid.csv", "bench_kernel_stats_by_grid.csv")
copy("nearest.bygrid.csv", "nearest_code_kernel_stats.csv")
copy("train.summary.csv", "d3pm_train_kernel_stats.csv")
if os.path.exists(os.path.join(P, "vqtrain.summary.csv")):
    copy("vqtrain.summary.csv", "vqvae_train_kernel_stats.csv")
copy("traffic.csv", "pmc_traffic.csv", (
    "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/bench_kernels.py attn step attnbwd nearest   (tools/profile_round.sh; tools/make_traffic_csv.py)",
    "one row per (kernel; grid size); counter unit KB; HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 on gfx950 (FETCH_SIZE reports half the bytes of wide coalesced reads: MI355X_MICROARCH.md section HBM).",
    "attention (sampler): grid 8192 = 2B = 32 rows; L = 4096; 16 heads; flat rows (algorithmic: q 33.5 + K/V images 134.2 + tile norms and key sums 2.6 read; 33.5 MB written); grid 4096 = the training forward at B = 16;",
    "d3pm_step: B = 16; L = 4096; K = 4096 (algorithmic 2147.5 MB read; 0.5 MB written); attention backward (the shipped fused variant alone): B = 16; L = 4096; 16 heads (algorithmic: q; k; v; o; dO in; dq; dk; dv out = 134 MB);",
    "nearest_code_*: grid 1024 = both microbenchmark sizes of the matrix-core kernel (32768 and 262144 latents x 4096 codes x 128; the codebook is split over more workgroups for the smaller one)."))

# SQ counters of the attention variants on flat and unit-scale rows, with vector instructions per 64 scores
scores64 = 32 * 16 * 4096 * 4096 / 64.0
with open(os.path.join(R, "profiles", f"{tag}_pmc_sq_counters.csv"), "w") as out:
    out.write("# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES "
              "-- python3 tools/bench_kernels.py attn   (GSDD_BENCH_SCALES=0.05 and =1; tools/profile_round.sh)\n")
    out.write("# 2B = 32; L = 4096; 16 heads: 8.59e9 scores per dispatch = 134.2 M wave-instruction slots of 64 scores; template argument: 1 = hi + lo "
              "everywhere; 0 = hi only; 8 / 12 = adaptive (bound first; then measured)\n")
    out.write("rows,kernel,counter,mean_per_dispatch,dispatches,per_64_scores\n")
    for src, label in (("sq_flat.summary.csv", "flat (q;k x0.05)"), ("sq_x1.summary.csv", "unit scale (q;k x1)")):
        for r in csv.DictReader(open(os.path.join(P, src))):
            per = f"{float(r['mean_per_dispatch']) / scores64:.2f}" if r["counter"] == "SQ_INSTS_VALU" else ""
            out.write(f"{label},{r['kernel']},{r['counter']},{r['mean_per_dispatch']},{r['dispatches']},{per}\n")
print("profiles written for", tag)
