#!/usr/bin/env python3
"""Copy the summaries tools/profile_round.sh left under gpurun_out/prof/ into profiles/ with the round's prefix and their header comments.
usage: collect_profiles.py r4"""
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
BENCH = ("python3 bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-extra   (the program started directly after `--`; tools/profile_round.sh); "
         "training kernels (attn_bwd_*; d3pm_train_bwd; rows_linear; wgrad; ln_bwd): the same passes over python3 tools/bench_train.py 16 4 with GSDD_TRAIN_GRAPH=0")
copy("traffic.csv", "pmc_traffic.csv", (
    "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- " + BENCH + "; tools/make_traffic_csv.py",
    "one row per (kernel; grid size); counter unit KB; HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 on gfx950 (FETCH_SIZE reports half the bytes of wide coalesced reads: MI355X_MICROARCH.md section HBM; other access widths are uncalibrated).",
    "the bench.py passes cover the headline loop (eager; one full batch of 2B = 32 rows: attention grid 8192; block 0 shared by the guidance copies: grid 4096) and rooflines(); the fused layer and logits kernels are persistent (grid 256 whatever the batch): only the headline shape runs in these passes."))
copy("sq_bench.csv", "pmc_sq_counters.csv", (
    "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -- " + BENCH + "; regime trained_like: the same with --trained-like; regime c4_training_step: tools/bench_train.py 16 4; tools/make_sq_csv.py",
    "means per dispatch; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over SQ_BUSY_CYCLES / 32 shader engines (= the dispatch's length in cycles); valu_issue_busy = 4 * SQ_ACTIVE_INST_VALU (quad-cycles) / 1024 over the same.",
    "bench.py reads `roofline.counters` and `extra.roofline_families[*].counters` from this file at run time (attention: the grid-8192 row = one full batch of 32 rows; 8.59e9 scores = 134.2 M wave-instruction slots of 64 scores)."))
print("profiles written for", tag)
