#!/bin/bash
# The rocprofv3 passes behind profiles/rN_*: kernel-trace stats and PMC counters are separate runs (the pool refuses mixed ones),
# each program is started directly after `--`.  Run from the repository root on the GPU box; results land in gpurun_out/prof/.
# Since round 4 every counter of the sampler's kernels comes from bench.py ITSELF (not from the microbenchmarks): the PMC passes run
# `bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline` -- the 100-step loop eagerly as one full batch, then rooflines(), then
# extra.configs (C5 per rank, C4, C2: so the training kernels are in the same passes).
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
O="$R/gpurun_out/prof"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O"/bench "$O"/train "$O"/vqtrain "$O"/nearest "$O"/sqb_flat "$O"/sqb_trained "$O"/fetch "$O"/write
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench" -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extra > "$O/bench.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/train" -- python3 "$R/tools/bench_train.py" 16 8 > "$O/train.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/vqtrain" -- python3 "$R/tools/bench_vqvae_train.py" 64 3 > "$O/vqtrain.log" 2>&1
export GSDD_NEAREST_M=32768
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/nearest" -- python3 "$R/tools/bench_kernels.py" nearest > "$O/nearest.log" 2>&1
unset GSDD_NEAREST_M
for d in bench train vqtrain nearest; do f=$(find "$O/$d" -name "*kernel_stats.csv" | head -1); python3 "$R/tools/summarize_prof.py" "$f" 18 > "$O/$d.summary.csv"; done
python3 "$R/tools/summarize_trace.py" "$O/bench" d3pm_ > "$O/bench.bygrid.csv"
python3 "$R/tools/summarize_trace.py" "$O/nearest" nearest_code code_norm > "$O/nearest.bygrid.csv"
if [ "$1" = "traces" ]; then
  find "$O" -name "*.csv" -size +2M -delete
  exit 0
fi
export GSDD_TRAIN_GRAPH=0          # the PMC passes serialise dispatches: the training step runs launch by launch there
CTR="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
BENCH="$R/bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d "$O/sqb_flat" -- python3 $BENCH > "$O/sqb_flat.log" 2>&1
rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d "$O/sqb_trained" -- python3 $BENCH --no-extra --trained-like > "$O/sqb_trained.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- python3 $BENCH > "$O/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/write" -- python3 $BENCH > "$O/write.log" 2>&1
unset GSDD_TRAIN_GRAPH
KERNELS="d3pm_attention_v4 d3pm_layer_h2 d3pm_logits d3pm_step gemm_kernel axial_attention attn_bwd d3pm_train_bwd rows_linear wgrad_kernel conv_wgrad nearest_code_mfma"
python3 "$R/tools/make_sq_csv.py" flat="$O/sqb_flat" trained_like="$O/sqb_trained" -- $KERNELS > "$O/sq_bench.csv"
python3 "$R/tools/make_traffic_csv.py" "$O/fetch" "$O/write" $KERNELS > "$O/traffic.csv"
# keep the transfer small: the raw traces stay on the box
find "$O" -name "*.csv" -size +2M -delete
