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
# PMC passes.  Sampler kernels: bench.py ITSELF, headline shape only (--no-extra: the persistent kernels -- fused layer, logits -- launch
# the same grid for every batch size, so other shapes in the same pass would blur their rows).  Training kernels (C4's): tools/bench_train.py,
# launch by launch (GSDD_TRAIN_GRAPH=0: the PMC passes serialise dispatches anyway).
export GSDD_TRAIN_GRAPH=0
CTR="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
BENCH="$R/bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-extra"
TRAIN="$R/tools/bench_train.py 16 4"
rm -rf "$O"/sq_train "$O"/fetch_train "$O"/write_train
rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d "$O/sqb_flat" -- python3 $BENCH > "$O/sqb_flat.log" 2>&1
rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d "$O/sqb_trained" -- python3 $BENCH --trained-like > "$O/sqb_trained.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- python3 $BENCH > "$O/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/write" -- python3 $BENCH > "$O/write.log" 2>&1
rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d "$O/sq_train" -- python3 $TRAIN > "$O/sq_train.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/fetch_train" -- python3 $TRAIN > "$O/fetch_train.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/write_train" -- python3 $TRAIN > "$O/write_train.log" 2>&1
unset GSDD_TRAIN_GRAPH
SAMPLER="d3pm_attention_v4 d3pm_layer_h2 d3pm_logits d3pm_step gemm_kernel axial_attention"
TRAINK="attn_bwd d3pm_train_bwd rows_linear wgrad_kernel d3pm_attention_v4 ln_bwd gelu2"
python3 "$R/tools/make_sq_csv.py" flat="$O/sqb_flat" trained_like="$O/sqb_trained" -- $SAMPLER > "$O/sq_bench.csv"
python3 "$R/tools/make_sq_csv.py" c4_training_step="$O/sq_train" -- $TRAINK | tail -n +2 >> "$O/sq_bench.csv"
python3 "$R/tools/make_traffic_csv.py" "$O/fetch" "$O/write" $SAMPLER > "$O/traffic.csv"
python3 "$R/tools/make_traffic_csv.py" "$O/fetch_train" "$O/write_train" attn_bwd d3pm_train_bwd rows_linear wgrad_kernel ln_bwd | tail -n +2 >> "$O/traffic.csv"
# keep the transfer small: the raw traces stay on the box
find "$O" -name "*.csv" -size +2M -delete
