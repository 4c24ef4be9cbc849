#!/bin/bash
# SQ counters of the sampler's kernels from bench.py ITSELF (the program started directly after `--`), one pass per weight regime:
# the whole 100-step loop runs eagerly as one full batch (--no-graph: one lane, grid sizes = the roofline's), then rooflines() repeats
# the guided pass with HIP events.  tools/make_sq_csv.py turns the two counter_collection files into profiles/rN_pmc_sq_counters.csv.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
O="$R/gpurun_out/prof"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
CTR="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
rm -rf "$O/sqb_flat" "$O/sqb_trained"
rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d "$O/sqb_flat" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-extra > "$O/sqb_flat.log" 2>&1
rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d "$O/sqb_trained" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-extra --trained-like > "$O/sqb_trained.log" 2>&1
python3 "$R/tools/make_sq_csv.py" flat="$O/sqb_flat" trained_like="$O/sqb_trained" -- d3pm_attention_v4 d3pm_layer_h2 d3pm_logits d3pm_step gemm_kernel axial_attention > "$O/sq_bench.csv"
find "$O/sqb_flat" "$O/sqb_trained" -name "*.csv" -size +2M -delete
