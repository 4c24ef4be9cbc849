#!/bin/bash
# The rocprofv3 passes behind profiles/rN_*: kernel-trace stats and PMC counters are separate runs (the pool refuses mixed ones),
# each program is started directly after `--`.  Run from the repository root on the GPU box; results land in gpurun_out/prof/.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
O="$R/gpurun_out/prof"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench" -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extra > "$O/bench.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/train" -- python3 "$R/tools/bench_train.py" 16 > "$O/train.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/vqtrain" -- python3 "$R/tools/bench_vqvae_train.py" 64 3 > "$O/vqtrain.log" 2>&1
export GSDD_NEAREST_M=32768
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/nearest" -- python3 "$R/tools/bench_kernels.py" nearest > "$O/nearest.log" 2>&1
unset GSDD_NEAREST_M
if [ "$1" = "traces" ]; then
  python3 "$R/tools/summarize_trace.py" "$O/bench" d3pm_ > "$O/bench.bygrid.csv"
  python3 "$R/tools/summarize_trace.py" "$O/nearest" nearest_code code_norm > "$O/nearest.bygrid.csv"
  for d in bench train vqtrain nearest; do f=$(find "$O/$d" -name "*kernel_stats.csv" | head -1); python3 "$R/tools/summarize_prof.py" "$f" 16 > "$O/$d.summary.csv"; done
  find "$O" -name "*.csv" -size +2M -delete
  exit 0
fi
export GSDD_BENCH_SCALES=0.05
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d "$O/sq_flat" -- python3 "$R/tools/bench_kernels.py" attn > "$O/sq_flat.log" 2>&1
export GSDD_BENCH_SCALES=1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d "$O/sq_x1" -- python3 "$R/tools/bench_kernels.py" attn > "$O/sq_x1.log" 2>&1
export GSDD_BENCH_SCALES=0.05 GSDD_BENCH_PMODES=a8,22 GSDD_BENCH_BWD_VARIANTS=fused
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- python3 "$R/tools/bench_kernels.py" attn step attnbwd nearest > "$O/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/write" -- python3 "$R/tools/bench_kernels.py" attn step attnbwd nearest > "$O/write.log" 2>&1
unset GSDD_BENCH_SCALES GSDD_BENCH_PMODES GSDD_BENCH_BWD_VARIANTS
for d in sq_flat sq_x1; do python3 "$R/tools/summarize_pmc.py" "$O/$d" d3pm_attention_v4 > "$O/$d.summary.csv"; done
python3 "$R/tools/make_traffic_csv.py" "$O/fetch" "$O/write" d3pm_attention_v4 d3pm_step attn_bwd nearest_code > "$O/traffic.csv"
for d in bench train vqtrain nearest; do f=$(find "$O/$d" -name "*kernel_stats.csv" | head -1); python3 "$R/tools/summarize_prof.py" "$f" 16 > "$O/$d.summary.csv"; done
python3 "$R/tools/summarize_trace.py" "$O/bench" d3pm_ > "$O/bench.bygrid.csv"
python3 "$R/tools/summarize_trace.py" "$O/nearest" nearest_code code_norm > "$O/nearest.bygrid.csv"
# keep the transfer small: the raw traces stay on the box
find "$O" -name "*.csv" -size +2M -delete
