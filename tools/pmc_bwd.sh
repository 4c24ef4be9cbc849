#!/bin/bash
# PMC passes over the attention backward variants (fused, fused without the dQ hand-over / product, the split pair)
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
O="$R/gpurun_out/pmc_bwd"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d "$O/a" -- python3 "$R/tools/bench_kernels.py" attnbwd > "$O/a.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT --output-format csv -d "$O/b" -- python3 "$R/tools/bench_kernels.py" attnbwd > "$O/b.log" 2>&1
for d in a b; do python3 "$R/tools/summarize_pmc.py" "$O/$d" attn_bwd > "$O/$d.summary.csv"; done
find "$O" -name "*.csv" -size +2M -delete
cat "$O/a.summary.csv" "$O/b.summary.csv"
