#!/bin/bash
# SQ counters of one tools/bench_kernels.py case: tools/pmc_one.sh <case> <kernel substring>
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
O="$R/gpurun_out/pmc_$1"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d "$O/a" -- python3 "$R/tools/bench_kernels.py" "$1" > "$O/a.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$O/b" -- python3 "$R/tools/bench_kernels.py" "$1" > "$O/b.log" 2>&1
for d in a b; do python3 "$R/tools/summarize_pmc.py" "$O/$d" "$2"; done
find "$O" -name "*.csv" -size +2M -delete
