#!/bin/bash
# kernel stats of a command under both builds (ab/libgsdd_old.so, ab/libgsdd_new.so): tools/ab_prof.sh <substring> <program args...>
R="$(cd "$(dirname "$0")/.." && pwd)"
P=$R/gif-synthesis-with-discrete-diffusion_amd/libgsdd.so
PAT="$1"; shift
cd /tmp && export TMPDIR=/tmp
for v in old new; do
  cp $R/ab/libgsdd_$v.so $P
  rm -rf /tmp/abp_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp_$v -- "$@" > /tmp/abp_$v.log 2>&1
  f=$(find /tmp/abp_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python3 $R/tools/summarize_prof.py "$f" 40 | grep -E "$PAT"
done
cp $R/ab/libgsdd_new.so $P
