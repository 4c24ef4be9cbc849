#!/bin/bash
# same-box A/B of library builds on the headline bench: tools/ab_bench.sh <tag> ...   (ab/libgsdd_<tag>.so each; flat and trained-like weights)
cd "$(dirname "$0")/.."
for tag in "$@"; do
  for regime in "" "--trained-like"; do
    GSDD_LIB_PATH=$PWD/ab/libgsdd_$tag.so python3 bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline $regime 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$tag', '${regime:-flat}', 'videos/s', d['value'], 'attn ms', r['ms_per_launch'], 'by block', r['ms_by_block'])"
  done
done
