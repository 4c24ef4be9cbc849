#!/bin/bash
# same-box A/B of two builds of the library: tools/ab.sh <command...>  (ab/libgsdd_old.so vs ab/libgsdd_new.so, each run twice, interleaved)
P=gif-synthesis-with-discrete-diffusion_amd/libgsdd.so
for r in 1 2; do
  for v in old new; do
    cp ab/libgsdd_$v.so $P
    echo "== $v (run $r)"
    "$@" 2>&1 | grep -v amdgpu.ids
  done
done
cp ab/libgsdd_new.so $P
