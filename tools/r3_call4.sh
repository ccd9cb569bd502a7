#!/bin/bash
out=gpurun_out/r3c4; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for lib in "" "--lib cmf_amd/csrc/_obj/dbg_HALFB.so"; do
  for cfg in "--B 256 --hw 28" "--B 256 --hw 28 --res 0" "--B 512 --hw 14"; do
    timeout -k 10 120 python tools/bench_conv.py $lib $cfg --fmode bits --iters 20 >> $out/variants.txt 2>&1 || { tail -5 $out/variants.txt; exit 1; }
  done
done
done
grep -v amdgpu.ids $out/variants.txt
