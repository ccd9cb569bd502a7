#!/bin/bash
# round 3, GPU call 2: phase stamps of the dominant kernel, the new tests, the default bench line with its legs, C5 training variants
out=gpurun_out/r3c2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for lib in "" "--lib cmf_amd/csrc/_obj/dbg_WSKIP.so" "--lib cmf_amd/csrc/_obj/dbg_NORING.so"; do
  for cfg in "--B 256 --hw 28" "--B 256 --hw 28 --res 0" "--B 512 --hw 14"; do
    timeout -k 10 120 python tools/bench_conv.py $lib $cfg --fmode bits >> $out/variants.txt 2>&1 || { tail -5 $out/variants.txt; exit 1; }
  done
done
grep -v amdgpu.ids $out/variants.txt
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -s > $out/tests_round3.log 2>&1; rc=$?
tail -15 $out/tests_round3.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/exp_c5_train.py > $out/c5_train.txt 2>&1 || { tail -5 $out/c5_train.txt; exit 1; }
cat $out/c5_train.txt
timeout -k 10 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -5 $out/bench_default.err; exit 1; }
cat $out/bench_default.json
timeout -k 10 200 python bench.py --no-graph --no-legs --cpu-batch 0 --no-f32-exact > $out/bench_eager.json 2> $out/bench_eager.err || { tail -5 $out/bench_eager.err; exit 1; }
cut -c1-300 $out/bench_eager.json
