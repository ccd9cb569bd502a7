#!/bin/bash
# round 3, GPU call 2: phase stamps of the dominant kernel, the new tests, the default bench line with its legs, C5 training variants
out=gpurun_out/r3c2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 120 python tools/read_stamps.py 1 > $out/stamps.txt 2>&1 || { tail -5 $out/stamps.txt; exit 1; }
for cfg in "--hw 28 --precision f32" "--hw 14 --precision f32" "--hw 28 --precision bf16x3" "--hw 14 --precision bf16x3"; do
  timeout -k 10 120 python tools/bench_conv.py --B 32 --nc 16 --res 1 --iters 20 $cfg >> $out/primal_shapes.txt 2>&1 || { tail -5 $out/primal_shapes.txt; exit 1; }
done
grep -v amdgpu.ids $out/primal_shapes.txt
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -s > $out/tests_round3.log 2>&1; rc=$?
tail -15 $out/tests_round3.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/exp_c5_train.py > $out/c5_train.txt 2>&1 || { tail -5 $out/c5_train.txt; exit 1; }
cat $out/c5_train.txt
timeout -k 10 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -5 $out/bench_default.err; exit 1; }
cat $out/bench_default.json
timeout -k 10 200 python bench.py --no-graph --no-legs --cpu-batch 0 --no-f32-exact > $out/bench_eager.json 2> $out/bench_eager.err || { tail -5 $out/bench_eager.err; exit 1; }
cut -c1-300 $out/bench_eager.json
