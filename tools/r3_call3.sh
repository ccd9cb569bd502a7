#!/bin/bash
# round 3, GPU call 3: the whole GPU suite on the current kernels, C5 training profile, a 256-sample C5 training step
out=gpurun_out/r3c3; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py -q -m gpu > $out/gpu_tests.log 2>&1; rc=$?
tail -8 $out/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/exp_c5_train.py --modes 32 > $out/c5_train.txt 2>&1 || { tail -5 $out/c5_train.txt; exit 1; }
timeout -k 10 300 python tools/exp_c5_train.py --modes 32 --B 256 --steps 2 >> $out/c5_train.txt 2>&1 || { tail -5 $out/c5_train.txt; exit 1; }
grep "C5 train" $out/c5_train.txt
CMD="python3 bench.py --config c5 --train --steps 2 --warmup 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/c5train_stats -- $CMD > $out/c5train_stats.log 2>&1 || { tail -5 $out/c5train_stats.log; exit 1; }
python3 tools/rocprof_stats.py $out/c5train_stats $out/c5_train_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- $CMD (MI355X; 3 steps in the trace; durations in us)" > /dev/null 2>&1
rm -rf $out/c5train_stats
head -40 $out/c5_train_kernel_stats.csv
