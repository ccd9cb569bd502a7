#!/bin/bash
# Everything under profiles/r0N_* that comes from the GPU box, in one gpurun call (about 14 GPU-minutes):
#   /usr/local/graft/bin/gpurun --timeout 1190 -- 'bash tools/refresh_profiles.sh <tag>'
# then, here:  python tools/install_profiles.py gpurun_out/<tag> r03      (copies the summaries into profiles/)
# Steps: pytest -m gpu, every bench line, the default command under rocprofv3 (stats + three PMC passes), the per-kernel
# HIP-event tables of an evaluation and of a training step, rocprofv3 stats of a training step, PMC of the weight-gradient kernel.
tag=${1:-refresh}
out=gpurun_out/$tag
mkdir -p $out
if [ -z "$SKIP_TESTS" ]; then                         # SKIP_TESTS=1: the suite ran green in its own call (it takes ~5 min by now)
  timeout -k 10 700 python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1; rc=$?
  tail -3 $out/gpu_tests.log
  [ $rc -ne 0 ] && exit $rc
fi
# the PMC passes first: bench.py's roofline.traffic cites profiles/pmc_conv_tangent_bf16x3.json, stamped with the kernel source's hash
bash tools/gpu_profile.sh $tag/eval > $out/eval_profile.log 2>&1 || { tail -5 $out/eval_profile.log; exit 1; }
cp $out/eval/pmc_conv_tangent_bf16x3.json profiles/pmc_conv_tangent_bf16x3.json
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $out/bench_c3.json 2> $out/bench_c3.err || exit 1   # the default line as the driver runs it: headline + every leg
timeout -k 10 200 python bench.py --config c5 --batch 256 --steps 5 --cpu-batch 0 > $out/bench_c5_b256.json 2>/dev/null || exit 1   # configs[4] on ONE GPU
for c in c1 c2a c2b c5; do timeout -k 10 200 python bench.py --config $c > $out/bench_$c.json 2>/dev/null || exit 1; done
timeout -k 10 200 python bench.py --config c5 --hutchinson > $out/bench_c5_hutch.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --train --batch 64 > $out/bench_train.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --config c5 --train > $out/bench_c5_train.json 2>/dev/null || exit 1
timeout -k 10 300 python tools/bench_train.py --batch 64 --steps 3 > $out/train_step_b64.txt 2>&1 || exit 1
bash tools/gpu_profile_train.sh $tag/train > $out/train_profile.log 2>&1 || { tail -5 $out/train_profile.log; exit 1; }
bash tools/gpu_profile_wgrad.sh $tag/wgrad > $out/wgrad_profile.log 2>&1 || { tail -5 $out/wgrad_profile.log; exit 1; }
# C5 training step (low-rank Hutchinson backward): per-kernel totals, and one 256-sample step on the single GPU
CMD5="python3 bench.py --config c5 --train --steps 2 --warmup 1"
mkdir -p $out/c5train
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/c5train/stats -- $CMD5 > $out/c5train/stats.log 2>&1 || { tail -5 $out/c5train/stats.log; exit 1; }
python3 tools/rocprof_stats.py $out/c5train/stats $out/c5train/kernel_stats.csv "rocprofv3 --kernel-trace --stats -- $CMD5  (MI355X, CIFAR d=128 model, 32 samples, train-mode Hutchinson S=4 + CG, low-rank backward; 3 steps in the trace; durations in us)" > /dev/null 2>&1
rm -rf $out/c5train/stats
timeout -k 10 300 python tools/exp_c5_train.py --modes 16,32,full > $out/c5_train_variants.txt 2>&1 || { tail -5 $out/c5_train_variants.txt; exit 1; }
timeout -k 10 300 python tools/exp_c5_train.py --modes 32 --B 256 --steps 2 >> $out/c5_train_variants.txt 2>&1 || { tail -5 $out/c5_train_variants.txt; exit 1; }
grep "C5 train" $out/c5_train_variants.txt
timeout -k 10 200 python tools/ubench/mfmapower.py > $out/mfma_sustained.txt 2>&1
timeout -k 10 200 python tools/stage_table.py c3 c5 > $out/stage_table.txt 2>&1
timeout -k 10 300 python tools/bench_train.py --config c5 --steps 3 > $out/c5_train_step.txt 2>&1
timeout -k 10 200 python tests/dev/f16x3_item_probe.py > $out/f16x3_item_sizes.txt 2>&1
du -sh $out
