#!/bin/bash
# Round profile of the default bench command on the GPU box: rocprofv3 kernel stats, then the PMC passes (SQ group, FETCH_SIZE,
# WRITE_SIZE: separate runs, MI355X_MICROARCH.md), then the per-kernel HIP-event table.  Output under gpurun_out/<tag>/; the
# summaries are made from it with tools/rocprof_stats.py and tools/pmc_kernel.py and copied to profiles/.
#   usage (through gpurun): bash tools/gpu_profile.sh <tag>
# rocprofv3 gets `python3 bench.py ...` itself after `--` (no env / bash -c hop: the profiler's preload initialises the GPU).
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
B2="python3 bench.py --steps 2 --warmup 1 --cpu-batch 0 --no-kernel-timer --no-f32-exact --no-legs"
B1="python3 bench.py --steps 1 --warmup 0 --cpu-batch 0 --no-kernel-timer --no-f32-exact --no-legs"
timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $out/stats -- $B2 > $out/stats.log 2>&1 || { tail -5 $out/stats.log; exit 1; }
timeout -k 10 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/pmc_sq -- $B1 > $out/pmc_sq.log 2>&1 || { tail -5 $out/pmc_sq.log; exit 1; }
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -- $B1 > $out/pmc_fetch.log 2>&1 || { tail -5 $out/pmc_fetch.log; exit 1; }
timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write -- $B1 > $out/pmc_write.log 2>&1 || { tail -5 $out/pmc_write.log; exit 1; }
# summaries on the box (the raw rocpd databases are ~20 MB per pass: too big to travel back), then drop the raw output
python3 tools/rocprof_stats.py $out/stats $out/kernel_stats.csv > /dev/null 2>&1
NOTE="rocprofv3 --pmc <group> --kernel-trace -- $B1 (separate passes: SQ group / FETCH_SIZE / WRITE_SIZE)"
python3 tools/pmc_kernel.py --out $out/pmc_gram_cholesky.json --kernel gram_chol --note "$NOTE" $out/pmc_sq $out/pmc_fetch $out/pmc_write > /dev/null 2>&1
python3 tools/pmc_kernel.py --out $out/pmc_acl_tangent.json --kernel acl_tangent --note "$NOTE" $out/pmc_sq $out/pmc_fetch $out/pmc_write > /dev/null 2>&1
python3 tools/pmc_kernel.py --source cmf_amd/csrc/conv_tangent_bf16x3.hip --out $out/pmc_conv_tangent_bf16x3.json --kernel "conv_tangent_bf16x3_kernel<4, 7, 3, false, false, false>" --note "$NOTE" $out/pmc_sq $out/pmc_fetch $out/pmc_write > /dev/null 2>&1
# round 5: the checkerboard-output form (the last hidden conv of the seven checkerboard couplers: 7 launches per step)
python3 tools/pmc_kernel.py --source cmf_amd/csrc/conv_tangent_bf16x3.hip --out $out/pmc_conv_tangent_bf16x3_live.json --kernel "conv_tangent_bf16x3_kernel<4, 7, 3, false, false, true>" --note "$NOTE" $out/pmc_sq $out/pmc_fetch $out/pmc_write > /dev/null 2>&1
python3 tools/pmc_kernel.py --source cmf_amd/csrc/conv_tangent_bf16x3.hip --out $out/pmc_conv_tangent_f16x3.json --kernel "conv_tangent_bf16x3_kernel<4, 7, 2, true" --note "$NOTE" $out/pmc_sq $out/pmc_fetch $out/pmc_write > /dev/null 2>&1
python3 tools/pmc_kernel.py --source cmf_amd/csrc/conv_tangent.hip --out $out/pmc_conv_tangent_thin.json --kernel "conv_tangent_thin_kernel" --note "$NOTE" $out/pmc_sq $out/pmc_fetch $out/pmc_write > /dev/null 2>&1
rm -rf $out/stats $out/pmc_sq $out/pmc_fetch $out/pmc_write
timeout -k 10 120 python3 tools/kernel_table.py > $out/kernel_table.txt 2>&1
tail -14 $out/kernel_table.txt
head -30 $out/kernel_stats.csv
cat $out/pmc_gram_cholesky.json $out/pmc_acl_tangent.json
du -sh $out
