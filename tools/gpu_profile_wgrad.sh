#!/bin/bash
# PMC passes on the split weight-gradient kernel alone (tools/bench_conv.py --wgrad): matrix-pipe occupancy, clock, fabric traffic.
#   usage (through gpurun): bash tools/gpu_profile_wgrad.sh <tag> [--lib cmf_amd/csrc/_obj/dbg_X.so]
tag=${1:-wgprof}; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
B="python3 tools/bench_conv.py --wgrad --iters 3 $*"
timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $out/stats -- $B > $out/stats.log 2>&1 || { tail -5 $out/stats.log; exit 1; }
timeout -k 10 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/pmc_sq -- $B > $out/pmc_sq.log 2>&1 || { tail -5 $out/pmc_sq.log; exit 1; }
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -- $B > $out/pmc_fetch.log 2>&1 || { tail -5 $out/pmc_fetch.log; exit 1; }
timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write -- $B > $out/pmc_write.log 2>&1 || { tail -5 $out/pmc_write.log; exit 1; }
python3 tools/rocprof_stats.py $out/stats $out/kernel_stats.csv "rocprofv3 --kernel-trace --stats -- $B  (MI355X; durations in us)" > /dev/null 2>&1
python3 tools/pmc_kernel.py --batch 128 --out $out/pmc_conv_wgrad.json --kernel "conv_wgrad3x3" --note "rocprofv3 --pmc <group> --kernel-trace -- $B (separate passes)" $out/pmc_sq $out/pmc_fetch $out/pmc_write > /dev/null 2>&1
rm -rf $out/stats $out/pmc_sq $out/pmc_fetch $out/pmc_write
grep conv_wgrad $out/kernel_stats.csv | cut -c1-200
cat $out/pmc_conv_wgrad.json
