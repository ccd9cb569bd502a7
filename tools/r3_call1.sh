#!/bin/bash
# round 3, GPU call 1: baseline on this box, MFMA-order variant, Infinity-Cache chain experiment, LDS / wait counters
out=gpurun_out/r3c1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 200 python bench.py --cpu-batch 0 --no-f32-exact > $out/bench_c3.json 2> $out/bench_c3.err || { tail -5 $out/bench_c3.err; exit 1; }
cat $out/bench_c3.json | cut -c1-400
for lib in "" "--lib cmf_amd/csrc/_obj/dbg_MFMAORD.so"; do
  for cfg in "--B 256 --hw 28" "--B 512 --hw 14" "--B 256 --hw 28 --res 0"; do
    timeout -k 10 120 python tools/bench_conv.py $lib $cfg --fmode bits >> $out/bench_conv.txt 2>&1 || { tail -5 $out/bench_conv.txt; exit 1; }
  done
done
cat $out/bench_conv.txt
timeout -k 10 200 python tools/exp_chain.py --B 512 --hw 28 --groups 512,64,32,16,8,4 > $out/chain28.txt 2>&1 || { tail -5 $out/chain28.txt; exit 1; }
cat $out/chain28.txt
timeout -k 10 200 python tools/exp_chain.py --B 512 --hw 14 --groups 512,128,64,32,16 > $out/chain14.txt 2>&1 || { tail -5 $out/chain14.txt; exit 1; }
cat $out/chain14.txt
P="python3 tools/bench_conv.py --B 256 --hw 28 --fmode bits --iters 2"
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $out/pmc_a -- $P > $out/pmc_a.log 2>&1 || { tail -5 $out/pmc_a.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/pmc_b -- $P > $out/pmc_b.log 2>&1 || { tail -5 $out/pmc_b.log; exit 1; }
python3 tools/pmc_kernel.py --out $out/pmc_conv_lds.json --kernel "conv_tangent_bf16x3_kernel" --batch 256 --note "rocprofv3 --pmc (2 passes) --kernel-trace -- $P" $out/pmc_a $out/pmc_b > /dev/null 2>&1
rm -rf $out/pmc_a $out/pmc_b
cat $out/pmc_conv_lds.json
