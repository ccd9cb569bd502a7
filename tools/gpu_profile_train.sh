#!/bin/bash
# rocprofv3 kernel statistics of the training step (bench.py --train --batch 64): which kernels the step consists of.
#   usage (through gpurun): bash tools/gpu_profile_train.sh <tag>
tag=${1:-trainprof}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -- python3 bench.py --train --batch 64 --steps 2 --warmup 1 > $out/stats.log 2>&1 || { tail -5 $out/stats.log; exit 1; }
python3 tools/rocprof_stats.py $out/stats $out/kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --train --batch 64 --steps 2 --warmup 1  (MI355X, C3 model, 64 samples; 3 training steps in the trace; durations in us; tools/gpu_profile_train.sh)" > /dev/null 2>&1
rm -rf $out/stats
python3 - $out/kernel_stats.csv <<'PY'
import csv, sys
rows = [r for r in csv.reader(open(sys.argv[1])) if len(r) == 5 and r[1].isdigit()]
tot = sum(float(r[2]) for r in rows)
print(f"total kernel time {tot/1e3:.1f} ms over 3 steps")
for r in rows[:40]:
    print(f"{float(r[2])/3e3:8.2f} ms/step {int(r[1])//3:5d}/step {float(r[3]):9.1f} us  {r[0][:110]}")
PY
