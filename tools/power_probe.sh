#!/bin/bash
# Board power and shader clock (rocm-smi, once a second) while a kernel loop runs: idle, the split forward conv, the split weight
# gradient, the pure-MFMA micro-benchmark.   usage (through gpurun): bash tools/power_probe.sh <tag>
tag=${1:-power}; out=gpurun_out/$tag; mkdir -p $out
probe() {  # $1 = label, rest = command
  label=$1; shift
  "$@" > $out/$label.log 2>&1 &
  pid=$!
  sleep 6                                             # imports + warm-up
  for i in 1 2 3 4 5; do
    /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level|Temperature \(Sensor junction\)" | tr '\n' ';' | sed "s/^/$label: /"; echo
    sleep 1
  done
  wait $pid
}
/opt/rocm/bin/rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power|sclk" | sed 's/^/idle: /'
probe fwd_conv   python tools/bench_conv.py --B 512 --iters 4000
probe wgrad      python tools/bench_conv.py --wgrad --B 128 --iters 12000
probe mfma_only  python -c "
import ctypes as C, torch
lib = C.CDLL('cmf_amd/csrc/_obj/ubench_mfmapower.so')
lib.run_mfmapower.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
src = torch.randn(4096 * 8, device='cuda').to(torch.bfloat16); out = torch.zeros(1024, device='cuda'); cyc = torch.zeros(1, dtype=torch.int64, device='cuda')
for _ in range(600):                                 # ~14 s of back-to-back MFMAs on live random operands, one wave per SIMD
    lib.run_mfmapower(src.data_ptr(), out.data_ptr(), cyc.data_ptr(), 40000, 256, 1, None)
torch.cuda.synchronize()
"
