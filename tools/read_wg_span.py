#!/usr/bin/env python3
"""Per-workgroup start / end times of ONE launch of the dominant kernel (CMF_DBG_STAMP build: s_memrealtime at entry and at the exit of
each workgroup's MFMA wave 0): how evenly do the 256 persistent workgroups finish their equal shares of the items?
  tools/build_variant.sh conv_tangent_bf16x3 STAMP ; python tools/read_wg_span.py [B] [hw] [res]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from cmf_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "cmf_amd/csrc/_obj", os.environ.get("CMF_DBG_LIB", "dbg_STAMP.so"))
from cmf_amd import engine as E
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
H = int(sys.argv[2]) if len(sys.argv) > 2 else 14
res = int(sys.argv[3]) if len(sys.argv) > 3 else 1
nc, ch = 64, 64; HW = H * H
x = torch.randn(B, ch, H, H, nc, device="cuda"); prim = torch.randn(B, ch, H, H, device="cuda")
r = torch.randn(B, ch, H, H, nc, device="cuda") if res else None
y = torch.empty(B, ch, H, H, nc, device="cuda"); w = torch.nn.Parameter(torch.randn(ch, ch, 3, 3, device="cuda") / 24)
st, sl = (ch * HW * nc, 16, ch * nc), ch * 16
bits = E.relu_bits(prim)
def run():
    E.conv_tangent(x, 0, *st, w, 9, y, *st, B, ch, ch, H, H, nc, fmode=E.F_RELU_BITS, f=bits.data, f_np=bits.np_bytes, res_t=r, x_sl=sl, y_sl=sl)
lib = _lib.load(); lib.cmf_debug_read_wg_span.argtypes = [C.c_void_p]
for trial in range(3):
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); run(); e1.record(); torch.cuda.synchronize()
    buf = np.zeros((1024, 2), dtype=np.uint64)
    assert lib.cmf_debug_read_wg_span(buf.ctypes.data) == 0
    t = buf[:256].astype(np.int64)
    t0 = t[:, 0].min()
    start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0            # us (100 MHz counter)
    dur = end - start
    print(f"B={B} {H}x{H} res={res}: launch {e0.elapsed_time(e1)*1e3:.0f} us (HIP events, stamped build) | workgroup start {start.min():.1f} .. {start.max():.1f} us | "
          f"end min {end.min():.1f} median {np.median(end):.1f} max {end.max():.1f} us | busy min {dur.min():.1f} median {np.median(dur):.1f} max {dur.max():.1f} us")
    xcd = np.arange(256) & 7
    print("   per XCD: end median / max: " + "  ".join(f"{np.median(end[xcd == k]):.0f}/{end[xcd == k].max():.0f}" for k in range(8)))
    print(f"   idle CU-time before the last workgroup ends: {np.mean(end.max() - end):.1f} us on average = {100 * np.mean(end.max() - end) / end.max():.1f} % of the launch")
