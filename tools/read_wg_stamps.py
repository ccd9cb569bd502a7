#!/usr/bin/env python3
"""One split weight-gradient launch with the CMF_DBG_WGSTAMP diagnostic library (tools/build_dbg.sh WGSTAMP): phase durations of
workgroup 0's MFMA wave 0 and producer wave 4 over the first 64 steps, in s_memtime cycles."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from cmf_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "cmf_amd/csrc/_obj", os.environ.get("CMF_DBG_LIB", "dbg_WGSTAMP.so"))
from cmf_amd import engine as E
B, H, nc, ch = 128, 28, 64, 64; HW = H * H
x = torch.randn(B, ch, H, H, nc, device="cuda"); prim = torch.randn(B, ch, H, H, device="cuda")
y = torch.randn(B, ch, H, H, nc, device="cuda"); dw = torch.zeros(ch, ch, 3, 3, device="cuda")
st, sl = (ch * HW * nc, 16, ch * nc), ch * 16          # slice-major hidden layout
def run():
    E.conv_tangent_wgrad(x, 0, *st, y, 0, *st, dw, 9, B, ch, ch, H, H, nc, fmode=E.F_RELU, f=prim, f_np=ch * HW, f_ci=HW, f_px=1, x_sl=sl, y_sl=sl)
for _ in range(3): run()
torch.cuda.synchronize()
buf = np.zeros((2, 64, 4), dtype=np.uint64)
lib = _lib.load(); lib.cmf_debug_read_wg_stamps.argtypes = [C.c_void_p]
assert lib.cmf_debug_read_wg_stamps(buf.ctypes.data) == 0
M, P = buf[0].astype(np.int64), buf[1].astype(np.int64)
t0 = M[0, 0]
print("step | MFMA wave: at barrier, wait, first unit, rest, | producer: at barrier, wait, vmcnt + split + park, fetch issue")
for g in range(8, 48):
    print(f"{g:3d} | {M[g,0]-t0:8d} {M[g,1]-M[g,0]:6d} {M[g,2]-M[g,1]:6d} {M[g,3]-M[g,2]:6d} | {P[g,0]-t0:8d} {P[g,1]-P[g,0]:6d} {P[g,2]-P[g,1]:6d} {P[g,3]-P[g,2]:6d}   step period {M[g+1,0]-M[g,0]:6d}")
