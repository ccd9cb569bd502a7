#!/usr/bin/env python3
"""Run one bf16x3 tangent-conv launch with the CMF_DBG_STAMP diagnostic library and print phase durations (cycles)."""
import ctypes as C, os, sys, subprocess
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from cmf_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "cmf_amd/csrc/_obj", os.environ.get("CMF_DBG_LIB", "dbg_STAMP.so"))
from cmf_amd import engine as E
res = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B, H, nc, ch = 128, 28, 64, 64; HW = H * H
x = torch.randn(B, ch, H, H, nc, device="cuda"); prim = torch.randn(B, ch, H, H, device="cuda")
r = torch.randn(B, ch, H, H, nc, device="cuda") if res else None
y = torch.empty(B, ch, H, H, nc, device="cuda"); w = torch.nn.Parameter(torch.randn(ch, ch, 3, 3, device="cuda") / 24)
st, sl = (ch*HW*nc, 16, ch*nc), ch*16          # slice-major hidden layout
def run():
    E.conv_tangent(x, 0, *st, w, 9, y, *st, B, ch, ch, H, H, nc, fmode=E.F_RELU, f=prim, f_np=ch*HW, f_ci=HW, f_px=1, res_t=r, x_sl=sl, y_sl=sl)
for _ in range(3): run()
torch.cuda.synchronize()
buf = np.zeros((3, 64, 4), dtype=np.uint64)
lib = _lib.load(); lib.cmf_debug_read_stamps.argtypes = [C.c_void_p]
assert lib.cmf_debug_read_stamps(buf.ctypes.data) == 0
M, L = buf[0].astype(np.int64), buf[1].astype(np.int64)
t0 = M[0, 0]
print("MFMA wave: g  start  compute  barrier_wait   (cycles @100MHz*? raw s_memtime units)")
for g in range(24):
    print(f"  g={g:2d} start={M[g,0]-t0:8d} compute={M[g,1]-M[g,0]:6d} barrier={M[g,2]-M[g,1]:6d} next_ctx={max(M[g,3]-M[g,0],0):6d}")
print("loader wave: g  start  prefetch_issue  wait_loads  commit  (then barrier)")
for g in range(24):
    print(f"  g={g:2d} start={L[g,0]-t0:8d} issue={L[g,1]-L[g,0]:6d} wait={L[g,2]-L[g,1]:6d} commit={L[g,3]-L[g,2]:6d} next_start-gap={L[g+1,0]-L[g,3]:6d}")

I = buf[2].astype(np.int64)
print("item: decode_done->acc_init_issued, first chunk start - that, loop_end->stores_issued, next item decode gap")
for it in range(4):
    print(f"  item={it} t_decode={I[it,0]-t0:8d} init_issue={I[it,1]-I[it,0]:6d} first_mfma_chunk_start={M[it*8,0]-I[it,1]:6d} loop={I[it,2]-I[it,1]:7d} stores={I[it,3]-I[it,2]:6d} to_next_decode={I[it+1,0]-I[it,3]:6d}")
