#!/usr/bin/env python3
"""One fp16-split PRIMAL conv launch with ONE work item per workgroup under the CMF_DBG_STAMP library: where the ~17 us go
(tools/build_variant.sh conv_tangent_bf16x3 STAMP; workgroup 0's MFMA wave 0 and loader wave 4).   python tools/read_stamps_f16.py [item_channels]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from cmf_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "cmf_amd/csrc/_obj", os.environ.get("CMF_DBG_LIB", "dbg_STAMP.so"))
from cmf_amd import engine as E
item = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = W = 16; B = 32; Cc = 64; HW = H * W; G = B // 16
xg = torch.randn(G * Cc * HW * 16, device="cuda"); yg = torch.empty_like(xg)
wd = torch.nn.Parameter(torch.randn(Cc, Cc, 3, 3, device="cuda") / 24); bias = torch.randn(Cc, device="cuda")
pn = (Cc * HW * 16, HW * 16, 16)
rng = torch.zeros(2, device="cuda"); E.absmax(xg, rng[0:1])
m = E.BitMask(B, HW, Cc, "cuda")
def run():
    E.conv_tangent(xg, 0, *pn, wd, 9, yg, *pn, G, Cc, Cc, H, W, 16, fmode=E.F_SELF_RELU, bias=bias, precision="f16x3", mask_out=m.data,
                   mask_np=m.np_bytes, amax_in=rng[0:1], amax_out=rng[1:2], item_channels=item)
for _ in range(5): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print(f"launch (HIP events, back to back): {e0.elapsed_time(e1) * 20:.1f} us")
buf = np.zeros((3, 64, 4), dtype=np.uint64)
lib = _lib.load(); lib.cmf_debug_read_stamps.argtypes = [C.c_void_p]
assert lib.cmf_debug_read_stamps(buf.ctypes.data) == 0
M, L, I = (buf[i].astype(np.int64) for i in range(3))
t0 = I[8, 0]                                                       # kernel entry of workgroup 0
print(f"kernel entry -> exit of MFMA wave 0: {I[8,1]-t0} ; loader wave 4 entry {I[9,0]-t0}, exit {I[9,1]-t0}   (s_memtime units)")
print("MFMA wave : g  chunk_start  compute  barrier_wait")
for g in range(8):
    print(f"  g={g} start={M[g,0]-t0:7d} compute={M[g,1]-M[g,0]:6d} barrier={M[g,2]-M[g,1]:6d}")
print("loader wave: g  start  prefetch_issue  wait_loads  commit")
for g in range(8):
    print(f"  g={g} start={L[g,0]-t0:7d} issue={L[g,1]-L[g,0]:6d} wait={L[g,2]-L[g,1]:6d} commit={L[g,3]-L[g,2]:6d}")
print(f"loader prologue: first prefetches issued {I[10,0]-t0}, first data landed {I[10,1]-t0}, stage 0 committed {I[10,2]-t0}")
print(f"MFMA epilogue: last chunk end {M[7,1]-t0}, stores / masks issued {I[11,0]-t0}, amax reduced {I[11,1]-t0}")
