"""One launch of the split-precision kernel in PLAIN mode (fmode NONE), then a sync: diagnostic for a GPU-side abort."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cmf_amd import engine as E
E.scope(tangent="bf16x3").__enter__()
B, C, H, W, nc = 3, 64, 14, 14, 64
HW = H * W
x = torch.randn(B, C, H, W, nc, device="cuda")
res = torch.randn(B, C, H, W, nc, device="cuda") if "--res" in sys.argv else None
y = torch.zeros(B * C * HW * nc, device="cuda")
w = torch.nn.Parameter(torch.randn(C, C, 3, 3, device="cuda") / 24)
st = (C * HW * nc, HW * nc, nc)
print("launching", flush=True)
E.conv_tangent(x, 0, *st, w, 9, y, *st, B, C, C, H, W, nc, res_t=res)
torch.cuda.synchronize()
print("done", float(y.abs().max()), flush=True)
