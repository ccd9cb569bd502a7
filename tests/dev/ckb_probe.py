"""Which compact pixels of the checkerboard-output launch differ from the full launch (dev probe)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cmf_amd import engine as E
H, W, live, with_res = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
gen = torch.Generator().manual_seed(1)
B, C, nc, HW = 2, 64, 32, H * W
S = nc // 16
x = torch.randn(B, C, H, W, nc, generator=gen)
prim = torch.randn(B, C, H, W, generator=gen)
res = torch.randn(B, C, H, W, nc, generator=gen)
w = torch.randn(C, C, 3, 3, generator=gen) / 24
to_dev = lambda t: t.reshape(B, C, -1, S, 16).permute(0, 2, 3, 1, 4).contiguous().cuda()
st, sl = (C * HW * nc, 16, C * nc), C * 16
wd = torch.nn.Parameter(w.cuda())
xd, rd = to_dev(x), to_dev(res) if with_res else None
fk = dict(fmode=E.F_RELU, f=prim.cuda(), f_np=C * HW, f_ci=HW, f_px=1)
full = torch.empty(B, HW, S, C, 16, device="cuda")
E.conv_tangent(xd, 0, *st, wd, 9, full, *st, B, C, C, H, W, nc, res_t=rd, x_sl=sl, y_sl=sl, precision="bf16x3", **fk)
comp = torch.full((B, HW // 2, S, C, 16), float("nan"), device="cuda")
E.conv_tangent(xd, 0, *st, wd, 9, comp, st[0] // 2, st[1], st[2], B, C, C, H, W, nc, res_t=rd, x_sl=sl, y_sl=sl, precision="bf16x3", live=live, res_np=st[0], **fk)
ii, jj = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
sel = ((ii + jj) % 2 == live - 1).reshape(-1)
ref = full[:, sel.cuda()]
ok = (comp == ref).all(dim=(0, 2, 3, 4)).cpu()
rows = ii.reshape(-1)[sel]; cols = jj.reshape(-1)[sel]
print("H W live res", H, W, live, with_res, "bad pixels:", int((~ok).sum()), "of", ok.numel())
for m in (~ok).nonzero().flatten().tolist()[:40]:
    d = (comp[:, m] - ref[:, m]).abs().max().item()
    # does it match some other full pixel?
    hit = [(int(p // W), int(p % W)) for p in range(HW) if torch.equal(comp[:, m], full[:, p])]
    print("compact", m, "= pixel", (int(rows[m]), int(cols[m])), "maxdiff", d, "equals full pixel", hit, "nan" if torch.isnan(comp[:, m]).any() else "")
chan_ok = (comp == ref).all(dim=(0, 1, 2, 4)).cpu()
print("bad channels:", (~chan_ok).nonzero().flatten().tolist())
