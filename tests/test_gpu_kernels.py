"""Unit parity of the individual HIP kernels against plain PyTorch fp32 on random data (``-m gpu``)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("cin,cout,H,W,taps,nc", [
    (64, 64, 14, 14, 9, 64), (64, 64, 28, 28, 9, 32), (2, 64, 14, 14, 9, 16), (64, 4, 14, 14, 1, 64),
    (1, 64, 28, 28, 9, 32), (1, 64, 5, 14, 9, 16), (2, 40, 6, 14, 9, 16), (64, 64, 16, 16, 9, 16), (64, 64, 32, 32, 9, 32), (32, 64, 14, 14, 9, 16), (64, 32, 14, 28, 9, 32), (32, 32, 16, 8, 9, 16),
    (128, 64, 14, 14, 9, 16), (96, 128, 6, 14, 9, 16), (3, 24, 9, 11, 9, 16), (8, 64, 14, 14, 9, 16), (16, 40, 5, 28, 9, 32), (64, 130, 3, 14, 9, 16), (17, 40, 5, 7, 9, 32), (64, 2, 28, 28, 1, 16), (130, 70, 1, 37, 1, 16),
])
@pytest.mark.parametrize("fmode", ["none", "relu", "tanh", "raw"])
@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
@pytest.mark.parametrize("layout", ["panel", "slice"])
def test_conv_tangent(cin, cout, H, W, taps, nc, fmode, precision, layout, kernel_scope):
    """layout 'panel' = [sample][channel][pixel][nc]; 'slice' = the slice-major hidden layout
    [sample][pixel][16-column slice][channel][16] addressed through x_sl / y_sl (include/cmf_amd.h)."""
    from cmf_amd import engine as E
    kernel_scope(tangent=precision)
    if precision == "bf16x3" and not E._use_bf16x3(taps, cin, W, False, H, cout):
        pytest.skip("shape not covered by the split-precision kernel (the engine falls back to fp32)")
    if layout == "slice" and (cin, cout, H) not in ((64, 64, 14), (64, 64, 28), (64, 64, 32), (2, 64, 14), (64, 4, 14), (16, 40, 5),
                                                   (32, 64, 14), (64, 32, 14), (128, 64, 14), (96, 128, 6)):
        pytest.skip("slice-major layout: a subset of the shapes is enough")
    gen = torch.Generator().manual_seed(cin * 1000 + cout + H)
    B = 3
    w = torch.randn(cout, cin, 3 if taps == 9 else 1, 3 if taps == 9 else 1, generator=gen) / (cin * taps) ** 0.5
    x = torch.randn(B, cin, H, W, nc, generator=gen)
    prim = torch.randn(B, cin, H, W, generator=gen)
    res = torch.randn(B, cout, H, W, nc, generator=gen)
    fac = {"none": torch.ones_like(prim), "relu": (prim > 0).float(), "tanh": 1 - torch.tanh(prim) ** 2,
           "raw": (prim > 0.3).float()}[fmode]
    src = {"none": None, "relu": prim, "tanh": torch.tanh(prim), "raw": (prim > 0.3).float()}[fmode]
    xin = (x * fac.unsqueeze(-1)).permute(0, 4, 1, 2, 3).reshape(B * nc, cin, H, W)
    want = F.conv2d(xin, w, padding=1 if taps == 9 else 0).reshape(B, nc, cout, H, W).permute(0, 2, 3, 4, 1) + res
    HW = H * W
    wd = torch.nn.Parameter(w.cuda())
    if layout == "panel":
        to_dev = lambda t: t.cuda()
        from_dev = lambda t, c: t
        st = lambda c: (c * HW * nc, HW * nc, nc)
        sl = lambda c: 16
    else:
        S = nc // 16
        to_dev = lambda t: t.reshape(B, -1, HW, S, 16).permute(0, 2, 3, 1, 4).contiguous().cuda()
        from_dev = lambda t, c: t.reshape(B, HW, S, c, 16).permute(0, 3, 1, 2, 4).reshape(B, c, H, W, nc)
        st = lambda c: (c * HW * nc, 16, c * nc)
        sl = lambda c: c * 16
    y = torch.full((B * cout * HW * nc,), float("nan"), device="cuda")
    E.conv_tangent(to_dev(x), 0, *st(cin), wd, taps, y, *st(cout), B, cin, cout, H, W, nc,
                   fmode={"none": E.F_NONE, "relu": E.F_RELU, "tanh": E.F_TANH, "raw": E.F_RAW}[fmode],
                   f=None if src is None else src.cuda(), f_np=cin * HW, f_ci=HW, f_px=1, res_t=to_dev(res),
                   x_sl=sl(cin), y_sl=sl(cout))
    y = from_dev(y, cout).reshape(B, cout, H, W, nc)
    # bf16x3: ~2^-16 per product, well inside the same bound as the fp32 kernel for these K sizes
    assert rel(y, want) < 2e-5


@pytest.mark.parametrize("cin,cout,H,W,nc", [(1, 64, 28, 28, 64), (2, 64, 14, 14, 32), (3, 64, 32, 32, 16), (1, 128, 5, 14, 16), (3, 64, 8, 8, 128),
                                             (2, 192, 6, 9, 16), (1, 64, 1, 1, 16), (3, 64, 3, 70, 16),
                                             (2, 64, 3, 39, 16), (1, 64, 2, 90, 16)])     # the widest row image the VALU kernel stages / one beyond
@pytest.mark.parametrize("fmode", ["none", "raw", "raw0"])
@pytest.mark.parametrize("layout", ["panel", "slice"])
def test_conv_tangent_thin_input(cin, cout, H, W, nc, fmode, layout):
    """A coupler's FIRST conv (1 - 3 input channels -> 64-channel groups, no residual): cmf_conv_tangent runs it on the VALU
    write-stream kernel (conv_tangent_thin_kernel) instead of the MFMA one.  Against float64 ``F.conv2d``; input factor none, a
    per-sample RAW factor, and the checkerboard mask's form (RAW with f_np = 0: one plane for every sample); odd image sizes, one
    pixel, more than one channel group, 1 / 2 / 8 column slices; panel input -> slice-major hidden output as net_tangent issues it."""
    from cmf_amd import engine as E
    gen = torch.Generator().manual_seed(cin * 100 + H + W)
    B, HW = 3, H * W
    w = torch.randn(cout, cin, 3, 3, generator=gen) / (cin * 9) ** 0.5
    x = torch.randn(B, cin, H, W, nc, generator=gen)
    fac = (torch.rand(B if fmode == "raw" else 1, cin, H, W, generator=gen) > 0.4).float() * 1.5
    xin = x * fac.unsqueeze(-1) if fmode != "none" else x
    want = F.conv2d(xin.permute(0, 4, 1, 2, 3).reshape(B * nc, cin, H, W).double(), w.double(), padding=1)
    want = want.reshape(B, nc, cout, H, W).permute(0, 2, 3, 4, 1)
    wd = torch.nn.Parameter(w.cuda())
    S = nc // 16
    if layout == "panel":
        st, sl, back = (cout * HW * nc, HW * nc, nc), 16, (lambda t: t.reshape(B, cout, H, W, nc))
    else:
        st, sl = (cout * HW * nc, 16, cout * nc), cout * 16
        back = lambda t: t.reshape(B, HW, S, cout, 16).permute(0, 3, 1, 2, 4).reshape(B, cout, H, W, nc)
    y = torch.full((B * cout * HW * nc,), float("nan"), device="cuda")
    with E.timing(lambda n: True) as timer:
        E.conv_tangent(x.cuda(), 0, cin * HW * nc, HW * nc, nc, wd, 9, y, *st, B, cin, cout, H, W, nc,
                       fmode=E.F_NONE if fmode == "none" else E.F_RAW, f=None if fmode == "none" else fac.cuda(),
                       f_np=cin * HW if fmode == "raw" else 0, f_ci=HW, f_px=1, y_sl=sl, precision="f32")
    assert list(timer.by_name()) == [f"conv_tangent_t9_ci{cin}_co{cout}"]
    assert rel(back(y), want) < 2e-6


@pytest.mark.parametrize("cin,cout,H,W,taps,nc", [
    (64, 64, 14, 14, 9, 32), (64, 64, 6, 7, 9, 64), (128, 64, 5, 9, 9, 32), (64, 128, 28, 28, 9, 64), (2, 64, 14, 14, 9, 16), (1, 64, 8, 8, 9, 16), (64, 4, 14, 14, 1, 32),
    (128, 64, 4, 14, 9, 16), (96, 130, 3, 5, 9, 16), (17, 40, 5, 7, 9, 32), (130, 70, 1, 37, 1, 16), (10, 128, 1, 50, 1, 16),
    # thin shapes (conv_wgrad_thin_kernel): n = ci * 9 + tap in 2 / 4 tiles, a padded output tile, three input blocks of a thin 1x1
    (3, 64, 8, 8, 9, 16), (6, 64, 8, 8, 9, 32), (7, 24, 5, 6, 9, 16), (130, 12, 3, 9, 1, 16),
])
@pytest.mark.parametrize("fmode", ["none", "relu", "tanh", "self", "bits"])
@pytest.mark.parametrize("layout", ["panel", "slice"])
@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_conv_tangent_weight_gradient(cin, cout, H, W, taps, nc, fmode, layout, precision):
    """dW of  <gy, conv(F x)>  against torch.autograd in float64; accumulates into an existing gradient.  bf16x3 = the
    split-precision kernel with operands shared through LDS (whole 64-channel blocks, column pairs, factor none / relu)."""
    from cmf_amd import engine as E
    if precision == "bf16x3" and not (taps == 9 and cin % 64 == 0 and cout % 64 == 0 and nc % 32 == 0 and fmode in ("none", "relu", "self", "bits")):
        pytest.skip("not covered by the split-precision weight-gradient kernel (the engine uses the fp32 one)")
    if fmode == "bits" and precision == "f32":
        pytest.skip("relu' bit masks are read by the split-precision kernel only")
    if layout == "slice" and cin not in (64, 2, 128, 17, 6):
        pytest.skip("slice-major layout: a subset of the shapes is enough")
    gen = torch.Generator().manual_seed(cin * 1000 + cout + H)
    B, HW, k = 3, H * W, 3 if taps == 9 else 1
    x = torch.randn(B, cin, H, W, nc, generator=gen)
    gy = torch.randn(B, cout, H, W, nc, generator=gen)
    prim = torch.randn(B, cin, H, W, generator=gen)
    fac = {"none": torch.ones_like(prim), "relu": (prim > 0).float(), "tanh": 1 - torch.tanh(prim) ** 2,
           "self": torch.ones_like(prim), "bits": (prim > 0).float()}[fmode]
    src = {"none": None, "relu": prim, "tanh": torch.tanh(prim), "self": None, "bits": None}[fmode]
    w = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    xe = torch.relu(x) if fmode == "self" else x                   # SELF_RELU: the input's own relu (primal data in the column slots)
    xin = (xe * fac.unsqueeze(-1)).permute(0, 4, 1, 2, 3).reshape(B * nc, cin, H, W).double()
    y = F.conv2d(xin, w, padding=1 if taps == 9 else 0)
    (y * gy.permute(0, 4, 1, 2, 3).reshape(B * nc, cout, H, W).double()).sum().backward()
    if layout == "panel":
        to_dev = lambda t: t.cuda()
        st = lambda c: (c * HW * nc, HW * nc, nc)
        sl = lambda c: 16
    else:
        S = nc // 16
        to_dev = lambda t: t.reshape(B, -1, HW, S, 16).permute(0, 2, 3, 1, 4).contiguous().cuda()
        st = lambda c: (c * HW * nc, 16, c * nc)
        sl = lambda c: c * 16
    prev = torch.randn(cout, cin, k, k, generator=gen)
    dw = prev.clone().cuda()
    E.conv_tangent_wgrad(to_dev(x), 0, *st(cin), to_dev(gy), 0, *st(cout), dw, taps, B, cin, cout, H, W, nc,
                         fmode={"none": E.F_NONE, "relu": E.F_RELU, "tanh": E.F_TANH, "self": E.F_SELF_RELU, "bits": E.F_NONE}[fmode],
                         f=E.relu_bits(prim.cuda()) if fmode == "bits" else None if src is None else src.cuda(),
                         f_np=cin * HW, f_ci=HW, f_px=1, x_sl=sl(cin), y_sl=sl(cout), precision=precision)
    assert rel(dw.cpu() - prev, w.grad) < (2e-5 if precision == "f32" else 5e-5)


@pytest.mark.parametrize("cin,cout,H,W,taps", [(64, 64, 28, 28, 9), (1, 64, 28, 28, 9), (64, 2, 28, 28, 1), (2, 64, 14, 14, 9),
                                               (3, 64, 32, 32, 9), (64, 24, 16, 16, 1), (5, 33, 7, 19, 9), (70, 130, 1, 300, 1)])
@pytest.mark.parametrize("imode", ["none", "relu", "raw"])
def test_conv_primal(cin, cout, H, W, taps, imode):
    from cmf_amd import engine as E
    gen = torch.Generator().manual_seed(cin + cout * 7 + H)
    B = 2
    k = 3 if taps == 9 else 1
    w = torch.randn(cout, cin, k, k, generator=gen) / (cin * taps) ** 0.5
    b = torch.randn(cout, generator=gen)
    x = torch.randn(B, cin, H, W, generator=gen)
    mask = (torch.randn(cin, H, W, generator=gen) > 0).float()
    res = torch.randn(B, cout, H, W, generator=gen)
    sw, sb = torch.rand(cout, generator=gen) + 0.5, torch.randn(cout, generator=gen)
    xin = {"none": x, "relu": torch.relu(x), "raw": x * mask}[imode]
    v = F.conv2d(xin, w, b, padding=1 if taps == 9 else 0) + res
    HW = H * W
    wd, y, gout = torch.nn.Parameter(w.cuda()), torch.empty(B, cout, H, W, device="cuda"), torch.empty(B, cout, H, W, device="cuda")
    E.conv_primal(x.cuda(), 0, cin * HW, HW, 1, wd, taps, b.cuda(), y, cout * HW, HW, 1, B, cin, cout, H, W,
                  imode={"none": E.F_NONE, "relu": E.F_RELU, "raw": E.F_RAW}[imode], mask=mask.cuda(), f_c=HW, f_px=1,
                  omode=E.O_STANH, sw=sw.cuda(), sb=sb.cuda(), g=gout, res=res.cuda())
    t = torch.tanh(v)
    assert rel(y, sw.view(1, -1, 1, 1) * t + sb.view(1, -1, 1, 1)) < 2e-5
    assert rel(gout, sw.view(1, -1, 1, 1) * (1 - t * t)) < 2e-5


@pytest.mark.parametrize("N,d,layout", [(784, 64, "panel"), (3072, 128, "panel"), (21, 10, "fmajor"), (3, 3, "fmajor"), (50, 33, "panel")])
def test_gram_cholesky(N, d, layout):
    from cmf_amd import engine as E
    gen = torch.Generator().manual_seed(N + d)
    B, nc = 5, E.ceil16(d)
    J = torch.zeros(B, N, nc)
    J[:, :, :d] = torch.randn(B, N, d, generator=gen) + (0.5 if N < d + 2 else 0) * torch.eye(N, d)
    data = J if layout == "panel" else J.permute(1, 0, 2)
    T = E.Tangent(B, N, nc, layout, "cuda", data=data.contiguous().reshape(-1).cuda())
    gr = E.gram_cholesky(T, d)
    G = torch.einsum("bni,bnj->bij", J[:, :, :d].double(), J[:, :, :d].double())
    assert rel(gr.jtj, G) < 1e-5
    assert rel(gr.logdet, torch.linalg.slogdet(G)[1]) < 1e-4
    diag = torch.diagonal(G, dim1=1, dim2=2).abs().sum(1)
    assert rel(gr.l1_diag, diag) < 1e-5 and rel(gr.l1_off, G.abs().sum((1, 2)) - diag) < 1e-5
    assert gr.fail.tolist()[0] == 0


@pytest.mark.parametrize("N,d,layout", [(784, 64, "panel"), (300, 128, "panel"), (100, 20, "fmajor"), (48, 2, "panel"), (50, 33, "fmajor")])
def test_gram_backward_matches_autograd(N, d, layout):
    """dJ of  sum_b a_b logdet(J^T J) + o_b sum_{i!=j} |G_ij| + c_b sum_i |G_ii|  against torch.autograd in float64."""
    from cmf_amd import engine as E
    gen = torch.Generator().manual_seed(3 * N + d)
    B, nc = 6, E.ceil16(d)
    J = torch.zeros(B, N, nc)
    J[:, :, :d] = torch.randn(B, N, d, generator=gen)
    ga, go, gd = (torch.randn(B, generator=gen) for _ in range(3))
    data = J if layout == "panel" else J.permute(1, 0, 2)
    T = E.Tangent(B, N, nc, layout, "cuda", data=data.contiguous().reshape(-1).cuda())
    gr = E.gram_cholesky(T, d)
    assert gr.fail.tolist()[0] == 0
    Jd = J[:, :, :d].double().requires_grad_(True)
    G = torch.einsum("bni,bnj->bij", Jd, Jd)
    diag = torch.diagonal(G, dim1=1, dim2=2).abs().sum(1)
    loss = (ga.double() * torch.linalg.slogdet(G)[1] + go.double() * (G.abs().sum((1, 2)) - diag) + gd.double() * diag).sum()
    loss.backward()
    for sel in [(1, 1, 1), (1, 0, 0), (0, 1, 0), (0, 0, 1)]:
        if sel == (1, 1, 1):
            want = Jd.grad
        else:
            Jd.grad = None
            G = torch.einsum("bni,bnj->bij", Jd, Jd)
            diag = torch.diagonal(G, dim1=1, dim2=2).abs().sum(1)
            (sel[0] * ga.double() * torch.linalg.slogdet(G)[1] + sel[1] * go.double() * (G.abs().sum((1, 2)) - diag)
             + sel[2] * gd.double() * diag).sum().backward()
            want = Jd.grad
        args = [g.cuda() if on else None for g, on in zip((ga, go, gd), sel)]
        dT = E.gram_backward(T, gr.jtj, *args)
        got = dT.to_dense(nc)
        assert rel(got[:, :, :d], want) < 1e-4
        assert float(got[:, :, d:].abs().max()) == 0.0 if d < nc else True


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
@pytest.mark.parametrize("H,W", [(14, 14), (28, 28), (5, 14)])
def test_conv_tangent_primal_in_column_slots(H, W, precision, kernel_scope):
    """16 samples in the 16 column slots: elementwise relu on load (SELF_RELU), bias, residual; and a tangent launch
    reading its relu' factor from that sample-grouped primal tensor (f_group = 16)."""
    from cmf_amd import engine as E
    kernel_scope(tangent=precision)
    gen = torch.Generator().manual_seed(H * W)
    B, C, HW = 32, 64, H * W
    x = torch.randn(B, C, H, W, generator=gen)
    w = torch.randn(C, C, 3, 3, generator=gen) / 24
    bias = torch.randn(C, generator=gen)
    res = torch.randn(B, C, H, W, generator=gen)
    want = F.conv2d(torch.relu(x), w, bias, padding=1) + res
    xg, rg = E.primal_regroup(x.cuda(), True), E.primal_regroup(res.cuda(), True)
    assert torch.equal(E.primal_regroup(xg.view(B // 16, -1), False).view(B, C, H, W).cpu(), x)
    yg = torch.empty_like(xg)
    wd = torch.nn.Parameter(w.cuda())
    pn = (C * HW * 16, HW * 16, 16)
    E.conv_tangent(xg, 0, *pn, wd, 9, yg, *pn, B // 16, C, C, H, W, 16, fmode=E.F_SELF_RELU, bias=bias.cuda(), res_t=rg)
    got = E.primal_regroup(yg.view(B // 16, -1), False).view(B, C, H, W)
    assert rel(got, want) < 2e-5
    # tangent pass with the grouped primal as factor source
    nc = 16
    v = torch.randn(B, C, H, W, nc, generator=gen)
    wantv = F.conv2d(((x > 0).float().unsqueeze(-1) * v).permute(0, 4, 1, 2, 3).reshape(B * nc, C, H, W), w, padding=1)
    wantv = wantv.reshape(B, nc, C, H, W).permute(0, 2, 3, 4, 1)
    yv = torch.empty(B, C, H, W, nc, device="cuda")
    E.conv_tangent(v.cuda(), 0, C * HW * nc, HW * nc, nc, wd, 9, yv, C * HW * nc, HW * nc, nc, B, C, C, H, W, nc,
                   fmode=E.F_RELU, f=xg, f_np=C * HW * 16, f_ci=HW * 16, f_px=16, f_group=16)
    assert rel(yv, wantv) < 2e-5


@pytest.mark.parametrize("H,W", [(14, 14), (16, 16)])
def test_relu_bit_masks_producer_and_consumer(H, W, kernel_scope):
    """The fp32 kernel's ``mask_out`` (sign bits of what it stores, one bit per sample / pixel / channel) and the split
    kernel's CMF_F_RELU_BITS factor mode: same tangent conv result as the float-activation RELU mode, bit for bit."""
    from cmf_amd import engine as E
    kernel_scope(tangent="bf16x3")
    gen = torch.Generator().manual_seed(H)
    B, C, nc, HW = 32, 64, 32, H * W
    G = B // 16
    # producer: a primal conv with 16 samples in the column slots writes activations (grouped) and their sign bits
    x = torch.randn(B, C, H, W, generator=gen)
    w1 = torch.nn.Parameter((torch.randn(C, C, 3, 3, generator=gen) / 24).cuda())
    b1 = torch.randn(C, generator=gen).cuda()
    xg = E.primal_regroup(x.cuda(), True)
    act_g = torch.empty(G * C * HW * 16, device="cuda")
    m = E.BitMask(B, HW, C, "cuda")
    m.data.fill_(0xAA)
    pn = (C * HW * 16, HW * 16, 16)
    E.conv_tangent(xg, 0, *pn, w1, 9, act_g, *pn, G, C, C, H, W, 16, fmode=E.F_SELF_RELU, bias=b1, precision="f32",
                   mask_out=m.data, mask_np=m.np_bytes)
    act = E.primal_regroup(act_g.view(G, -1), False).view(B, C, H, W)
    want_bits = (act > 0).permute(0, 2, 3, 1).reshape(B, HW, C // 8, 8)                     # bit j of byte o = channel 8*o + j
    want = (want_bits.to(torch.int32) << torch.arange(8, device="cuda", dtype=torch.int32)).sum(-1).to(torch.uint8)
    assert torch.equal(m.data, want)
    # consumer: the same tangent conv with relu' from the floats and from the bits
    st, sl = (C * HW * nc, 16, C * nc), C * 16
    T = torch.randn(B * C * HW * nc, generator=gen).cuda()
    w2 = torch.nn.Parameter((torch.randn(C, C, 3, 3, generator=gen) / 24).cuda())
    y1, y2 = torch.empty_like(T), torch.empty_like(T)
    E.conv_tangent(T, 0, *st, w2, 9, y1, *st, B, C, C, H, W, nc, fmode=E.F_RELU, f=act, f_np=C * HW, f_ci=HW, f_px=1, x_sl=sl, y_sl=sl)
    E.conv_tangent(T, 0, *st, w2, 9, y2, *st, B, C, C, H, W, nc, fmode=E.F_RELU_BITS, f=m.data, f_np=m.np_bytes, x_sl=sl, y_sl=sl)
    assert torch.equal(y1, y2)


@pytest.mark.parametrize("layout,with_g", [("panel", True), ("fmajor", False), ("panel", False)])
def test_acl_cross_terms_match_autograd(layout, with_g):
    """Primal cotangents of the coupling layer's tangent update  out = e^{-s} (v - z gs sd) - gt td  (column reductions)."""
    from cmf_amd import engine as E
    gen = torch.Generator().manual_seed(17)
    B, N, NY, n_mod, nc, S = 4, 30, 24, 11, 32, 20
    zi = torch.randperm(N, generator=gen)[:n_mod].int()
    ti = torch.randperm(NY // 2, generator=gen)[:n_mod].int()
    si = ti + NY // 2
    maps = {"zi": zi.cuda(), "si": si.cuda(), "ti": ti.cuda(), "n": n_mod}
    z, y = torch.randn(B, N, generator=gen), 0.5 * torch.randn(B, NY, generator=gen)
    g = torch.rand(B, NY, generator=gen) + 0.2 if with_g else None
    t_in, yt, c = (torch.randn(B, n, S, generator=gen) for n in (N, NY, N))
    zd, yd = z.double().requires_grad_(True), y.double().requires_grad_(True)
    gd = g.double().requires_grad_(True) if with_g else torch.ones(B, NY, dtype=torch.float64)
    zl, sl, tl = zi.long(), si.long(), ti.long()
    out = torch.exp(-yd[:, sl]).unsqueeze(-1) * (t_in.double()[:, zl] - (zd[:, zl] * gd[:, sl]).unsqueeze(-1) * yt.double()[:, sl]) \
        - gd[:, tl].unsqueeze(-1) * yt.double()[:, tl]
    (out * c.double()[:, zl]).sum().backward()
    T = E.Tangent.from_dense(t_in.cuda(), nc, layout)
    V = E.modified_rows(T, maps)
    assert rel(V.to_dense(S), t_in[:, zl]) == 0.0
    dz, dy = torch.zeros(B, N).cuda(), torch.zeros(B, NY).cuda()
    dg = torch.zeros(B, NY).cuda() if with_g else None
    E.acl_cross_terms(E.Tangent.from_dense(c.cuda(), nc, layout), V, E.Tangent.from_dense(yt.cuda(), nc, layout), z.cuda(), y.cuda(),
                      None if g is None else g.cuda(), maps, dz, dy, dg)
    assert rel(dz, zd.grad) < 1e-5 and rel(dy, yd.grad) < 1e-5
    if with_g:
        assert rel(dg, gd.grad) < 1e-5


def test_scaled_tanh_backward_matches_autograd():
    from cmf_amd import engine as E
    gen = torch.Generator().manual_seed(23)
    B, Cc, H, W = 5, 4, 7, 6
    u = torch.randn(B, Cc, H, W, generator=gen).double().requires_grad_(True)
    sw = (torch.randn(Cc, 1, 1, generator=gen) + 1.5).double().requires_grad_(True)
    sb = torch.randn(Cc, 1, 1, generator=gen).double().requires_grad_(True)
    dy, dg = torch.randn(B, Cc, H, W, generator=gen), torch.randn(B, Cc, H, W, generator=gen)
    t = torch.tanh(u)
    y, g = sw * t + sb, sw * (1 - t * t)
    for use_dg in (True, False):
        gu, gw, gb = torch.autograd.grad((y * dy.double()).sum() + ((g * dg.double()).sum() if use_dg else 0), [u, sw, sb],
                                         retain_graph=True)
        dsw, dsb = torch.ones(Cc).cuda(), torch.ones(Cc).cuda()     # accumulated into
        du = E.stanh_backward(dy.cuda(), dg.cuda() if use_dg else None, y.detach().float().cuda(), g.detach().float().cuda(),
                              sw.detach().float().cuda(), sb.detach().float().cuda(), dsw, dsb)
        assert rel(du, gu) < 1e-5 and rel(dsw - 1, gw.reshape(-1)) < 1e-5 and rel(dsb - 1, gb.reshape(-1)) < 1e-5


@pytest.mark.parametrize("layout", ["panel", "slice"])
def test_channel_sum(layout):
    from cmf_amd import engine as E
    gen = torch.Generator().manual_seed(29)
    B, Cc, HW, nc = 3, 40, 35, 32
    t = torch.randn(B, Cc, HW, nc, generator=gen)
    out = torch.full((Cc,), 2.0).cuda()
    if layout == "panel":
        E.channel_sum(t.cuda(), Cc * HW * nc, HW * nc, nc, B, Cc, HW, nc, out)
    else:
        dev = t.reshape(B, Cc, HW, nc // 16, 16).permute(0, 2, 3, 1, 4).contiguous().cuda()
        E.channel_sum(dev, Cc * HW * nc, 16, Cc * nc, B, Cc, HW, nc, out, t_sl=Cc * 16)
    assert rel(out - 2, t.double().sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize("H,W,cl_out,cl_in", [(14, 14, 64, 64), (28, 28, 64, 64), (8, 16, 64, 128), (6, 14, 128, 64)])
@pytest.mark.parametrize("layout", ["panel", "slice"])
def test_transposed_conv_with_output_bit_mask_on_the_split_kernel(H, W, cl_out, cl_in, layout, kernel_scope):
    """Reverse-sweep conv on the split-precision kernel: transposed / tap-flipped bf16x3 pack, NO input factor, relu' of a
    float activation as an OUTPUT bit mask (cmf_relu_bits), then the skip connection through cmf_accumulate:
    y = skip + [act > 0] . conv^T(x)   against conv_transpose2d."""
    from cmf_amd import engine as E
    kernel_scope(tangent="bf16x3")
    gen = torch.Generator().manual_seed(H * W + cl_in)
    B, nc, HW = 2, 32, H * W
    w = torch.randn(cl_out, cl_in, 3, 3, generator=gen) / (9 * cl_out) ** 0.5        # the LAYER's weight: cl_in -> cl_out
    x = torch.randn(B, cl_out, H, W, nc, generator=gen)                               # cotangent of the layer's output
    act = torch.randn(B, cl_in, H, W, generator=gen)
    skip = torch.randn(B, cl_in, H, W, nc, generator=gen)
    xin = x.permute(0, 4, 1, 2, 3).reshape(B * nc, cl_out, H, W)
    want = F.conv_transpose2d(xin, w, padding=1).reshape(B, nc, cl_in, H, W).permute(0, 2, 3, 4, 1)
    want = skip + (act > 0).float().unsqueeze(-1) * want
    if layout == "panel":
        to_dev = lambda t: t.contiguous().cuda()
        from_dev = lambda t, c: t
        st = lambda c: (c * HW * nc, HW * nc, nc)
        sl = lambda c: 16
    else:
        S = nc // 16
        to_dev = lambda t: t.reshape(B, -1, HW, S, 16).permute(0, 2, 3, 1, 4).contiguous().cuda()
        from_dev = lambda t, c: t.reshape(B, HW, S, c, 16).permute(0, 3, 1, 2, 4).reshape(B, c, H, W, nc)
        st = lambda c: (c * HW * nc, 16, c * nc)
        sl = lambda c: c * 16
    bits = E.relu_bits(act.cuda())
    ref_bits = ((act > 0).permute(0, 2, 3, 1).reshape(B, HW, cl_in // 8, 8).int() * (1 << torch.arange(8)).int()).sum(-1)
    assert torch.equal(bits.data.cpu().int(), ref_bits)
    y = torch.full((B * cl_in * HW * nc,), float("nan"), device="cuda")
    wd = torch.nn.Parameter(w.cuda())
    E.conv_tangent(to_dev(x), 0, *st(cl_out), wd, 9, y, *st(cl_in), B, cl_out, cl_in, H, W, nc, fo=bits, transpose=True,
                   x_sl=sl(cl_out), y_sl=sl(cl_in))
    E.accumulate(y, to_dev(skip).reshape(-1))
    assert rel(from_dev(y, cl_in).reshape(B, cl_in, H, W, nc), want) < 2e-5
    if True:
        # the same skip connection IN PLACE: the kernel starts from the output tensor and the lanes the mask switches off do not
        # store -- masked-off entries must come back BIT-identical
        y2 = to_dev(skip).reshape(-1).clone()
        before = y2.clone()
        E.conv_tangent(to_dev(x), 0, *st(cl_out), wd, 9, y2, *st(cl_in), B, cl_out, cl_in, H, W, nc, fo=bits, transpose=True,
                       x_sl=sl(cl_out), y_sl=sl(cl_in), res_t=y2)
        got = from_dev(y2, cl_in).reshape(B, cl_in, H, W, nc)
        assert rel(got, want) < 2e-5
        off = ~(act > 0).unsqueeze(-1).expand_as(got)
        assert torch.equal(got.cpu()[off], from_dev(before, cl_in).reshape(B, cl_in, H, W, nc).cpu()[off])
