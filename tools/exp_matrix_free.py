#!/usr/bin/env python3
"""VERDICT r3 missing #5: what ONE matrix-free product (J^T J) v costs on the C5 shard (CIFAR d = 128, 32 samples, S = 4 probes):
``head.jtj_matvec`` = one JVP sweep over 16 column slots + a primal decode keeping the layers' state + one reverse sweep
(non_square.py:190-201).  The reference's CG (max_iter = min(cfg, d), tolerance 1: 11 iterations, + the surrogate's own product =
12 products) against the ONE d-column sweep + explicit Gram matrix + cmf_hutch_cg that train mode uses here.

  python tools/exp_matrix_free.py [--batch 32] [--S 4]
"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import Workload
from cmf_amd import engine as E
B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 32
S = int(sys.argv[sys.argv.index("--S") + 1]) if "--S" in sys.argv else 4
wl = Workload("c5", B, 0, torch.device("cuda"))
a, c, logit, head = wl.inner._fused_prehead()
prog = head.program


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    y, _ = E.prehead(wl.x, None, a, c, logit)
    z_low = prog.encode(y)[0]
    v = torch.randn(B, prog.d, S, device="cuda")
    t_mv = timed(lambda: head.jtj_matvec(z_low, v))
    t_jvp = timed(lambda: prog.decode(z_low, tangents=True, eps=v))
    t_full = timed(lambda: prog.decode(z_low, tangents=True))

    def explicit():
        x_hat, T = prog.decode(z_low, tangents=True)
        g = E.gram_cholesky(T, prog.d, 1)
        E.hutch_cg(g.jtj, v, prog.d, 1.0)
    t_exp = timed(explicit)
print(f"C5 shard, {B} samples, d = {prog.d}, S = {S} probes (16 column slots):")
print(f"  one matrix-free product (J^T J) v   = JVP sweep + state-keeping primal decode + reverse sweep : {t_mv:8.2f} ms")
print(f"  (its JVP sweep alone: {t_jvp:.2f} ms)")
print(f"  reference-style CG, 11 iterations + the surrogate's product = 12 products                       : {12 * t_mv:8.2f} ms")
print(f"  d-column sweep ({t_full:.2f} ms) + explicit Gram + cmf_hutch_cg on it (what train mode runs)     : {t_exp:8.2f} ms")
