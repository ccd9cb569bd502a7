#!/usr/bin/env python3
"""Per-(FlowProgram phase, libcmf_amd symbol) kernel times of one evaluation: HIP events around EVERY launch (cmf_amd._lib.trace).

  python tools/stage_table.py [c3|c5|c2b ...] [--batch N]
"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import CONFIGS, Workload
from cmf_amd import _lib
args = [a for a in sys.argv[1:] if not a.startswith("--")] or ["c3"]
batch = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else None
for cfgname in args:
    wl = Workload(cfgname, batch or CONFIGS[cfgname][2], 0, torch.device("cuda"))
    with torch.no_grad():
        wl.inner.elbo(wl.x, **wl.kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); wl.inner.elbo(wl.x, **wl.kw); e1.record(); torch.cuda.synchronize()
        with _lib.trace() as rec:
            wl.inner.elbo(wl.x, **wl.kw)
            torch.cuda.synchronize()
    rows = {}
    for name, ph, a, b in rec:
        n, ms = rows.get((ph, name), (0, 0.0))
        rows[(ph, name)] = (n + 1, ms + a.elapsed_time(b))
    tot = sum(v[1] for v in rows.values())
    print(f"{wl.label}; B = {wl.B}: one eager evaluation {e0.elapsed_time(e1):.3f} ms; traced kernels {tot:.3f} ms in {sum(v[0] for v in rows.values())} launches")
    for (ph, name), (n, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print(f"  {str(ph):24s} {name:32s} {n:5d} launches {1e3 * ms / n:9.1f} us avg {ms:9.3f} ms {100 * ms / tot:5.1f} %")
    del wl
    torch.cuda.empty_cache()
