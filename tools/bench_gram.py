#!/usr/bin/env python3
"""Time the fused Gram + Cholesky kernel: python tools/bench_gram.py [B N d [B N d ...]]

CMF_DBG_LIB=dbg_<FLAGS>.so selects a diagnostic build (tools/build_dbg.sh): GSTAMP adds per-workgroup phase time stamps
(Gram loop / reduction + output / elimination) which are summarised here, GRAMOLD runs the round-1 kernel, GRAMPF=<n> sets the
prefetch depth of the d <= 64 kernel."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from cmf_amd import _lib
if os.environ.get("CMF_DBG_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "cmf_amd/csrc/_obj", os.environ["CMF_DBG_LIB"])
from cmf_amd import engine as E


def stamps(B):
    lib = _lib.load()
    if not hasattr(lib, "cmf_debug_read_gram_stamps"):
        return ""
    buf = np.zeros((4096, 4), dtype=np.uint64)
    lib.cmf_debug_read_gram_stamps.argtypes = [C.c_void_p]
    assert lib.cmf_debug_read_gram_stamps(buf.ctypes.data) == 0
    s = buf[:min(B, 4096)].astype(np.int64)
    t0 = s[:, 0].min()
    ph = np.diff(s, axis=1)
    return (f"  stamps (s_memtime ticks, mean over workgroups): gram {ph[:, 0].mean():.0f}  reduce+out {ph[:, 1].mean():.0f}  "
            f"cholesky {ph[:, 2].mean():.0f}  | start spread {s[:, 0].max() - t0}  last end {s[:, 3].max() - t0}")


def run(B, N, d, iters=20):
    nc = E.ceil16(d)
    T = E.Tangent(B, N, nc, "panel", "cuda", data=torch.randn(B * N * nc, device="cuda"))
    for _ in range(3): E.gram_cholesky(T, d, max_attempts=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): E.gram_cholesky(T, d, max_attempts=1)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    fl = 2.0 * N * d * d * B
    print(f"B={B} N={N} d={d}: {us:8.1f} us  gram {fl/us/1e6:7.2f} TFLOP/s ({fl/us/1e6/157.3*100:.1f}% of fp32 MFMA peak)  {4.0*B*N*nc/us/1e3:7.1f} GB/s" + stamps(B), flush=True)


args = [int(a) for a in sys.argv[1:]]
print("lib:", os.environ.get("CMF_DBG_LIB", "libcmf_amd.so"))
if args:
    for i in range(0, len(args) - 2, 3): run(*args[i:i + 3])          # any number of "B N d" triples
else:
    run(512, 784, 64); run(512, 16, 64); run(512, 3072, 128); run(512, 16, 128); run(4096, 784, 64); run(256, 784, 64); run(512, 784, 48)
if not args:
    run(512, 784, 20); run(4096, 21, 10)
