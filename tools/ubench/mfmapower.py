"""Sustained bf16 MFMA rate of the whole chip on live (random, changing) operands vs a constant operand pair: the clock the
chip holds under a pure matrix load, i.e. the reachable part of the 2.5 PFLOP/s (2.4 GHz) peak the roofline is priced against."""
import ctypes as C, torch, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.path.join(ROOT, "cmf_amd/csrc/_obj/ubench_mfmapower.so"))
lib.run_mfmapower.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
lib.run_mfmapower32.argtypes = lib.run_mfmapower.argtypes
out = torch.zeros(1024, device="cuda"); cyc = torch.zeros(1, dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, src in (("random N(0,1)", torch.randn(4096 * 8, device="cuda").to(torch.bfloat16)),
                  ("zeros", torch.zeros(4096 * 8, device="cuda", dtype=torch.bfloat16))):
    for live in (0, 1):
        for nb in (256, 512):                               # one / two waves per SIMD
            iters = 40000                                   # 64 MFMAs per iteration: ~20 ms at one wave per SIMD
            for _ in range(2):
                lib.run_mfmapower(src.data_ptr(), out.data_ptr(), cyc.data_ptr(), iters, nb, live, st); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): lib.run_mfmapower(src.data_ptr(), out.data_ptr(), cyc.data_ptr(), iters, nb, live, st)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            tf = nb * 4 * iters * 64 * 16 * 16 * 32 * 2 / ms / 1e9
            print(f"{name:14s} {'changing operands' if live else 'one operand pair '} waves/SIMD={nb // 256}: {ms:7.2f} ms  {tf:6.0f} TFLOP/s "
                  f"= {tf / 2516.6:.3f} of the 2.4 GHz peak; s_memtime clock {cyc.item() / ms / 1e6:.2f} GHz", flush=True)

# the 32x32x16 form: 32 MFMAs of 32 Kflop per iteration = the same flops per iteration as 64 of 16 Kflop
for name, src in (("random N(0,1)", torch.randn(4096 * 8, device="cuda").to(torch.bfloat16)),):
    for live in (0, 1):
        for nb in (256, 512):
            iters = 40000
            for _ in range(2):
                lib.run_mfmapower32(src.data_ptr(), out.data_ptr(), cyc.data_ptr(), iters, nb, live, st); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): lib.run_mfmapower32(src.data_ptr(), out.data_ptr(), cyc.data_ptr(), iters, nb, live, st)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            tf = nb * 4 * iters * 32 * 32 * 32 * 16 * 2 / ms / 1e9
            print(f"32x32x16 {name:14s} {'changing operands' if live else 'one operand pair '} waves/SIMD={nb // 256}: {ms:7.2f} ms  {tf:6.0f} TFLOP/s "
                  f"= {tf / 2516.6:.3f} of the 2.4 GHz peak", flush=True)
