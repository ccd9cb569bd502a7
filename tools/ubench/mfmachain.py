import ctypes as C, torch, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.path.join(ROOT, "cmf_amd/csrc/_obj/ubench_mfmachain.so"))
lib.run_mfmachain.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
out = torch.zeros(64, device="cuda"); cyc = torch.zeros(4, dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for nb, th in ((1, 64), (256, 256), (1024, 256)):      # one wave; one wave per SIMD on every CU; four per SIMD
    iters = 20000
    for _ in range(2):
        lib.run_mfmachain(out.data_ptr(), cyc.data_ptr(), iters, nb, th, st); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lib.run_mfmachain(out.data_ptr(), cyc.data_ptr(), iters, nb, th, st); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    c = cyc.cpu().tolist()
    waves = nb * th // 64
    tf = waves * 4 * iters * 8 * 16 * 16 * 32 * 2 / ms / 1e9
    print(f"blocks={nb} threads={th}: cycles per v_mfma_f32_16x16x32_bf16 with N chains: " + "  ".join(f"N={n}: {v/(8*iters):.1f}" for n, v in zip((1, 2, 4, 8), c))
          + f"   wall {ms:.2f} ms = {tf:.0f} TFLOP/s; implied clock {sum(c)/ms/1e6:.2f} GHz", flush=True)
