import ctypes as C, torch, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.path.join(ROOT, "cmf_amd/csrc/_obj/ubench_curate2.so"))
lib.run_curate2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p]
buf = torch.zeros(1 << 30, dtype=torch.float32, device="cuda"); out = torch.zeros(16, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
plane = 784 * 256   # bytes: 784 pixels x 64 floats
iters = 784 // 8
for mode, name in ((0, "load "), (1, "store")):
    for seg in (0, 1):
        for nb in (1, 16):
            for threads in (256,):
                waves = nb * threads // 64
                assert waves * 16 * plane <= buf.numel() * 4
                for _ in range(2): lib.run_curate2(buf.data_ptr(), out.data_ptr(), mode, seg, plane, iters, nb, threads, st)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); lib.run_curate2(buf.data_ptr(), out.data_ptr(), mode, seg, plane, iters, nb, threads, st); e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1); byt = waves * iters * 8192.0
                print(f"{name} {'16x64B planes' if seg else 'contiguous  '} CUs={nb:3d} waves/CU={threads//64}: {byt/ms/1e6:8.1f} GB/s {byt/ms/1e6/nb/2.4:6.1f} B/clk/CU", flush=True)
