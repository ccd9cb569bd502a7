"""Per-CU VMEM throughput (B/clk/CU) for loads / stores, L2-resident or streaming, 1..256 active CUs, 4 or 8 waves per CU."""
import ctypes as C, torch, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = os.path.join(ROOT, "cmf_amd/csrc/_obj/ubench_curate.so")
lib = C.CDLL(so)
lib.run_curate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p]
buf = torch.zeros(1 << 30, dtype=torch.float32, device="cuda")  # 4 GiB
out = torch.zeros(16, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
GHZ = 2.4
for mode, name in ((0, "load "), (1, "store")):
    for nb in (1, 16, 256):
        for threads in (256, 512):
            for window, wn in ((64 << 10, "L2 64K/wave"), (2 << 20, "HBM 2M/wave")):
                waves = nb * threads // 64
                iters = 256 if window > (1 << 20) else 1024
                if waves * window > buf.numel() * 4: continue
                for _ in range(2): lib.run_curate(buf.data_ptr(), out.data_ptr(), mode, window, iters, nb, threads, st)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); lib.run_curate(buf.data_ptr(), out.data_ptr(), mode, window, iters, nb, threads, st); e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1); byt = waves * iters * 8192.0
                print(f"{name} CUs={nb:3d} waves/CU={threads//64} {wn}: {byt/ms/1e6:8.1f} GB/s  {byt/ms/1e6/nb/GHZ:6.1f} B/clk/CU", flush=True)
