import ctypes as C, torch, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.path.join(ROOT, "cmf_amd/csrc/_obj/ubench_mfmarate.so"))
lib.run_mfmarate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
out = torch.zeros(64, device="cuda"); cyc = torch.zeros(2, dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(2):
    lib.run_mfmarate(out.data_ptr(), cyc.data_ptr(), 1000, st); torch.cuda.synchronize()
c = cyc.cpu().tolist()
print(f"16x16x32 bf16: {c[0]/8000:.2f} cycles/MFMA   16x16x16 bf16_1k: {c[1]/8000:.2f} cycles/MFMA")
