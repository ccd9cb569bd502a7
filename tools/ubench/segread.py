import ctypes as C, torch, sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib=C.CDLL(os.path.join(ROOT,"cmf_amd/csrc/_obj/ubench_segread.so"))
lib.run_segread.argtypes=[C.c_void_p,C.c_void_p,C.c_int,C.c_int,C.c_longlong,C.c_int,C.c_int,C.c_void_p]
N=1<<30   # 4 GiB of floats? no: 1<<30 floats = 4 GiB
buf=torch.empty(N,dtype=torch.float32,device="cuda").normal_()
out=torch.zeros(16,device="cuda")
for waves_per_cu in (4,8,16):
  for seg,stride in ((64,256),(128,256),(256,256),(64,128),(1024,1024)):
    nblocks=256*waves_per_cu//4; threads=256
    nwaves=nblocks*4
    rows=(buf.numel()*4//stride)//nwaves; rows-=rows%(64//(seg//16)*8)
    st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2): lib.run_segread(buf.data_ptr(),out.data_ptr(),seg,stride,rows,nblocks,threads,st)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); lib.run_segread(buf.data_ptr(),out.data_ptr(),seg,stride,rows,nblocks,threads,st); e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1); useful=nwaves*rows*seg
    print(f"waves/CU={waves_per_cu:2d} seg={seg:4d}B stride={stride:4d}B: {useful/ms/1e6:8.1f} GB/s useful ({useful/ms/1e6/256/2.3:.1f} B/clk/CU)")
