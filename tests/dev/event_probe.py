"""HIP-event elapsed time against the host clock around the same region (dev probe)."""
import time, torch
x = torch.randn(8192, 8192, device="cuda")
def work(n=6):
    y = x
    for _ in range(n): y = y @ x
    return y
work(); torch.cuda.synchronize()
for pre in (False, True):
    for n in (2, 6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        if pre: torch.zeros(1, device="cuda")
        t0 = time.perf_counter(); e0.record(); work(n); e1.record(); torch.cuda.synchronize(); dt = 1e3 * (time.perf_counter() - t0)
        print(f"pre-kernel={pre} n={n}: wall {dt:.2f} ms, events {e0.elapsed_time(e1):.2f} ms")
