"""Does touching weights ahead of time (Infinity Cache / MALL, 256 MB) speed up the HBM-bound decode GEMV?  (not product)
Rotates 4 weight sets (1.09 GB) so that nothing is resident by accident; 'prefetch' = torch sum over the first X MB of the
set that the NEXT gemv call will stream."""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omchat_amd import _lib
lib = _lib.lib()
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
N, K = 37888, 3584
ws = [(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for _ in range(4)]
x = torch.randn(K, device="cuda").bfloat16()
y = torch.empty(N // 2, dtype=torch.bfloat16, device="cuda")
filler = torch.randn(64, 1024, device="cuda")      # stands for the latency-bound small kernels (HBM idle)
def gemv(i): _lib.check(lib.omchat_op_gemv(_lib.BF16, p(x), K, p(ws[i % 4]), K, p(y), N // 2, 1, N, K, None, None, 0, 4, 0, None))
side = torch.cuda.Stream()
if len(sys.argv) > 1: lib.omchat_op_set_tuning(1, int(sys.argv[1]))      # 1 = MFMA-form GEMV (plain loads instead of non-temporal)
for mb in (0, 32, 64, 128, 192, 256):
    ts = []
    for it in range(24):
        w = ws[(it + 1) % 4]
        n_el = mb * 1024 * 1024 // 2
        ev = torch.cuda.Event(); ev.record()
        if mb:
            with torch.cuda.stream(side):
                side.wait_event(ev)
                w.view(-1)[:n_el].view(torch.int32).sum()          # touch the first `mb` MB of the next weights
        for _ in range(6): filler = filler * 1.0001                 # ~6 tiny kernels on the main stream while the prefetch runs
        torch.cuda.current_stream().wait_stream(side) if False else None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()                                    # prefetch certainly finished: measures the pure cache effect
        e0.record(); gemv(it + 1); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[4:])
    print(f"prefetched {mb:4d} MB of 271 MB: gemv {ts[len(ts)//2]:6.1f} us (min {ts[0]:.1f})", flush=True)
