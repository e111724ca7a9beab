"""GEMM fixed-cost probe (not product): time vs K at fixed (M, N) for one tile config -> per-K-step slope and K = 0 intercept
(launch + prologue + epilogue + end-of-kernel write-back).  python tools/bench_gemm_k.py [tile]"""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omchat_amd import _lib
lib = _lib.lib()
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 2
p = lambda t: C.c_void_p(t.data_ptr())
flush = torch.empty(384 << 20, dtype=torch.uint8, device="cuda")
for M, N in ((3075, 12800), (3072, 12800), (3072, 3328), (3072, 9600), (256, 256)):
    for epi in (0, 1, 2):
        line = f"M={M:5d} N={N:5d} epi={epi} tile={tile}:"
        pts = []
        for K in (64, 512, 1600, 3200, 6400):
            A = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16(); W = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.02).bfloat16()
            bias = torch.zeros(N, device="cuda", dtype=torch.bfloat16); ls = torch.ones(N, device="cuda", dtype=torch.bfloat16)
            out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16); res = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
            def run():
                _lib.check(lib.omchat_op_gemm(_lib.BF16, p(A), K, p(W), K, p(out), N, M, N, K, p(bias) if epi in (1, 2) else None, p(ls), p(res), N, epi, tile, None))
            for _ in range(3): run()
            torch.cuda.synchronize()
            best = 1e9
            for cold in (0, 1):
                ts = []
                for _ in range(6):
                    if cold: flush.fill_(1)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); run(); e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                pts.append((K, cold, sorted(ts)[len(ts) // 2]))
        for cold in (0, 1):
            line += ("  cold:" if cold else "  warm:") + " ".join(f"K{k}={us:6.1f}" for k, c, us in pts if c == cold)
        print(line, flush=True)
