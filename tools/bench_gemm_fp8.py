"""fp8 x fp8 GEMM (BASELINE configs[4] prefill shapes) -- not product.  Run once per build to A/B the block-scaled MFMA form against the
non-scaled twin:   python3 tools/bench_gemm_fp8.py;  OMCHAT_LIB=ab_lib/f8_nonscaled/libomchat_hip.so python3 tools/bench_gemm_fp8.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omchat_amd import _lib

lib = _lib.lib()
P = lambda t: t.data_ptr() if t is not None else None
g = torch.Generator(device="cuda").manual_seed(1)
print("library:", _lib.LIB_PATH)
for name, M, N, K, epi, nw in (("gate|up S=33280 swiglu", 33280, 37888, 3584, _lib.EPI_SWIGLU, 3), ("qkv     S=33280", 33280, 4608, 3584, _lib.EPI_NONE, 8),
                               ("gate|up S=3584  swiglu", 3584, 37888, 3584, _lib.EPI_SWIGLU, 3), ("gate|up S=8704  swiglu", 8704, 37888, 3584, _lib.EPI_SWIGLU, 3)):
    A8 = (torch.randn(M, K, device="cuda", generator=g) * 2).to(torch.float8_e4m3fn)
    Ws = [(torch.randn(N, K, device="cuda", generator=g) * 2).to(torch.float8_e4m3fn) for _ in range(nw)]
    sa = torch.rand(M, device="cuda", generator=g) * 0.01 + 0.01
    sw = torch.rand(N, device="cuda", generator=g) * 0.01 + 0.01
    Nc = N // 2 if epi == _lib.EPI_SWIGLU else N
    C = torch.empty(M, Nc, device="cuda", dtype=torch.bfloat16)
    def run(i):
        _lib.check(lib.omchat_op_gemm_fp8(_lib.BF16, P(A8), P(sa), P(Ws[i % nw]), P(sw), P(C), Nc, M, N, K, None, None, 0, epi, None))
    for i in range(nw): run(i)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(2 * nw): run(i)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / (2 * nw))
    tf = 2.0 * M * N * K / best / 1e6
    print(f"{name:26s} {best:9.1f} us  {tf:7.1f} TF/s  = {tf / 5000:.3f} of the 5 PF fp8 peak ({tf / 2500:.3f} of 2.5 PF)", flush=True)
