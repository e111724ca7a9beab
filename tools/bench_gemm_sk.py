"""Stream-K tail on the multi-round prefill GEMMs (not product): omchat_op_gemm (data-parallel tiles only) against omchat_op_gemm_sk with the
   tail required, over several weight copies in turn (cold weights, as in the model).  python tools/bench_gemm_sk.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omchat_amd import _lib

lib = _lib.lib()
P = lambda t: t.data_ptr() if t is not None else None
g = torch.Generator(device="cuda").manual_seed(1)
wsb = lib.omchat_op_gemm_sk_ws()
ws = torch.zeros(wsb, dtype=torch.uint8, device="cuda")
SHAPES = [("prefill gate|up  M=3584 N=37888 K=3584 swiglu", 3584, 37888, 3584, _lib.EPI_SWIGLU, 6),
          ("prefill qkv      M=3584 N=4608  K=3584", 3584, 4608, 3584, _lib.EPI_NONE, 12),
          ("prefill down     M=3584 N=3584  K=18944 resid", 3584, 3584, 18944, _lib.EPI_RESID, 8),
          ("vit fc1          M=3075 N=12800 K=3200 gelu", 3075, 12800, 3200, _lib.EPI_GELU, 12),
          ("vit qkv          M=3075 N=9600  K=3200", 3075, 9600, 3200, _lib.EPI_NONE, 12),
          ("configs3 gate|up M=8704 N=37888 K=3584 swiglu", 8704, 37888, 3584, _lib.EPI_SWIGLU, 4)]
for name, M, N, K, epi, nw in SHAPES:
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).bfloat16()
    Ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16() for _ in range(nw)]
    Nc = N // 2 if epi == _lib.EPI_SWIGLU else N
    C = torch.empty(M, Nc, device="cuda", dtype=torch.bfloat16)
    R = (torch.randn(M, Nc, device="cuda", generator=g)).bfloat16() if epi == _lib.EPI_RESID else None
    outs = {}
    for mode in ("dp", "sk", "dp", "sk"):
        def run(i):
            if mode == "dp":
                _lib.check(lib.omchat_op_gemm(1, P(A), K, P(Ws[i % nw]), K, P(C), Nc, M, N, K, None, None, P(R), Nc, epi, 2, None))
            else:
                _lib.check(lib.omchat_op_gemm_sk(1, P(A), K, P(Ws[i % nw]), K, P(C), Nc, M, N, K, None, None, P(R), Nc, epi, 2, P(ws), wsb, 1, None))
        for i in range(nw): run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(3 * nw): run(i)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / (3 * nw)
        run(0); torch.cuda.synchronize(); outs[mode] = C.float().clone()
        tiles = -(-M // 256) * -(-N // 256)
        print(f"{name:52s} {mode}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF   tiles {tiles} = {tiles / 256:.2f} rounds", flush=True)
    d = (outs["dp"] - outs["sk"]).abs().max().item()
    print(f"    max |dp - sk| = {d:.3e}")
