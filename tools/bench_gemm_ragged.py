"""Ragged last row tile (M = 3075 against 3072) on the ViT shapes, per tile kernel (not product).  python tools/bench_gemm_ragged.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omchat_amd import _lib
lib = _lib.lib()
P = lambda t: t.data_ptr() if t is not None else None
g = torch.Generator(device="cuda").manual_seed(1)
for name, N, K, epi, tiles in (("fc2  N=3200 K=12800 ls+resid", 3200, 12800, _lib.EPI_LS_RESID, (7, 2)), ("proj N=3200 K=3200 ls+resid", 3200, 3200, _lib.EPI_LS_RESID, (9, 2)),
                               ("fc1  N=12800 K=3200 gelu", 12800, 3200, _lib.EPI_GELU, (2, 8)), ("qkv  N=9600 K=3200", 9600, 3200, _lib.EPI_NONE, (2, 8))):
    nw = 8
    Ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16() for _ in range(nw)]
    bias = torch.randn(N, device="cuda", generator=g).bfloat16(); ls = torch.randn(N, device="cuda", generator=g).bfloat16()
    for M in (3075, 3072):
        A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).bfloat16()
        R = torch.randn(M, N, device="cuda", generator=g).bfloat16()
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for tile in tiles:
            def run(i):
                _lib.check(lib.omchat_op_gemm(1, P(A), K, P(Ws[i % nw]), K, P(C), N, M, N, K, P(bias), P(ls) if epi == _lib.EPI_LS_RESID else None,
                                              P(R) if epi == _lib.EPI_LS_RESID else None, N, epi, tile, None))
            for i in range(nw): run(i)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(4 * nw): run(i)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1000 / (4 * nw))
            print(f"{name:32s} M={M} tile {tile}: {best:7.1f} us  {2.0 * M * N * K / best / 1e6:7.1f} TF", flush=True)
