"""Batch-1 decode GEMVs in isolation, 28 weight matrices in turn (cold weights): qkv (RMSNorm + 4608 x 3584 + bias), gate|up (RMSNorm + 37888 x 3584,
SwiGLU), down_proj (3584 x 18944 + residual), o_proj (3584 x 3584 + residual).  python tools/bench_gemv_b1.py   (OMCHAT_LIB selects another build)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omchat_amd import _lib
lib = _lib.lib()
P = lambda t: t.data_ptr() if t is not None else None
g = torch.Generator(device="cuda").manual_seed(1)
H, It, NL = 3584, 18944, 28
def timed(run, reps=5):
    run(NL); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(4 * NL); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / (4 * NL))
    return best
x = torch.randn(H, device="cuda", generator=g).bfloat16(); nw = torch.ones(H, device="cuda").bfloat16()
# gate|up
Ws = [(torch.randn(2 * It, H, device="cuda", generator=g) * 0.02).bfloat16() for _ in range(NL)]
y = torch.empty(It, device="cuda", dtype=torch.bfloat16)
t = timed(lambda n: [_lib.check(lib.omchat_op_gemv_norm(1, P(x), P(Ws[i % NL]), H, P(y), 2 * It, H, P(nw), 1e-6, None, _lib.EPI_SWIGLU, 0, None)) for i in range(n)])
print(f"gate|up + norm   {t:7.2f} us  ({2 * It * H * 2 / t / 1e6:.2f} TB/s)")
del Ws
# qkv
Ws = [(torch.randn(4608, H, device="cuda", generator=g) * 0.02).bfloat16() for _ in range(NL)]
bq = torch.randn(4608, device="cuda", generator=g).bfloat16(); yq = torch.empty(4608, device="cuda", dtype=torch.bfloat16)
t = timed(lambda n: [_lib.check(lib.omchat_op_gemv_norm(1, P(x), P(Ws[i % NL]), H, P(yq), 4608, H, P(nw), 1e-6, P(bq), _lib.EPI_NONE, 0, None)) for i in range(n)])
print(f"qkv + norm       {t:7.2f} us  ({4608 * H * 2 / t / 1e6:.2f} TB/s)")
del Ws
# down_proj (long K, residual in place)
Ws = [(torch.randn(H, It, device="cuda", generator=g) * 0.02).bfloat16() for _ in range(NL)]
xa = torch.randn(It, device="cuda", generator=g).bfloat16(); yd = torch.randn(H, device="cuda", generator=g).bfloat16()
t = timed(lambda n: [_lib.check(lib.omchat_op_gemv(1, P(xa), It, P(Ws[i % NL]), It, P(yd), H, 1, H, It, None, P(yd), H, _lib.EPI_RESID, 0, None)) for i in range(n)])
print(f"down_proj + res  {t:7.2f} us  ({H * It * 2 / t / 1e6:.2f} TB/s)")
del Ws
Ws = [(torch.randn(H, H, device="cuda", generator=g) * 0.02).bfloat16() for _ in range(NL)]
t = timed(lambda n: [_lib.check(lib.omchat_op_gemv(1, P(x), H, P(Ws[i % NL]), H, P(yd), H, 1, H, H, None, P(yd), H, _lib.EPI_RESID, 0, None)) for i in range(n)])
print(f"o_proj + res     {t:7.2f} us  ({H * H * 2 / t / 1e6:.2f} TB/s)")
