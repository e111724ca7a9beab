"""Experiment: do per-XCD shares shorten the batch-1 gate|up GEMV?  Times omchat_op_gemv_norm (RMSNorm + gate|up + SwiGLU, 18944 outputs,
K = 3584) over 28 weight matrices in turn (cold weights, as in the model) for the equal-share loop form and for shares skewed by
blockIdx % 8 (tuning key 28), every rotation of the skew table (which label is which XCD is not fixed)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omchat_amd import _lib

lib = _lib.lib()
K, It, NL = 3584, 18944, 28
g = torch.Generator(device="cuda").manual_seed(1)
Ws = [(torch.randn(2 * It, K, device="cuda", generator=g) * 0.02).bfloat16() for _ in range(NL)]
x = torch.randn(K, device="cuda", generator=g).bfloat16()
nw = torch.ones(K, device="cuda").bfloat16()
y = torch.empty(It, device="cuda", dtype=torch.bfloat16)
P = lambda t: t.data_ptr()


def run(n):
    for i in range(n):
        _lib.check(lib.omchat_op_gemv_norm(1, P(x), P(Ws[i % NL]), K, P(y), 2 * It, K, P(nw), 1e-6, None, _lib.EPI_SWIGLU, 0, None))


def timed(reps=5):
    run(NL); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(4 * NL); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / (4 * NL))
    return best


def pack(d):
    assert sum(d) == 0
    v = 0
    for i, di in enumerate(d): v |= (di + 8) << (4 * i)
    return v - (1 << 32) if v >= (1 << 31) else v


ref = None
lib.omchat_op_set_tuning(28, 0)
t0 = timed(); y0 = y.clone()
print(f"equal shares                         {t0:7.2f} us per launch ({2 * It * K * 2 / t0 / 1e6:.2f} TB/s)")
base = [4, -4, 1, -2, 4, -2, 1, -2]
for scale, name in ((1, "measured skew"), (0.5, "half skew")):
    d0 = [int(round(v * scale)) for v in base]
    d0[0] -= sum(d0)
    for r in range(8):
        d = d0[-r:] + d0[:-r] if r else d0
        lib.omchat_op_set_tuning(28, pack(d))
        t = timed(3)
        same = bool(torch.equal(y, y0))
        print(f"{name:14s} rot {r} {str(d):34s} {t:7.2f} us  ({t / t0:.3f} x)  same bits {same}")
lib.omchat_op_set_tuning(28, 0)
print(f"equal shares again                   {timed():7.2f} us")
