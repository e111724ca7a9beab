"""Kernel-level A/B of the batched decode attention (attention + merge launches) at the configs[2] shape: b sequences x 4 kv heads x L keys.
   python tools/bench_attn_decode.py [b] [L]     prints us per call for the register multi-tile form and the LDS-DMA ring form at several
   slot counts (tuning keys 25 / 26)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omchat_amd import _lib

b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = int(sys.argv[2]) if len(sys.argv) > 2 else 3700
cap_arg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lib = _lib.lib()
Hq, Hkv, cap = 28, 4, cap_arg or (L + 255) // 256 * 256
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(b, Hq, 128, device="cuda", generator=g).bfloat16()
k = torch.randn(b, Hkv, cap, 128, device="cuda", generator=g).bfloat16()
v = torch.randn(b, Hkv, cap, 128, device="cuda", generator=g).bfloat16()
out = torch.empty(b, Hq, 128, device="cuda", dtype=torch.bfloat16)
wsb = lib.omchat_op_attn_decode_ws(b, Hq, L)
ws = torch.empty(wsb // 4 + 4, dtype=torch.float32, device="cuda")
dl = torch.full((b,), L, dtype=torch.int32, device="cuda")
P = lambda t: t.data_ptr()
# a second set of K / V so that consecutive calls do not hit the 256 MB Infinity Cache on the same lines
k2, v2 = k.clone(), v.clone()
mb = 2 * b * Hkv * L * 256 / 1e6


def run(n):
    for i in range(n):
        kk, vv = (k, v) if i & 1 else (k2, v2)
        _lib.check(lib.omchat_op_attn_decode(1, P(q), P(kk), P(vv), P(out), b, Hq, Hkv, cap, L, P(dl), 128 ** -0.5, P(ws), wsb, None))


ref = None
tpw_force = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # tuning key 10: 0 = the launcher's own choice of form
lib.omchat_op_set_tuning(10, tpw_force)
for name, dma, slots, rot in (("register form", 0, 4, 0), ("dma ring, 4 slots x 2 stages", 1, 4 + 2 * 256, 0), ("dma ring, 2 slots x 2 stages", 1, 2 + 2 * 256, 0),
                              ("dma ring, 2 slots x 4 stages", 1, 2 + 4 * 256, 0), ("register form again", 0, 4, 0)):
    lib.omchat_op_set_tuning(25, dma); lib.omchat_op_set_tuning(26, slots); lib.omchat_op_set_tuning(27, rot)
    run(20); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(200); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 200
    o = out.float().clone()
    if ref is None: ref = o
    print(f"b {b} L {L} cap {cap} ({mb:.0f} MB)  {name:34s} {us:7.2f} us per attention + merge   ({mb / us / 1e3 * 1e3:.0f} GB/s incl. merge)   max |diff| vs first {float((o - ref).abs().max()):.3e}")
lib.omchat_op_set_tuning(25, 1); lib.omchat_op_set_tuning(26, 0); lib.omchat_op_set_tuning(27, 0); lib.omchat_op_set_tuning(10, 0)
