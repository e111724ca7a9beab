"""Kernel-level A/B of the decode attention over the e4m3 KV cache (attention + merge launches): b sequences x 4 kv heads x L keys.
   python tools/bench_attn_decode_kv8.py [b] [L]     prints us per call for one wave per tile and for a wave walking 2 / 3 / 4 tiles
   (tuning key 47).  The caches rotate through enough copies that no call finds its lines in the 256 MB Infinity Cache."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omchat_amd import _lib

b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L = int(sys.argv[2]) if len(sys.argv) > 2 else 33000
lib = _lib.lib()
Hq, Hkv, cap = 28, 4, (L + 255) // 256 * 256
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(b, Hq, 128, device="cuda", generator=g).bfloat16()
mb = 2 * b * Hkv * L * (128 + 4) / 1e6
ncopy = max(2, int(600 / max(mb, 1e-3)) + 1)
sets = []
for c in range(ncopy):
    k8 = torch.randint(0, 0x78, (b, Hkv, cap, 128), device="cuda", generator=g, dtype=torch.uint8)
    k8 |= torch.randint(0, 2, k8.shape, device="cuda", generator=g, dtype=torch.uint8) << 7
    v8 = torch.randint(0, 0x78, (b, Hkv, cap, 128), device="cuda", generator=g, dtype=torch.uint8)
    ks = torch.rand(b, Hkv, cap, device="cuda", generator=g) * 0.01 + 0.001
    vs = torch.rand(b, Hkv, cap, device="cuda", generator=g) * 0.01 + 0.001
    sets.append((k8, v8, ks, vs))
out = torch.empty(b, Hq, 128, device="cuda", dtype=torch.bfloat16)
wsb = lib.omchat_op_attn_decode_ws(b, Hq, L)
ws = torch.empty(wsb // 4 + 4, dtype=torch.float32, device="cuda")
dl = torch.full((b,), L, dtype=torch.int32, device="cuda")
P = lambda t: t.data_ptr()


def run(n, only=None):
    for i in range(n):
        k8, v8, ks, vs = sets[(i % ncopy) if only is None else only]
        _lib.check(lib.omchat_op_attn_decode_kv8(1, P(q), P(k8), P(v8), P(ks), P(vs), P(out), b, Hq, Hkv, cap, L, P(dl), 128 ** -0.5, P(ws), wsb, None))


ref = None
for name, tpw in (("one wave per tile", 1), ("2 tiles per wave", 2), ("3 tiles per wave", 3), ("4 tiles per wave", 4), ("3 tiles x 4 waves, LDS fold", 11), ("launcher's choice", 0), ("one wave per tile again", 1)):
    lib.omchat_op_set_tuning(47, tpw)
    run(20); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record(); run(n); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / n
    run(1, only=0); torch.cuda.synchronize()
    o = out.float().clone()
    if ref is None: ref = o
    print(f"kv8 b {b} L {L} ({mb:.1f} MB, {ncopy} cache copies)  {name:28s} {us:7.2f} us per attention + merge   ({mb / us * 1e3:.0f} GB/s incl. merge)   "
          f"rel diff vs first {float((o - ref).norm() / ref.norm()):.3e}")
lib.omchat_op_set_tuning(47, 0)
