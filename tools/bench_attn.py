"""Attention micro-bench through the C ABI (not product)."""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omchat_amd import _lib
lib = _lib.lib()
p = lambda t: C.c_void_p(t.data_ptr())
it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
if len(sys.argv) > 2:
    lib.omchat_op_set_tuning(8, int(sys.argv[2]))      # 0 = first-generation 16x16x32 kernel
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
for name, b, S, Hq, Hkv, causal in [("vit", 3, 1025, 25, 25, 0), ("vit24", 24, 1025, 25, 25, 0), ("dec", 1, 3584, 28, 4, 1), ("dec_b4", 4, 3584, 28, 4, 1),
                                 ("dec_b16", 16, 3584, 28, 4, 1), ("long16k", 1, 16384, 28, 4, 1), ("long33k", 1, 33280, 28, 4, 1)]:
    if only and name not in only:
        continue
    q = torch.randn(b, S, Hq, 128, device="cuda").bfloat16(); k = torch.randn(b, Hkv, S, 128, device="cuda").bfloat16(); v = torch.randn_like(k)
    o = torch.empty_like(q)
    run = lambda: _lib.check(lib.omchat_op_attn_prefill(_lib.BF16, p(q), p(k), p(v), p(o), b, S, S, Hq, Hkv, None, causal, 0, 128 ** -0.5, None))
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / it
    fl = 4.0 * b * Hq * S * S * 128 * (0.5 if causal else 1.0)
    print(f"{name}: {us:8.1f} us  {fl/us/1e6:7.1f} TF", flush=True)
