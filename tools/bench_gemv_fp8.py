"""fp8 vs 16-bit batch-1 GEMV micro-bench through the C ABI (not product)."""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omchat_amd import _lib
lib = _lib.lib()
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
it = 50
for name, N, K, epi, ks in [("qkv", 4608, 3584, 0, 1), ("o", 3584, 3584, 5, 3), ("gate|up", 37888, 3584, 4, 1), ("down", 3584, 18944, 5, 8), ("lm_head", 152064, 3584, 0, 1)]:
    w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16(); x = torch.randn(K, device="cuda").bfloat16()
    w8 = torch.empty(N, K, dtype=torch.uint8, device="cuda"); sc = torch.empty(N, dtype=torch.float32, device="cuda")
    _lib.check(lib.omchat_op_quant_fp8(_lib.BF16, p(w), N, K, p(w8), p(sc), None))
    f32 = name == "lm_head"
    y = torch.empty(max(ks, 1) * N, dtype=torch.float32 if (epi == 5 or f32) else torch.bfloat16, device="cuda")
    # a second weight set defeats the 256 MB MALL between iterations for the small shapes
    ws = [w] + [w.clone() for _ in range(3 if N * K < 2e8 else 0)]
    w8s = [w8] + [w8.clone() for _ in range(3 if N * K < 2e8 else 0)]
    def run16(i): _lib.check(lib.omchat_op_gemv(_lib.BF16, p(x), K, p(ws[i % len(ws)]), K, p(y), N, 1, N, K, None, None, 0, epi, int(f32), None)) if epi != 5 else None
    def run8(i): _lib.check(lib.omchat_op_gemv_fp8(_lib.BF16, p(x), p(w8s[i % len(w8s)]), p(sc), p(y), N, K, None, None, epi, int(f32), ks, None))
    for fn, label, by in ((run16, "bf16", 2), (run8, "fp8", 1)):
        if label == "bf16" and epi == 5: continue
        for i in range(3): fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(it): fn(i)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / it
        print(f"{name:8s} {label:5s} {us:7.1f} us  {N*K*by/us/1e6:6.2f} TB/s", flush=True)
