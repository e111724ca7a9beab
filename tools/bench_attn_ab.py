"""Prefill attention A/B through the C ABI (not product): tuning key <key> = each of <values>, interleaved in one process, outputs compared.

    python tools/bench_attn_ab.py 15 2,3 [iters] [shapes]
"""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omchat_amd import _lib
lib = _lib.lib()
p = lambda t: C.c_void_p(t.data_ptr())
key = int(sys.argv[1]); vals = [int(v) for v in sys.argv[2].split(",")]
it = int(sys.argv[3]) if len(sys.argv) > 3 else 20
only = sys.argv[4].split(",") if len(sys.argv) > 4 else None
SHAPES = [("dec2048", 1, 2048, 28, 4, 1), ("dec5000", 1, 5000, 28, 4, 1), ("vit", 3, 1025, 25, 25, 0), ("vit24", 24, 1025, 25, 25, 0), ("dec", 1, 3584, 28, 4, 1), ("dec_b4", 4, 3584, 28, 4, 1),
          ("dec_b16", 16, 3584, 28, 4, 1), ("c3_8704", 1, 8704, 28, 4, 1), ("long33k", 1, 33280, 28, 4, 1)]
for name, b, S, Hq, Hkv, causal in SHAPES:
    if only and name not in only:
        continue
    g = torch.Generator(device="cuda").manual_seed(1)
    q = torch.randn(b, S, Hq, 128, device="cuda", generator=g).bfloat16(); k = torch.randn(b, Hkv, S, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(b, Hkv, S, 128, device="cuda", generator=g).bfloat16()
    outs, best = {}, {x: 1e30 for x in vals}
    for rep in range(3):
        for x in vals:
            lib.omchat_op_set_tuning(key, x)
            o = torch.full_like(q, float("nan"))
            run = lambda: _lib.check(lib.omchat_op_attn_prefill(_lib.BF16, p(q), p(k), p(v), p(o), b, S, S, Hq, Hkv, None, causal, 0, 128 ** -0.5, None))
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(it): run()
            e1.record(); torch.cuda.synchronize()
            best[x] = min(best[x], e0.elapsed_time(e1) * 1e3 / it)
            outs[x] = o
    fl = 4.0 * b * Hq * S * S * 128 * (0.5 if causal else 1.0)
    same = all(torch.equal(outs[vals[0]], outs[x]) for x in vals[1:])
    print(name, "  ".join(f"key{key}={x}: {best[x]:8.1f} us {fl / best[x] / 1e6:7.1f} TF" for x in vals), "identical" if same else
          f"max diff {max((outs[vals[0]].float() - outs[x].float()).abs().max().item() for x in vals[1:]):.3e}", flush=True)
