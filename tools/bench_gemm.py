"""GEMM micro-bench through the C ABI (not product): TFLOP/s per tile config on the hot-path shapes."""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omchat_amd import _lib
lib = _lib.lib()
shapes = [("vit_qkv", 3075, 9600, 3200, 0), ("vit_proj", 3075, 3200, 3200, 2), ("vit_fc1", 3075, 12800, 3200, 1), ("vit_fc2", 3075, 3200, 12800, 2),
          ("dec_qkv", 3584, 4608, 3584, 0), ("dec_o", 3584, 3584, 3584, 3), ("dec_gateup", 3584, 37888, 3584, 4), ("dec_down", 3584, 3584, 18944, 3),
          ("sq4096", 4096, 4096, 4096, 0), ("sq8192", 8192, 8192, 8192, 0)]
tiles = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2, 6, 3, 1]
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0      # operand magnitude: uniform(-scale, scale); power (and so the clock) depends on it
p = lambda t: C.c_void_p(t.data_ptr())
wsb = lib.omchat_op_gemm_sk_ws()
ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
for name, M, N, K, epi in shapes:
    A = ((torch.rand(M, K, device="cuda") * 2 - 1) * scale).bfloat16(); W = ((torch.rand(N, K, device="cuda") * 2 - 1) * scale).bfloat16()
    bias = torch.zeros(N, device="cuda", dtype=torch.bfloat16); ls = torch.ones(N, device="cuda", dtype=torch.bfloat16)
    No = N // 2 if epi == 4 else N
    out = torch.zeros(M, No, device="cuda", dtype=torch.bfloat16); res = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    line = f"{name:10s} M={M:5d} N={N:5d} K={K:5d} epi={epi}:"
    for tile in tiles:
        if epi == 4 and tile == 3:
            continue
        lib.omchat_op_set_tuning(0, 0)
        if tile >= 100:          # 1xx: 256^2 staggered kernel with start skew xx
            lib.omchat_op_set_tuning(0, tile - 100); tile_code = 2
        else:
            tile_code = tile
        def run():
            if tile == 20:      # 256^2 staggered kernel + stream-K tail
                _lib.check(lib.omchat_op_gemm_sk(_lib.BF16, p(A), K, p(W), K, p(out), No, M, N, K, p(bias) if epi in (1, 2) else None, p(ls), p(res), N, epi, 2, p(ws), wsb, 1, None))
            else:
                _lib.check(lib.omchat_op_gemm(_lib.BF16, p(A), K, p(W), K, p(out), No, M, N, K, p(bias) if epi in (1, 2) else None, p(ls), p(res), N, epi, tile_code, None))
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 20
        e0.record()
        for _ in range(it): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / it
        line += f"  t{tile}: {us:7.1f}us {2.0*M*N*K/us/1e6:6.0f}TF"
    print(line, flush=True)
