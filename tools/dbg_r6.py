import sys, ctypes as C, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from gpu_util import DT, CODE, dev, rnd, randn, ptr, sync
from omchat_amd import _lib
lib = _lib.lib()
dt = "bf16"
for tile, M, N, K in [(2, 515, 3200, 128), (10, 515, 3200, 128), (1, 130, 520, 64)]:
    A = rnd(randn((M, K), 1), dt); W = rnd(randn((N, K), 2, 0.05), dt)
    bias = rnd(randn((N,), 3, 0.1), dt); ls = rnd(randn((N,), 4, 0.1) + 0.1, dt); resid = rnd(randn((M, N), 5), dt)
    dA, dW, db, dl, dr = dev(A, dt), dev(W, dt), dev(bias, dt), dev(ls, dt), dev(resid, dt)
    ld = (N + 47) // 48 + 1
    for epi, base in ((7, 2), (8, 0)):
        plain = torch.full((M, N), 77.0, dtype=DT[dt], device="cuda")
        _lib.check(lib.omchat_op_gemm_fused(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(plain), N, M, N, K, ptr(db), ptr(dl), ptr(dr), N, base, tile, None, None, 0, None, None))
        out = torch.full((M, N), 77.0, dtype=DT[dt], device="cuda")
        stats = torch.full((M, ld), -1.0, dtype=torch.float32, device="cuda")
        slot = C.c_int(0)
        _lib.check(lib.omchat_op_gemm_fused(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(out), N, M, N, K, ptr(db), ptr(dl), ptr(dr), N, epi, tile, None, ptr(stats), ld, C.byref(slot), None))
        sync()
        ne = (out != plain)
        print(tile, M, N, K, "epi", epi, "slot", slot.value, "mismatch", int(ne.sum()), "of", ne.numel())
        if ne.any():
            idx = ne.nonzero()[:8].tolist()
            print(" first:", idx, [(float(out[i, j]), float(plain[i, j])) for i, j in idx[:4]])
            print(" rows with mismatch:", sorted(set(ne.nonzero()[:, 0].tolist()))[:20], "cols:", sorted(set(ne.nonzero()[:, 1].tolist()))[:20])
print("---- which one is right")
tile, M, N, K = 2, 515, 3200, 128
A = rnd(randn((M, K), 1), dt); W = rnd(randn((N, K), 2, 0.05), dt)
bias = rnd(randn((N,), 3, 0.1), dt); ls = rnd(randn((N,), 4, 0.1) + 0.1, dt); resid = rnd(randn((M, N), 5), dt)
dA, dW, db, dl, dr = dev(A, dt), dev(W, dt), dev(bias, dt), dev(ls, dt), dev(resid, dt)
ref = rnd(resid + rnd(rnd(A @ W.t() + bias, dt) * ls, dt), dt)
for epi in (2, 7):
    out = torch.full((M, N), 77.0, dtype=DT[dt], device="cuda")
    stats = torch.full((M, 70), -1.0, dtype=torch.float32, device="cuda")
    _lib.check(lib.omchat_op_gemm_fused(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(out), N, M, N, K, ptr(db), ptr(dl), ptr(dr), N, epi, tile, None, ptr(stats), 70, None, None))
    sync()
    ne = out.float().cpu() != ref
    print("epi", epi, "mismatch vs fp32 restatement", int(ne.sum()), "by column mod 4:", [int(ne[:, r::4].sum()) for r in range(4)])
