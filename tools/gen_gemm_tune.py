#!/usr/bin/env python3
"""Measure the GEMM tile choice of every GEMM class the hot path launches in bench.py (OmChat-13B: the 3-tile sample, the 32-sample batch
and the 32-frame clip; OmChat-2.1-8B: the 8-picture sample of configs[3]; TP 1 / 2 / 4 / 8 each) on this GPU and write omchat_amd/gemm_tune_gfx950.txt, which the binding loads with the library so
that first-use tuning never runs on a live request or inside a timed / multi-rank region.

    python tools/gen_gemm_tune.py          # on the MI355X box (gpurun); commit the resulting file
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from omchat_amd import _lib
from omchat_amd.config import omchat13b, omchat8b_21
from omchat_amd.tp import local_dims

NONE, GELU, LS_RESID, RESID, SWIGLU = 0, 1, 2, 3, 4


def chunks(M, tp, N):
    """row chunks of a row-parallel projection (model.hip: gemm_allreduce): up to 4, while a chunk still holds >= 256 tiles of 256^2"""
    if tp == 1:
        return [M]
    nch = 4 if M >= 3 * 1024 else (2 if M >= 1024 else 1)
    tiles = -(-M // 256) * -(-N // 256)
    while nch > 1 and tiles // nch < 256:
        nch //= 2
    if nch == 1:
        return [M]
    per = -(-(-(-M // nch)) // 256) * 256
    out, r0 = [], 0
    while r0 < M:
        out.append(min(per, M - r0)); r0 += per
    return out


def classes(cfg, tp, tiles, rows):
    d = local_dims(cfg, 0, tp)
    C, H = cfg.vision["hidden_size"], cfg.text["hidden_size"]
    Cq, I = d["v_heads"] * cfg.vision.get("head_dim", 128), d["v_mlp"]
    qd, kvd, It = d["t_heads"] * 128, d["t_kv_heads"] * 128, d["t_mlp"]
    qkvd = qd + 2 * kvd
    M = tiles * 1025
    out = [(tiles * 1024, C, 640, NONE, C), (M, 3 * Cq, C, NONE, 3 * Cq), (M, I, C, GELU, I),
           (tiles * 1024, H, C, GELU, H), (tiles * 1024, H, H, NONE, H),
           (rows, qkvd, H, NONE, qkvd), (rows, 2 * It, H, SWIGLU, It)]
    for m in sorted(set(chunks(M, tp, C))):
        out += [(m, C, Cq, LS_RESID, C), (m, C, I, LS_RESID, C)]
    for m in sorted(set(chunks(rows, tp, H))):
        out += [(m, H, qd, RESID, H), (m, H, It, RESID, H)]
    return out


def main():
    lib = _lib.lib()
    todo = []
    # (config, tiles per tower launch, prefill rows): configs[1], configs[2] (24-tile tower chunks), configs[4] (24 + 8 tile chunks,
    # 33 280 rows), configs[3] (InternViT-300M, 8 tiles, 8704 rows)
    work = [(omchat13b(), 3, 3584), (omchat13b(), 24, 32 * 3584), (omchat13b(), 8, 32 * 1024 + 512), (omchat13b(), 24, 32 * 1024 + 512),
            (omchat8b_21(), 8, 8 * 1024 + 512)]
    for tp in (1, 2, 4, 8):
        for cfg, tiles, rows in work:
            for c in classes(cfg, tp, tiles, rows):
                if c not in todo:
                    todo.append(c)
    print(f"{len(todo)} GEMM classes")
    for dt, code in ((torch.bfloat16, _lib.BF16), (torch.float16, _lib.F16)):
        for M, N, K, epi, ldc in todo:
            Mt = min(M, 8192)                       # the tuner times at most 8192 rows; the class key keeps ceil(M / 256) of the real M
            A = torch.randn(Mt, K, device="cuda").to(dt)
            W = (torch.randn(N, K, device="cuda") * 0.02).to(dt)
            Cb = torch.empty(Mt, ldc, device="cuda", dtype=dt)
            bias = torch.zeros(N, device="cuda", dtype=dt)
            # the library keys on ceil(M / 256): pass the real M only when it fits, otherwise a row count in the same class is impossible,
            # so large-M classes are registered through a problem of the real size with A / C re-used modulo (not needed: M <= 114688 fits)
            if M > Mt:
                A = torch.randn(M, K, device="cuda").to(dt)
                Cb = torch.empty(M, ldc, device="cuda", dtype=dt)
            R = Cb if epi in (LS_RESID, RESID) else None
            _lib.check(lib.omchat_op_gemm(code, _lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(Cb), ldc, M, N, K, _lib.ptr(bias) if epi != SWIGLU else None,
                                          _lib.ptr(bias) if epi == LS_RESID else None, _lib.ptr(R), ldc, epi, 0, None))
            torch.cuda.synchronize()
            del A, W, Cb
        torch.cuda.empty_cache()
    out = os.path.join(ROOT, "omchat_amd", "gemm_tune_gfx950.txt")
    n = lib.omchat_gemm_tune_dump(out.encode())
    dst = os.path.join(ROOT, "gpurun_out", "gemm_tune_gfx950.txt")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    lib.omchat_gemm_tune_dump(dst.encode())
    print(f"wrote {n} entries ({lib.omchat_gemm_tune_runs()} measured here) -> {out} and {dst}")


if __name__ == "__main__":
    main()
