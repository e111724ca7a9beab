"""GPU parity: every HIP kernel through the C ABI against a plain fp32 restatement on the same (16-bit rounded) inputs."""
import ctypes as C
import math
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
from gpu_util import DT, CODE, TOL, dev, rnd, rel, ptr, sync, randn
from omchat_amd import _lib, synth
import oracle

DTS = ["bf16", "f16"]


@pytest.mark.parametrize("dt", DTS)
def test_fill_uniform_bit_exact_with_host_generator(gpu_lib, dt):
    for name, n, std, off in [("a.weight", 100003, 0.02, 0.0), ("b.norm", 4099, 0.05, 1.0)]:
        out = torch.empty(n, dtype=DT[dt], device="cuda")
        key = synth.fnv1a64(name) ^ 7
        scale = float(np.float32(std * np.sqrt(3.0)))
        _lib.check(gpu_lib.omchat_op_fill_uniform(CODE[dt], ptr(out), n, key, scale, off, None))
        sync()
        ref = synth.uniform(name, (n,), 7, std, off)
        assert np.array_equal(out.float().cpu().numpy(), ref)


def _gemm_ref(A, W, bias, ls, resid, epi, dt):
    y = A @ W.t()
    if epi == _lib.EPI_SWIGLU:
        N = W.shape[0]
        blocks = y.reshape(y.shape[0], N // 32, 2, 16)
        g, u = rnd(blocks[:, :, 0], dt), rnd(blocks[:, :, 1], dt)
        return (rnd(F.silu(g), dt) * u).reshape(y.shape[0], N // 2)
    if bias is not None:
        y = y + bias
    y = rnd(y, dt)
    if epi == _lib.EPI_GELU:
        y = F.gelu(y)
    elif epi == _lib.EPI_LS_RESID:
        y = resid + rnd(y * ls, dt)
    elif epi == _lib.EPI_RESID:
        y = resid + y
    return y


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 10, 11])
@pytest.mark.parametrize("M,N,K", [(300, 224, 128), (1025, 416, 320), (64, 3200, 640), (515, 512, 1024)])
def test_gemm_epilogues(gpu_lib, dt, tile, M, N, K):
    A = rnd(randn((M, K), 1), dt); W = rnd(randn((N, K), 2, 0.05), dt)
    bias = rnd(randn((N,), 3, 0.1), dt); ls = rnd(randn((N,), 4, 0.1) + 0.1, dt); resid = rnd(randn((M, N), 5), dt)
    for epi in (_lib.EPI_NONE, _lib.EPI_GELU, _lib.EPI_LS_RESID, _lib.EPI_RESID, _lib.EPI_SWIGLU):
        if epi == _lib.EPI_SWIGLU and (N % 32 or tile in (3, 10, 11)):
            continue
        No = N // 2 if epi == _lib.EPI_SWIGLU else N
        use_bias = epi != _lib.EPI_SWIGLU and epi != _lib.EPI_RESID
        dA, dW, db, dl, dr = dev(A, dt), dev(W, dt), dev(bias, dt), dev(ls, dt), dev(resid, dt)
        out = torch.full((M, No), float("nan"), dtype=DT[dt], device="cuda")
        _lib.check(gpu_lib.omchat_op_gemm(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(out), No, M, N, K, ptr(db) if use_bias else None,
                                          ptr(dl), ptr(dr), N, epi, tile, None))
        sync()
        ref = _gemm_ref(A, W, bias if use_bias else None, ls, resid, epi, dt)
        assert torch.isfinite(out.float()).all(), (epi, "non-finite / unwritten outputs")
        assert rel(out, ref) < TOL[dt], (epi, tile, rel(out, ref))


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,N,K", [(3584, 1024, 3584), (3075, 768, 12800), (512, 512, 64), (700, 900, 128)])
def test_gemm_staggered_kernel_race_screen(gpu_lib, dt, M, N, K):
    """the 4-phase staggered 256^2 kernel: production-size K loops, repeated launches must be bit-identical and correct"""
    A = rnd(randn((M, K), 11, 0.5), dt); W = rnd(randn((N, K), 12, 0.05), dt)
    dA, dW = dev(A, dt), dev(W, dt)
    ref = A @ W.t()
    outs = []
    for it in range(4):
        out = torch.full((M, N), float("nan"), dtype=DT[dt], device="cuda")
        _lib.check(gpu_lib.omchat_op_gemm(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(out), N, M, N, K, None, None, None, 0, _lib.EPI_NONE, 2, None))
        sync()
        outs.append(out)
    assert rel(outs[0], ref) < TOL[dt], rel(outs[0], ref)
    # element-wise: no wrong tile may hide inside a small Frobenius error
    err = (outs[0].float().cpu() - ref).abs()
    assert float(err.max()) < (0.25 if dt == "bf16" else 0.05) * float(ref.abs().max())
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    old = torch.empty_like(outs[0])
    _lib.check(gpu_lib.omchat_op_gemm(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(old), N, M, N, K, None, None, None, 0, _lib.EPI_NONE, 6, None))
    sync()
    assert torch.equal(old, outs[0])          # same fp32 accumulation order per element as the one-barrier kernel


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,N,K,epi", [(3075, 3200, 3200, _lib.EPI_LS_RESID), (3584, 3584, 3584, _lib.EPI_RESID), (3075, 12800, 3200, _lib.EPI_GELU),
                                       (1025, 3200, 12800, _lib.EPI_NONE), (3584, 6400, 1024, _lib.EPI_SWIGLU), (700, 3000, 2048, _lib.EPI_NONE)])
def test_gemm_stream_k_tail(gpu_lib, dt, M, N, K, epi):
    """stream-K tail (partial slabs + flags between workgroups) == data-parallel result, bit for bit, on repeated launches"""
    A = rnd(randn((M, K), 21, 0.5), dt); W = rnd(randn((N, K), 22, 0.05), dt)
    bias = rnd(randn((N,), 23, 0.1), dt); ls = rnd(randn((N,), 24, 0.1) + 0.1, dt); resid = rnd(randn((M, N), 25), dt)
    dA, dW, db, dl, dr = dev(A, dt), dev(W, dt), dev(bias, dt), dev(ls, dt), dev(resid, dt)
    No = N // 2 if epi == _lib.EPI_SWIGLU else N
    use_bias = epi in (_lib.EPI_GELU, _lib.EPI_LS_RESID)
    wsb = gpu_lib.omchat_op_gemm_sk_ws()
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    base = torch.full((M, No), float("nan"), dtype=DT[dt], device="cuda")
    _lib.check(gpu_lib.omchat_op_gemm(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(base), No, M, N, K, ptr(db) if use_bias else None, ptr(dl), ptr(dr), N, epi, 2, None))
    sync()
    ref = _gemm_ref(A, W, bias if use_bias else None, ls, resid, epi, dt)
    assert rel(base, ref) < TOL[dt]
    for it in range(5):
        out = torch.full((M, No), float("nan"), dtype=DT[dt], device="cuda")
        _lib.check(gpu_lib.omchat_op_gemm_sk(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(out), No, M, N, K, ptr(db) if use_bias else None, ptr(dl), ptr(dr), N,
                                             epi, 2, ptr(ws), wsb, 1, None))
        sync()
        assert torch.isfinite(out.float()).all()
        assert rel(out, ref) < TOL[dt], (it, rel(out, ref))
        # the split accumulation adds slabs in a different association than one long K loop: equal up to fp32 rounding
        assert rel(out, base) < 2e-3


@pytest.mark.parametrize("dt", DTS)
def test_gemm_identity_asymmetric(gpu_lib, dt):
    """A = I against an asymmetric W catches a transposed C write (cdna_hip_programming.md §3)."""
    K = 256
    A = torch.eye(K); W = rnd(randn((192, K), 9), dt)
    out = torch.empty(K, 192, dtype=DT[dt], device="cuda")
    dA, dW = dev(A, dt), dev(W, dt)
    _lib.check(gpu_lib.omchat_op_gemm(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(out), 192, K, 192, K, None, None, None, 0, 0, 0, None))
    sync()
    assert torch.equal(out.float().cpu(), W.t().contiguous())


@pytest.mark.parametrize("dt", DTS)
def test_gemm_in_place_residual(gpu_lib, dt):
    M, N, K = 200, 256, 128
    A = rnd(randn((M, K), 1), dt); W = rnd(randn((N, K), 2, 0.05), dt); x = rnd(randn((M, N), 3), dt)
    dA, dW, dx = dev(A, dt), dev(W, dt), dev(x, dt)
    _lib.check(gpu_lib.omchat_op_gemm(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(dx), N, M, N, K, None, None, ptr(dx), N, _lib.EPI_RESID, 0, None))
    sync()
    assert rel(dx, x + rnd(A @ W.t(), dt)) < TOL[dt]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("form", ["auto", "mfma"])
@pytest.mark.parametrize("b,N,K", [(1, 100, 256), (5, 4608, 3584), (16, 320, 512), (1, 3584, 18944), (3, 160, 64), (1, 37888, 3584), (1, 96, 2368),
                                      (17, 320, 512), (32, 4608, 3584), (24, 160, 18944)])       # two batch tiles per weight fragment
def test_gemv(gpu_lib, dt, form, b, N, K):
    """b == 1 takes the whole-row streaming (v_dot2) form unless the MFMA form is forced; both must agree with the reference"""
    gpu_lib.omchat_op_set_tuning(1, 1 if form == "mfma" else 0)
    try:
        _gemv_case(gpu_lib, dt, b, N, K)
    finally:
        gpu_lib.omchat_op_set_tuning(1, 0)


def _gemv_case(gpu_lib, dt, b, N, K):
    X = rnd(randn((b, K), 1), dt); W = rnd(randn((N, K), 2, 0.03), dt)
    bias = rnd(randn((N,), 3, 0.1), dt); resid = rnd(randn((b, N), 4), dt)
    dX, dW, db, dr = dev(X, dt), dev(W, dt), dev(bias, dt), dev(resid, dt)
    y = X @ W.t()
    # plain + bias
    out = torch.full((b, N), float("nan"), dtype=DT[dt], device="cuda")
    _lib.check(gpu_lib.omchat_op_gemv(CODE[dt], ptr(dX), K, ptr(dW), K, ptr(out), N, b, N, K, ptr(db), None, 0, _lib.EPI_NONE, 0, None))
    sync(); assert rel(out, rnd(y + bias, dt)) < TOL[dt]
    # fp32 logits
    outf = torch.full((b, N), float("nan"), dtype=torch.float32, device="cuda")
    _lib.check(gpu_lib.omchat_op_gemv(CODE[dt], ptr(dX), K, ptr(dW), K, ptr(outf), N, b, N, K, None, None, 0, _lib.EPI_NONE, 1, None))
    sync(); assert rel(outf, y) < 1e-4
    # residual
    out2 = torch.full((b, N), float("nan"), dtype=DT[dt], device="cuda")
    _lib.check(gpu_lib.omchat_op_gemv(CODE[dt], ptr(dX), K, ptr(dW), K, ptr(out2), N, b, N, K, None, ptr(dr), N, _lib.EPI_RESID, 0, None))
    sync(); assert rel(out2, resid + rnd(y, dt)) < TOL[dt]
    if N % 32 == 0:
        out3 = torch.full((b, N // 2), float("nan"), dtype=DT[dt], device="cuda")
        _lib.check(gpu_lib.omchat_op_gemv(CODE[dt], ptr(dX), K, ptr(dW), K, ptr(out3), N // 2, b, N, K, None, None, 0, _lib.EPI_SWIGLU, 0, None))
        sync(); assert rel(out3, _gemm_ref(X, W, None, None, None, _lib.EPI_SWIGLU, dt)) < TOL[dt]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("H", [256, 3200, 3584])
def test_rmsnorm(gpu_lib, dt, H):
    x = rnd(randn((37, H), 1, 2.0), dt); w = rnd(randn((H,), 2, 0.1) + 1.0, dt)
    dx, dw = dev(x, dt), dev(w, dt)
    out = torch.empty_like(dx)
    _lib.check(gpu_lib.omchat_op_rmsnorm(CODE[dt], ptr(dx), ptr(dw), ptr(out), 37, H, 1e-6, None))
    sync()
    ref = oracle.rms_norm(x.to(DT[dt]), w.to(DT[dt]), 1e-6).float()       # the reference's own rounding sequence
    assert rel(out, ref) < 2e-3
    # two-rounding semantics reproduced exactly on the vast majority of elements
    assert (out.float().cpu() == ref).float().mean() > 0.98


@pytest.mark.parametrize("dt", DTS)
def test_vit_qknorm(gpu_lib, dt):
    rows, C = 19, 384
    qkv = rnd(randn((rows, 3 * C), 1), dt); wq = rnd(randn((C,), 2, 0.1) + 1, dt); wk = rnd(randn((C,), 3, 0.1) + 1, dt)
    d = dev(qkv, dt); dwq = dev(wq, dt); dwk = dev(wk, dt)          # keep device tensors alive across the async launch
    scale = 128 ** -0.5
    _lib.check(gpu_lib.omchat_op_vit_qknorm(CODE[dt], ptr(d), 3 * C, ptr(dwq), ptr(dwk), rows, C, C, 1e-6, scale, None))
    sync()
    T = DT[dt]
    q = (oracle.rms_norm(qkv[:, :C].to(T), wq.to(T), 1e-6) * scale).float()
    k = oracle.rms_norm(qkv[:, C:2 * C].to(T), wk.to(T), 1e-6).float()
    assert rel(d[:, :C], q) < 3e-3 and rel(d[:, C:2 * C], k) < 3e-3
    assert torch.equal(d[:, 2 * C:].float().cpu(), qkv[:, 2 * C:])


def _attn_ref(q, k, v, scale, causal, q_pos0, kv_len):
    """q [b,Sq,Hq,D], k/v [b,Hkv,Skv,D] fp32"""
    b, Sq, Hq, D = q.shape
    Hkv, Skv = k.shape[1], k.shape[2]
    rep = Hq // Hkv
    kk = k[:, :, None].expand(b, Hkv, rep, Skv, D).reshape(b, Hq, Skv, D)
    vv = v[:, :, None].expand(b, Hkv, rep, Skv, D).reshape(b, Hq, Skv, D)
    s = torch.einsum("bqhd,bhkd->bhqk", q, kk) * scale
    kpos = torch.arange(Skv)[None, None, None, :]
    allowed = kpos < torch.tensor(kv_len)[:, None, None, None]
    if causal:
        allowed = allowed & (kpos <= (torch.arange(Sq)[None, None, :, None] + q_pos0))
    s = s.masked_fill(~allowed, float("-inf"))
    p = torch.softmax(s, dim=-1)
    return torch.einsum("bhqk,bhkd->bqhd", p, vv)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("b,Sq,Skv,Hq,Hkv,causal,lens", [
    (2, 1025, 1025, 3, 3, 0, None),          # ViT shape: ragged 1024+1 tail
    (1, 300, 300, 7, 1, 1, None),            # GQA group 7, causal
    (2, 200, 200, 4, 2, 1, [200, 77]),       # right-padded batch
    (1, 17, 17, 2, 2, 0, None),
    (1, 129, 129, 2, 1, 1, None),
])
def test_attn_prefill(gpu_lib, dt, b, Sq, Skv, Hq, Hkv, causal, lens):
    q = rnd(randn((b, Sq, Hq, 128), 1), dt); k = rnd(randn((b, Hkv, Skv, 128), 2), dt); v = rnd(randn((b, Hkv, Skv, 128), 3), dt)
    dq, dk, dv = dev(q, dt), dev(k, dt), dev(v, dt)
    out = torch.full((b, Sq, Hq, 128), float("nan"), dtype=DT[dt], device="cuda")
    dl = None if lens is None else torch.tensor(lens, dtype=torch.int32, device="cuda")
    scale = 128 ** -0.5
    _lib.check(gpu_lib.omchat_op_attn_prefill(CODE[dt], ptr(dq), ptr(dk), ptr(dv), ptr(out), b, Sq, Skv, Hq, Hkv, ptr(dl), causal, 0, scale, None))
    sync()
    ref = _attn_ref(q, k, v, scale, causal, 0, lens or [Skv] * b)
    for i in range(b):
        n = Sq if lens is None else lens[i]
        assert torch.isfinite(out[i, :n].float()).all()
        assert rel(out[i, :n], ref[i, :n]) < TOL[dt], rel(out[i, :n], ref[i, :n])


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("b,Sq,Skv,Hq,Hkv,causal,lens", [
    (1, 257, 257, 28, 4, 1, None),           # the Qwen2-7B head pattern: 7 query heads per kv head in one workgroup
    (2, 333, 333, 8, 1, 1, [333, 130]),      # group of 8 = the widest workgroup
    (1, 96, 96, 6, 2, 0, None),              # GQA without the causal mask
    (1, 70, 70, 2, 2, 1, None),              # MHA + causal
    (1, 64, 64, 5, 5, 0, None),              # exactly one kv tile
])
def test_attn_prefill_second_generation_shapes(gpu_lib, dt, b, Sq, Skv, Hq, Hkv, causal, lens):
    """more head groupings for the 32x32x16 kernel, and the first-generation kernel (tuning key 8 = 0) on the same inputs"""
    q = rnd(randn((b, Sq, Hq, 128), 1), dt); k = rnd(randn((b, Hkv, Skv, 128), 2), dt); v = rnd(randn((b, Hkv, Skv, 128), 3), dt)
    dq, dk, dv = dev(q, dt), dev(k, dt), dev(v, dt)
    dl = None if lens is None else torch.tensor(lens, dtype=torch.int32, device="cuda")
    scale = 128 ** -0.5
    ref = _attn_ref(q, k, v, scale, causal, 0, lens or [Skv] * b)
    try:
        for gen in (1, 0):
            gpu_lib.omchat_op_set_tuning(8, gen)
            out = torch.full((b, Sq, Hq, 128), float("nan"), dtype=DT[dt], device="cuda")
            _lib.check(gpu_lib.omchat_op_attn_prefill(CODE[dt], ptr(dq), ptr(dk), ptr(dv), ptr(out), b, Sq, Skv, Hq, Hkv, ptr(dl), causal, 0, scale, None))
            sync()
            for i in range(b):
                n = Sq if lens is None else lens[i]
                assert torch.isfinite(out[i, :n].float()).all()
                assert rel(out[i, :n], ref[i, :n]) < TOL[dt], (gen, rel(out[i, :n], ref[i, :n]))
    finally:
        gpu_lib.omchat_op_set_tuning(8, 1)


@pytest.mark.parametrize("dt", DTS)
def test_attn_prefill_softmax_spike(gpu_lib, dt):
    """forces the running max to jump late (online-softmax rescale path, cdna guide rule 26)"""
    Sq = Skv = 256
    q = rnd(randn((1, Sq, 1, 128), 1, 0.3), dt); k = rnd(randn((1, 1, Skv, 128), 2, 0.3), dt); v = rnd(randn((1, 1, Skv, 128), 3), dt)
    k[0, 0, 200] = q[0, 5, 0] * 40.0                # key 200 spikes for query 5 in the 4th kv tile
    k = rnd(k, dt)
    out = torch.empty((1, Sq, 1, 128), dtype=DT[dt], device="cuda")
    dq, dk, dv = dev(q, dt), dev(k, dt), dev(v, dt)
    _lib.check(gpu_lib.omchat_op_attn_prefill(CODE[dt], ptr(dq), ptr(dk), ptr(dv), ptr(out), 1, Sq, Skv, 1, 1, None, 0, 0, 1.0, None))
    sync()
    ref = _attn_ref(q, k, v, 1.0, 0, 0, [Skv])
    assert rel(out, ref) < TOL[dt]
    assert rel(out[0, 5], ref[0, 5]) < TOL[dt]


@pytest.mark.parametrize("dt", DTS)
def test_mha_fwd_packed_qkv(gpu_lib, dt):
    """the reference's one native seam: FlashAttention.forward(qkv[B,S,3,H,D]) (flash_attention.py:30-75)"""
    B, S, H = 2, 65, 3
    qkv = rnd(randn((B, S, 3, H, 128), 4), dt)
    out = torch.empty((B, S, H, 128), dtype=DT[dt], device="cuda")
    dqkv = dev(qkv, dt)
    _lib.check(gpu_lib.omchat_mha_fwd(ptr(dqkv), B, S, H, 0.0, 0, ptr(out), CODE[dt], None))
    sync()
    q, k, v = qkv.unbind(2)
    ref = _attn_ref(q, k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), 128 ** -0.5, 0, 0, [S] * B)
    assert rel(out, ref) < TOL[dt]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("b,Hq,Hkv,cap,lens", [(1, 7, 1, 64, [1]), (1, 28, 4, 800, [700]), (3, 4, 2, 256, [63, 64, 65]), (2, 7, 1, 4096, [3585, 17]),
                                                  (1, 28, 4, 8832, [8801]),            # 138 splits: the merge's 65..256-split path (configs[3] context)
                                                  (2, 7, 1, 16384, [16384, 4161]),     # 256 splits exactly, and 66 in the same launch
                                                  (1, 7, 1, 33280, [33280]),           # 520 splits (configs[4] context): the 1024-thread form, two batches
                                                  (2, 4, 2, 66000, [65537, 300])])     # 1025 splits: the general path; 5 in the same launch
def test_attn_decode(gpu_lib, dt, b, Hq, Hkv, cap, lens):
    q = rnd(randn((b, Hq, 128), 1), dt); k = rnd(randn((b, Hkv, cap, 128), 2), dt); v = rnd(randn((b, Hkv, cap, 128), 3), dt)
    for i, n in enumerate(lens):                      # poison the unused tail: must never leak
        k[i, :, n:] = float("nan"); v[i, :, n:] = float("nan")
    dq, dk, dv = dev(q, dt), dev(k, dt), dev(v, dt)
    out = torch.full((b, Hq, 128), float("nan"), dtype=DT[dt], device="cuda")
    L = max(lens)
    wsb = gpu_lib.omchat_op_attn_decode_ws(b, Hq, L)
    ws = torch.empty(wsb // 4 + 4, dtype=torch.float32, device="cuda")
    dl = torch.tensor(lens, dtype=torch.int32, device="cuda")
    scale = 128 ** -0.5
    kk = torch.nan_to_num(k); vv = torch.nan_to_num(v)
    ref = _attn_ref(q[:, None], kk, vv, scale, 0, 0, lens)[:, 0]
    done_auto = 0
    try:
        for tpw in (0, 1, 2, 4, 0, 0):                # key tiles per wave: automatic, then each forced split size (running max / sum inside a wave);
            # the last two passes: the split-KV merge with 2 column groups below 256 partials / without column groups (tuning key 21)
            gpu_lib.omchat_op_set_tuning(21, 1)
            if tpw == 0 and done_auto:
                gpu_lib.omchat_op_set_tuning(21, 2 if done_auto == 1 else 0)
            done_auto += tpw == 0
            gpu_lib.omchat_op_set_tuning(10, tpw)
            out.fill_(float("nan")); ws.fill_(float("nan"))
            _lib.check(gpu_lib.omchat_op_attn_decode(CODE[dt], ptr(dq), ptr(dk), ptr(dv), ptr(out), b, Hq, Hkv, cap, L, ptr(dl), scale, ptr(ws), wsb, None))
            sync()
            assert torch.isfinite(out.float()).all(), tpw
            assert rel(out, ref) < TOL[dt], (tpw, rel(out, ref))
    finally:
        gpu_lib.omchat_op_set_tuning(10, 0)
        gpu_lib.omchat_op_set_tuning(21, 1)


@pytest.mark.parametrize("dt", DTS)
def test_rope_kv(gpu_lib, dt):
    b, S, Hq, Hkv, cap, pos0 = 2, 9, 4, 2, 32, 5
    qkv = rnd(randn((b * S, (Hq + 2 * Hkv) * 128), 1), dt)
    d = dev(qkv, dt)
    kc = torch.zeros(b, Hkv, cap, 128, dtype=DT[dt], device="cuda"); vc = torch.zeros_like(kc)
    _lib.check(gpu_lib.omchat_op_rope_kv(CODE[dt], ptr(d), b, S, Hq, Hkv, pos0, 1e6, ptr(kc), ptr(vc), cap, None))
    sync()
    T = DT[dt]
    x = qkv.to(T).view(b, S, Hq + 2 * Hkv, 128)
    q = x[:, :, :Hq].transpose(1, 2); k = x[:, :, Hq:Hq + Hkv].transpose(1, 2); v = x[:, :, Hq + Hkv:].transpose(1, 2)
    pos = (pos0 + torch.arange(S))[None].expand(b, S)
    cos, sin = oracle.rope_cos_sin(pos, 128, 1e6, T)
    qr, kr = oracle.apply_rope(q, k, cos, sin)
    got_q = d.view(b, S, Hq + 2 * Hkv, 128)[:, :, :Hq].transpose(1, 2)
    assert rel(got_q, qr.float()) < 3e-3
    assert rel(kc[:, :, pos0:pos0 + S], kr.float()) < 3e-3
    assert torch.equal(vc[:, :, pos0:pos0 + S].float().cpu(), v.float())
    assert float(kc[:, :, :pos0].abs().max()) == 0.0 and float(kc[:, :, pos0 + S:].abs().max()) == 0.0


def test_argmax_first_index_wins(gpu_lib):
    x = torch.zeros(3, 152064); x[0, 7] = 5; x[0, 90000] = 5; x[1, 152063] = 1; x[2, :] = -1; x[2, 4000] = -0.5
    d = x.cuda(); out = torch.empty(3, dtype=torch.int32, device="cuda")
    _lib.check(gpu_lib.omchat_op_argmax(ptr(d), 3, 152064, ptr(out), None))
    sync()
    assert out.cpu().tolist() == [7, 152063, 4000]


def test_errors_are_raised(gpu_lib):
    a = torch.zeros(64, 100, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(ValueError):
        _lib.check(gpu_lib.omchat_op_gemm(_lib.BF16, ptr(a), 100, ptr(a), 100, ptr(a), 64, 64, 64, 100, None, None, None, 0, 0, 0, None))
