"""Round-3 GPU tests (all through the C ABI):
  * BASELINE configs[3] at FULL size: OmChat-2.1-8B geometry (InternViT-300M 24 layers + Qwen2-7B 28 layers), one sample = 8 pictures
    through the dynamic tiling of mm_utils.py:276-323 -> 8 tiles in one tower batch, S = 8704: size-independent properties
    (determinism, batch independence of the tower at 8 tiles, splice = pure copy, prefill / decode consistency)
  * the second-generation prefill attention at head_dim 64 (intern_vit_300m/modeling_intern_vit.py:205-222 shapes) vs fp32 torch
  * a weight reload while fp8 decode stays ENABLED must not stream the stale e4m3 replica
  * bench.py --shard-of: one rank's shard context with the no-op all-reduce runs the whole path
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
from gpu_util import DT, CODE, TOL, TOL_DEEP, dev, rnd, rel, sync, ptr, randn
from test_gpu_ops import _attn_ref
from omchat_amd import synth, _lib
from omchat_amd.config import omchat8b_21, omchat13b, tiny
from omchat_amd.engine import Engine
from omchat_amd.image_processing import HipImageProcessor

DEFAULT_KEY16 = 0          # loop form of the norm-in-GEMV launches: off since round 5 (the gate|up launch takes the one-pair-per-wave form, key 38)
DTS = ["bf16", "f16"]
CONSIST_TOL = {"bf16": 6e-2, "f16": 1e-2}


# ---------------------------------------------------------------------------------------------------------------------
# head_dim 64 on attn2_kernel: MHA shapes (the GQA shapes of test_gpu_vit300m.py stay on the first-generation kernel)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("b,S,H,causal,lens", [
    (3, 1025, 16, 0, None),            # InternViT-300M: 16 heads, 1024 + 1 tokens (ragged last key tile with ONE key, last query block with one row)
    (2, 200, 4, 0, [200, 77]),         # ragged key lengths
    (1, 300, 2, 1, None),              # causal diagonal inside tiles
    (2, 129, 3, 1, [129, 64]),
    (1, 64, 1, 0, None),               # exactly one tile
    (1, 1, 2, 0, None),                # a single token
])
def test_attn2_head_dim_64_mha(gpu_lib, dt, b, S, H, causal, lens):
    D = 64
    q = rnd(randn((b, S, H, D), 11), dt); k = rnd(randn((b, H, S, D), 12), dt); v = rnd(randn((b, H, S, D), 13), dt)
    dq, dk, dv = dev(q, dt), dev(k, dt), dev(v, dt)
    out = torch.full((b, S, H, D), float("nan"), dtype=DT[dt], device="cuda")
    dl = None if lens is None else torch.tensor(lens, dtype=torch.int32, device="cuda")
    _lib.check(gpu_lib.omchat_op_attn_prefill_d(CODE[dt], ptr(dq), ptr(dk), ptr(dv), ptr(out), b, S, S, H, H, D, ptr(dl), causal, 0, 0.125, None))
    sync()
    ref = _attn_ref(q, k, v, 0.125, causal, 0, lens or [S] * b)
    for i in range(b):
        n = S if lens is None else lens[i]
        assert torch.isfinite(out[i, :n].float()).all()
        assert rel(out[i, :n], ref[i, :n]) < TOL[dt], rel(out[i, :n], ref[i, :n])
    # the first-generation kernel (tuning key 8 = 0) computes the same quantity: A/B seam stays usable
    out1 = torch.full_like(out, float("nan"))
    _lib.check(gpu_lib.omchat_op_set_tuning(8, 0))
    try:
        _lib.check(gpu_lib.omchat_op_attn_prefill_d(CODE[dt], ptr(dq), ptr(dk), ptr(dv), ptr(out1), b, S, S, H, H, D, ptr(dl), causal, 0, 0.125, None))
        sync()
    finally:
        _lib.check(gpu_lib.omchat_op_set_tuning(8, 1))
    for i in range(b):
        n = S if lens is None else lens[i]
        assert rel(out1[i, :n], out[i, :n]) < 2 * TOL[dt]


def test_attn2_head_dim_64_rescale_branch_is_exercised(gpu_lib):
    """cdna_hip_programming.md rule 26: the lazy-rescale branch (running max moves by > 2^8) must be FORCED: one key row is spiked against
    every query late in the sequence so that the reference point jumps at a chosen tile"""
    dt, D, S, H = "bf16", 64, 320, 2
    q = rnd(randn((1, S, H, D), 21), dt); k = rnd(randn((1, H, S, D), 22), dt); v = rnd(randn((1, H, S, D), 23), dt)
    k[0, :, 200] = rnd(q[0, 5] * 12.0, dt)            # scores ~ 12 * |q|^2 * 0.125 ~ 100 >> 8 / log2(e): forces the rescale at key tile 3
    dq, dk, dv = dev(q, dt), dev(k, dt), dev(v, dt)
    out = torch.full((1, S, H, D), float("nan"), dtype=DT[dt], device="cuda")
    _lib.check(gpu_lib.omchat_op_attn_prefill_d(CODE[dt], ptr(dq), ptr(dk), ptr(dv), ptr(out), 1, S, S, H, H, D, None, 0, 0, 0.125, None))
    sync()
    ref = _attn_ref(q, k, v, 0.125, 0, 0, [S])
    assert torch.isfinite(out.float()).all()
    assert rel(out[0], ref[0]) < TOL[dt], rel(out[0], ref[0])


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3] at full size
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def eng8b(gpu_lib):
    cfg = omchat8b_21()
    S = 8 * 1024 + 512
    e = Engine(cfg, dtype="bf16", max_seq=S + 40, max_batch=1, max_tiles=8, max_prefill_rows=S + 8)
    e.fill_synthetic(0)
    yield e
    e.close()


def _eight_pictures(e):
    """8 pictures of 448 x 448 through dynamic_preprocess + preprocess on the device: a square tile-sized picture picks the 1 x 1 grid
    (no thumbnail: mm_utils.py:307), so the sample is 8 tiles"""
    proc = HipImageProcessor(crop_size=448)
    rng = np.random.default_rng(3)
    tiles = [proc.process_dynamic(rng.integers(0, 256, (448, 448, 3), dtype=np.uint8), max_num=6, dtype=e.torch_dtype) for _ in range(8)]
    assert all(t.shape == (1, 3, 448, 448) for t in tiles)
    return torch.cat(tiles, 0).contiguous()


def test_full_size_configs3_tower_8_tiles(eng8b):
    px = _eight_pictures(eng8b)
    a = eng8b.encode_images(px); sync()
    b = eng8b.encode_images(px); sync()
    assert a.shape == (8, 1024, 3584) and torch.isfinite(a.float()).all()
    assert torch.equal(a, b)                                              # determinism
    # batch independence at 8 tiles: tile i alone, and the tiles in another order, give the same rows
    one = eng8b.encode_images(px[5:6]); sync()
    assert rel(one[0], a[5]) < 1e-6
    perm = torch.tensor([7, 2, 5, 0, 1, 6, 3, 4])
    c = eng8b.encode_images(px[perm]); sync()
    assert rel(c, a[perm]) < 1e-6
    # a wide picture takes the multi-tile branch of the same front end: 896 x 448 -> 2 x 1 grid + thumbnail = 3 tiles in the same batch
    proc = HipImageProcessor(crop_size=448)
    wide = proc.process_dynamic(np.random.default_rng(4).integers(0, 256, (448, 896, 3), dtype=np.uint8), max_num=6, dtype=eng8b.torch_dtype)
    assert wide.shape[0] == 3
    w = eng8b.encode_images(torch.cat([wide, px[:5]], 0)); sync()
    assert rel(w[3:], a[:5]) < 1e-6


def test_full_size_configs3_prefill_decode_consistency(eng8b):
    S = 8 * 1024 + 512
    runs0 = _lib.lib().omchat_gemm_tune_runs()
    px = _eight_pictures(eng8b)
    feats = eng8b.encode_images(px)
    text = synth.token_ids(512, 151643, 1).tolist()
    row = []
    for i in range(8):
        row += [-200, text[i]]
    ids = torch.tensor([row[:-1] + text[7:]])
    embeds, lengths, valid = eng8b.splice(ids, None, feats); sync()
    assert lengths == [S] and bool(valid.all())
    for i in range(8):                                                   # pure copies, tile i at its sentinel (omchat_arch.py:133-158)
        assert torch.equal(embeds[0, i * 1025:i * 1025 + 1024], feats[i])
    logits_a, _ = eng8b.prefill(embeds, [S]); sync()
    tok = int(torch.argmax(logits_a[0]))
    nxt, step_logits = eng8b.decode_step(torch.tensor([tok]), want_logits=True); sync()
    seq = [tok, int(nxt[0])]
    t2 = nxt
    for _ in range(6):
        t2, _ = eng8b.decode_step(t2)
        seq.append(int(t2[0]))
    ids2 = torch.cat([ids, torch.tensor([[tok]])], dim=1)
    embeds2, _, _ = eng8b.splice(ids2, None, feats)
    logits_b, _ = eng8b.prefill(embeds2, [S + 1]); sync()
    r = rel(step_logits[0], logits_b[0])
    print("configs3 shape prefill/decode consistency", r)
    assert torch.isfinite(logits_b).all() and r < CONSIST_TOL["bf16"], r
    logits_c, _ = eng8b.prefill(embeds, [S])
    assert torch.equal(logits_c, logits_a)
    t3 = eng8b.argmax(logits_c)
    seq2 = [int(t3[0])]
    for _ in range(7):
        t3, _ = eng8b.decode_step(t3)
        seq2.append(int(t3[0]))
    assert seq2 == seq
    assert _lib.lib().omchat_gemm_tune_runs() == runs0, "a GEMM class of the configs[3] workload is missing from gemm_tune_gfx950.txt"


# ---------------------------------------------------------------------------------------------------------------------
# ADVICE r02 (model.hip): reload with fp8 decode left on
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("graph", [False, True])
def test_fp8_decode_follows_a_reload_without_reenabling(gpu_lib, graph):
    cfg = tiny()
    sd = synth.state_dict(cfg, 5)
    e = Engine(cfg, dtype="bf16", max_seq=64, max_batch=1, max_tiles=1, vision=False)
    e.load_state_dict({k: v for k, v in sd.items() if not k.startswith(synth.TOWER) and "mm_projector" not in k})
    if graph:
        e.enable_decode_graph(True)
    x = torch.randn(1, 9, 256, generator=torch.Generator().manual_seed(0)) * 0.5
    e.enable_fp8_decode(True)

    def step():
        e.prefill(x)
        _, lg = e.decode_step(torch.tensor([7]), want_logits=True); sync()
        return lg.clone()
    a = step()
    name = "model.layers.1.mlp.down_proj.weight"
    e.load_tensor(name, torch.from_numpy(sd[name]) * 3.0)               # fp8 decode stays enabled; nobody calls enable again
    b = step()
    assert not torch.equal(a, b), "decode streamed the stale e4m3 replica after a weight reload"
    e.enable_fp8_decode(False)
    if graph:
        e.enable_decode_graph(False)
    ref = step()
    assert rel(b, ref) < 0.12                                           # e4m3 weights vs the 16-bit weights of the SAME (new) values
    e.close()


# ---------------------------------------------------------------------------------------------------------------------
# bench.py --shard-of N: a rank context of a TP group with the no-op all-reduce
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [2, 8])
def test_shard_context_runs_the_whole_path(gpu_lib, n):
    cfg = tiny(heads_v=2, q_heads=8, kv_heads=2, mlp_v=1024, mlp_t=1024, vocab=512) if n == 2 else tiny(heads_v=2, q_heads=8, kv_heads=4, mlp_v=1024, mlp_t=1024, vocab=512)
    e = Engine(cfg, dtype="bf16", max_seq=96, max_batch=2, max_tiles=2, tp_rank=0, tp_size=n)
    e.set_noop_allreduce()
    e.fill_synthetic(0, local=True)
    px = torch.from_numpy(synth.pixels(2, cfg.vision["image_size"], 0))
    feats = e.encode_images(px)
    ids = torch.tensor([[3, -200, 5, 6], [4, -200, 8, 9]])
    embeds, lengths, _ = e.splice(ids, None, feats)
    logits, _ = e.prefill(embeds, lengths)
    tok = e.argmax(logits)
    for _ in range(3):
        tok, _ = e.decode_step(tok)
    sync()
    assert logits.shape[1] == cfg.text["vocab_size"] // n and torch.isfinite(logits).all()
    assert all(0 <= int(t) < cfg.text["vocab_size"] for t in tok)
    e.close()


# ---------------------------------------------------------------------------------------------------------------------
# batched decode attention: K tiles as whole rows through LDS (tuning key 12) vs fragment-shaped register loads
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTS)
def test_decode_attention_k_through_lds_is_bit_identical(gpu_lib, dt):
    """same keys, same MFMA operands, same order: the K load path (whole 256-byte rows -> LDS image -> ds_read_b128 fragments vs 16 rows x
    64 B straight to registers) must not change a bit -- op level (ragged lengths, poisoned cache tail) and inside the decode step, where
    the tile that owns the new position rotates (RoPE) and appends the fresh K row (positions at tile edges)"""
    b, Hq, Hkv, cap, lens = 3, 8, 2, 1024, [1000, 513, 64]
    q = rnd(randn((b, Hq, 128), 1), dt); k = rnd(randn((b, Hkv, cap, 128), 2), dt); v = rnd(randn((b, Hkv, cap, 128), 3), dt)
    for i, n in enumerate(lens):
        k[i, :, n:] = float("nan"); v[i, :, n:] = float("nan")
    dq, dk, dv = dev(q, dt), dev(k, dt), dev(v, dt)
    L = max(lens)
    wsb = gpu_lib.omchat_op_attn_decode_ws(b, Hq, L)
    ws = torch.empty(wsb // 4 + 4, dtype=torch.float32, device="cuda")
    dl = torch.tensor(lens, dtype=torch.int32, device="cuda")
    ref = _attn_ref(q[:, None], torch.nan_to_num(k), torch.nan_to_num(v), 128 ** -0.5, 0, 0, lens)[:, 0]
    outs = {}
    try:
        gpu_lib.omchat_op_set_tuning(10, 4)
        gpu_lib.omchat_op_set_tuning(25, 0)      # the register multi-tile kernel (the LDS-DMA ring form of round 4 has its own tests)
        for klds in (0, 1):
            gpu_lib.omchat_op_set_tuning(12, klds)
            out = torch.full((b, Hq, 128), float("nan"), dtype=DT[dt], device="cuda")
            _lib.check(gpu_lib.omchat_op_attn_decode(CODE[dt], ptr(dq), ptr(dk), ptr(dv), ptr(out), b, Hq, Hkv, cap, L, ptr(dl), 128 ** -0.5, ptr(ws), wsb, None))
            sync()
            assert rel(out, ref) < TOL[dt]
            outs[klds] = out.clone()
        assert torch.equal(outs[0], outs[1])
        # inside the model: fused RoPE + append of the fresh row
        cfg = tiny(q_heads=4, kv_heads=2)
        sd = {k_: v_ for k_, v_ in synth.state_dict(cfg, 5).items() if not k_.startswith(synth.TOWER) and "mm_projector" not in k_}
        x = torch.randn(4, 200, 256, generator=torch.Generator().manual_seed(11)) * 0.5
        lens2 = [200, 127, 129, 64]
        runs = {}
        for klds in (0, 1):
            gpu_lib.omchat_op_set_tuning(12, klds)
            e = Engine(cfg, dtype=dt, max_seq=256, max_batch=4, max_tiles=1, vision=False)
            e.load_state_dict(sd)
            e.prefill(x, lens2)
            tok = torch.arange(4) % 300 + 7
            seq = []
            for _ in range(4):
                tok, lg = e.decode_step(tok, want_logits=True)
                seq.append(lg.float().cpu().clone())
            sync()
            runs[klds] = seq
            e.close()
        for a_, b_ in zip(runs[0], runs[1]):
            assert torch.isfinite(a_).all() and torch.equal(a_, b_)
    finally:
        gpu_lib.omchat_op_set_tuning(10, 0)
        gpu_lib.omchat_op_set_tuning(12, 0)
        gpu_lib.omchat_op_set_tuning(25, 1)


# ---------------------------------------------------------------------------------------------------------------------
# persistent 256x256 GEMM (one workgroup per CU, next tile's prologue issued before the epilogue; tuning key 13)
# ---------------------------------------------------------------------------------------------------------------------
from test_gpu_ops import _gemm_ref


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,N,K", [(3075, 5376, 192), (2049, 8224, 64), (4100, 4128, 448)])
def test_gemm_persistent_equals_one_workgroup_per_tile(gpu_lib, dt, M, N, K):
    """multi-round launches (> 256 tiles of 256^2; ragged last round, ragged M and N edges, K of one and of several steps): every epilogue
    against the fp32 reference and BIT-IDENTICAL to the one-workgroup-per-tile launch (same per-element accumulation order)"""
    assert -(-M // 256) * -(-N // 256) > 256
    A = rnd(randn((M, K), 1), dt); W = rnd(randn((N, K), 2, 0.05), dt)
    bias = rnd(randn((N,), 3, 0.1), dt); ls = rnd(randn((N,), 4, 0.1) + 0.1, dt); resid = rnd(randn((M, N), 5), dt)
    dA, dW, db, dl, dr = dev(A, dt), dev(W, dt), dev(bias, dt), dev(ls, dt), dev(resid, dt)
    try:
        for epi in (_lib.EPI_NONE, _lib.EPI_GELU, _lib.EPI_LS_RESID, _lib.EPI_RESID, _lib.EPI_SWIGLU):
            No = N // 2 if epi == _lib.EPI_SWIGLU else N
            use_bias = epi != _lib.EPI_SWIGLU and epi != _lib.EPI_RESID
            outs = {}
            for persist in (1, 0):
                gpu_lib.omchat_op_set_tuning(13, persist)
                out = torch.full((M, No), float("nan"), dtype=DT[dt], device="cuda")
                _lib.check(gpu_lib.omchat_op_gemm(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(out), No, M, N, K, ptr(db) if use_bias else None,
                                                  ptr(dl), ptr(dr), N, epi, 2, None))
                sync()
                outs[persist] = out
            assert torch.isfinite(outs[1].float()).all(), (epi, "non-finite / unwritten outputs")
            ref = _gemm_ref(A, W, bias if use_bias else None, ls, resid, epi, dt)
            assert rel(outs[1], ref) < TOL[dt], (epi, rel(outs[1], ref))
            assert torch.equal(outs[0], outs[1]), epi
    finally:
        gpu_lib.omchat_op_set_tuning(13, 0)


def test_gemm_persistent_race_screen(gpu_lib):
    """production K loop (ViT fc1: M = 3075, N = 12800, K = 3200 -> 650 tiles, 3 rounds): repeated launches bit-identical, element-wise
    error bounded (no wrong tile hiding inside a small Frobenius error), residual aliasing the output as the model uses it"""
    dt, M, N, K = "bf16", 3075, 12800, 3200
    A = rnd(randn((M, K), 11, 0.5), dt); W = rnd(randn((N, K), 12, 0.05), dt)
    dA, dW = dev(A, dt), dev(W, dt)
    ref = A @ W.t()
    outs = []
    gpu_lib.omchat_op_set_tuning(13, 1)
    for it in range(3):
        out = torch.full((M, N), float("nan"), dtype=DT[dt], device="cuda")
        _lib.check(gpu_lib.omchat_op_gemm(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(out), N, M, N, K, None, None, None, 0, _lib.EPI_NONE, 2, None))
        sync()
        outs.append(out)
    err = (outs[0].float().cpu() - ref).abs()
    assert float(err.max()) < 0.25 * float(ref.abs().max())
    assert rel(outs[0], ref) < TOL[dt]
    assert torch.equal(outs[1], outs[0]) and torch.equal(outs[2], outs[0])
    # in-place residual (C aliases resid), as the decoder's o_proj / down_proj run
    x = rnd(randn((M, N), 13), dt)
    dx = dev(x, dt)
    _lib.check(gpu_lib.omchat_op_gemm(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(dx), N, M, N, K, None, None, ptr(dx), N, _lib.EPI_RESID, 2, None))
    sync()
    gpu_lib.omchat_op_set_tuning(13, 0)
    assert rel(dx, x + rnd(ref, dt)) < TOL[dt]


# ---------------------------------------------------------------------------------------------------------------------
# batch-1 decode: post-attention RMSNorm in the registers of the gate|up GEMV (tuning key 14), o_proj writing x + attn itself
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("N,K,epi", [(512, 256, "none"), (37888, 3584, "swiglu"), (4608, 3584, "none"), (96, 4096, "swiglu"), (320, 64, "none")])
def test_gemv_with_the_norm_in_registers(gpu_lib, dt, N, K, epi):
    """y = epi(W RMSNorm(x)) against fp32 torch with Qwen2RMSNorm's rounding points (x * rsqrt(mean(x^2) + eps) rounded to T, times w rounded
    to T: modeling_qwen2.py:247-252), SwiGLU on the 16-row interleaved gate | up layout"""
    x = rnd(randn((K,), 1), dt); w = rnd(randn((N, K), 2, 0.05), dt); nw = rnd(randn((K,), 3, 0.1) + 1.0, dt)
    eps = 1e-6
    xn = rnd(nw * rnd(x * torch.rsqrt((x * x).mean() + eps), dt), dt)
    y = w @ xn
    code = _lib.EPI_NONE
    if epi == "swiglu":
        code = _lib.EPI_SWIGLU
        blocks = y.reshape(N // 32, 2, 16)
        g, u = rnd(blocks[:, 0], dt), rnd(blocks[:, 1], dt)
        y = (rnd(torch.nn.functional.silu(g), dt) * u).reshape(N // 2)
    dx, dw, dn = dev(x, dt), dev(w, dt), dev(nw, dt)
    out = torch.full((y.numel(),), float("nan"), dtype=DT[dt], device="cuda")
    _lib.check(gpu_lib.omchat_op_gemv_norm(CODE[dt], ptr(dx), ptr(dw), K, ptr(out), N, K, ptr(dn), eps, None, code, 0, None))
    sync()
    assert torch.isfinite(out.float()).all()
    assert rel(out, y) < TOL[dt], rel(out, y)
    with pytest.raises(ValueError):           # K > 4096 does not fit a wave's registers: refused, never silently un-normalised
        big = torch.zeros(8192, dtype=DT[dt], device="cuda"); bw = torch.zeros(32, 8192, dtype=DT[dt], device="cuda")
        _lib.check(gpu_lib.omchat_op_gemv_norm(CODE[dt], ptr(big), ptr(bw), 8192, ptr(out), 32, 8192, ptr(big), eps, None, _lib.EPI_NONE, 0, None))


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("N,K,epi", [(37888, 3584, "swiglu"), (4608, 3584, "none"), (8192 + 64, 2048, "swiglu"), (5000, 4096, "none"), (4100, 3000 + 8, "none")])
def test_gemv_norm_loop_form_is_bit_identical(gpu_lib, dt, N, K, epi):
    """gemv_rows_norm_loop_kernel (tuning key 16: one resident round of workgroups, each wave walking its outputs through a three-buffer
    register ring) against the one-shot form: the same bits; ragged K (not a multiple of 512), output counts that do not divide by the grid"""
    x = rnd(randn((K,), 1), dt); w = rnd(randn((N, K), 2, 0.05), dt); nw = rnd(randn((K,), 3, 0.1) + 1.0, dt); bias = rnd(randn((N,), 4), dt)
    code = _lib.EPI_SWIGLU if epi == "swiglu" else _lib.EPI_NONE
    dx, dw, dn, db = dev(x, dt), dev(w, dt), dev(nw, dt), dev(bias, dt)
    n_y = N // 2 if epi == "swiglu" else N
    outs = []
    try:
        for key in (0, 15):
            gpu_lib.omchat_op_set_tuning(16, key)
            out = torch.full((n_y,), float("nan"), dtype=DT[dt], device="cuda")
            _lib.check(gpu_lib.omchat_op_gemv_norm(CODE[dt], ptr(dx), ptr(dw), K, ptr(out), N, K, ptr(dn), 1e-6, None if epi == "swiglu" else ptr(db), code, 0, None))
            sync()
            outs.append(out)
    finally:
        gpu_lib.omchat_op_set_tuning(16, DEFAULT_KEY16)
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("dt", DTS)
def test_gemv_balanced_workgroups_for_o_proj_and_qkv(gpu_lib, dt):
    """tuning key 17 (default 1): rows that deal evenly to two workgroups per CU take N / (2 CUs) waves per workgroup -- o_proj (3584 rows,
    residual epilogue: 7 waves, the same bits as the four-wave launch) and qkv with the norm in registers (4608 rows: 9 waves, the sum of
    squares meets in a different order: fp32 rounding of one sum); both against fp32 torch"""
    K = 3584
    x = rnd(randn((K,), 1), dt); nw = rnd(randn((K,), 3, 0.1) + 1.0, dt)
    wo = rnd(randn((3584, K), 2, 0.05), dt); r = rnd(randn((3584,), 5), dt)
    wq = rnd(randn((4608, K), 6, 0.05), dt); bq = rnd(randn((4608,), 7), dt)
    ref_o = r + rnd(wo @ x, dt)
    xn = rnd(nw * rnd(x * torch.rsqrt((x * x).mean() + 1e-6), dt), dt)
    ref_q = wq @ xn + bq
    dx, dn, dwo, dr, dwq, dbq = dev(x, dt), dev(nw, dt), dev(wo, dt), dev(r, dt), dev(wq, dt), dev(bq, dt)
    got = {}
    try:
        for key in (0, 1):
            gpu_lib.omchat_op_set_tuning(17, key)
            yo = torch.full((3584,), float("nan"), dtype=DT[dt], device="cuda")
            _lib.check(gpu_lib.omchat_op_gemv(CODE[dt], ptr(dx), K, ptr(dwo), K, ptr(yo), 3584, 1, 3584, K, None, ptr(dr), 3584, _lib.EPI_RESID, 0, None))
            yq = torch.full((4608,), float("nan"), dtype=DT[dt], device="cuda")
            _lib.check(gpu_lib.omchat_op_gemv_norm(CODE[dt], ptr(dx), ptr(dwq), K, ptr(yq), 4608, K, ptr(dn), 1e-6, ptr(dbq), _lib.EPI_NONE, 0, None))
            sync()
            got[key] = (yo, yq)
    finally:
        gpu_lib.omchat_op_set_tuning(17, 1)
    for key in (0, 1):
        assert rel(got[key][0], ref_o) < TOL[dt] and rel(got[key][1], ref_q) < TOL[dt]
    assert torch.equal(got[0][0], got[1][0])
    assert rel(got[0][1], got[1][1].float().cpu()) < TOL[dt]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("fp8", [False, True])
def test_decode_with_norm_in_gemv_vs_separate_launch_and_oracle(gpu_lib, dt, fp8):
    """the seven-launch layer (key 14 = 1, default) against the eight-launch layer (key 14 = 0) and the oracle: same model, 5 decode steps;
    the two differ only in the fp32 summation order of the o_proj K sum and of the sum of squares"""
    cfg = tiny(q_heads=4, kv_heads=2)
    sd = {k: v for k, v in synth.state_dict(cfg, 9).items() if not k.startswith(synth.TOWER) and "mm_projector" not in k}
    x = rnd(torch.randn(1, 21, 256, generator=torch.Generator().manual_seed(2)) * 0.5, dt)
    runs = {}
    try:
        for key in (1, 0):
            gpu_lib.omchat_op_set_tuning(14, 7 if key else 0)
            e = Engine(cfg, dtype=dt, max_seq=64, max_batch=1, max_tiles=1, vision=False)
            e.load_state_dict(sd)
            if fp8:
                e.enable_fp8_decode(True)
            e.prefill(x)
            tok, outs = torch.tensor([11]), []
            for _ in range(5):
                _, lg = e.decode_step(tok, want_logits=True)
                outs.append(lg.float().cpu().clone())
                tok = torch.tensor([int(torch.argmax(outs[-1][0])) if key == 1 or not runs else runs[1][1][len(outs) - 1]])
            sync()
            runs[key] = (outs, [int(torch.argmax(o[0])) for o in outs])
            e.close()
    finally:
        gpu_lib.omchat_op_set_tuning(14, 3)
    for a_, b_ in zip(runs[1][0], runs[0][0]):
        assert rel(a_, b_) < (1.2e-2 if dt == "bf16" else 2.5e-3), rel(a_, b_)
    if not fp8:
        import oracle
        sdt = {k: rnd(torch.from_numpy(v), dt) for k, v in sd.items()}
        cache = oracle.KVCache(cfg.text["num_hidden_layers"])
        oracle.qwen2_model(x, sdt, cfg.text, cache)
        tok = 11
        for s_ in range(5):
            ref = oracle.decode_step(torch.tensor([[tok]]), sdt, cfg.text, cache)[0, 0]
            assert rel(runs[1][0][s_][0], ref) < TOL_DEEP[dt], (s_, rel(runs[1][0][s_][0], ref))
            tok = runs[1][1][s_]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("N,K,rows_resid", [(3584, 18944, True), (256, 4608, True), (96, 8256, False), (7, 32768, True), (3000, 12352, True)])
def test_gemv_long_k_without_split(gpu_lib, dt, N, K, rows_resid):
    """down_proj of a batch-1 step without split-K (gemv_rows_longk_kernel: x through LDS, weights in passes of 8 x 512): y = resid +
    T(W x) in place, against fp32 torch; ragged K (not a multiple of 512 / of a pass), N not a multiple of the rows per workgroup"""
    x = rnd(randn((K,), 1, 0.5), dt); w = rnd(randn((N, K), 2, 0.05), dt); r = rnd(randn((N,), 3), dt)
    ref = r + rnd(w @ x, dt)
    dx, dw, dr = dev(x, dt), dev(w, dt), dev(r, dt)
    # in place (the decode step's form) or with the residual in its own buffer
    y = dr if rows_resid else torch.full((N,), float("nan"), dtype=DT[dt], device="cuda")
    _lib.check(gpu_lib.omchat_op_gemv(CODE[dt], ptr(dx), K, ptr(dw), K, ptr(y), N, 1, N, K, None, ptr(dr), N, _lib.EPI_RESID, 0, None))
    sync()
    assert torch.isfinite(y.float()).all()
    assert rel(y, ref) < TOL[dt], rel(y, ref)


@pytest.mark.parametrize("dt", DTS)
def test_decode_six_launch_layer_modes_agree(gpu_lib, dt):
    """tuning key 14 = 3 (default: both norms inside their consumers, o_proj / down_proj un-split), 1 (post-attention norm only), 0 (eight
    launches): the same model, the same tokens, logits equal to fp32-summation-order noise; mlp width 4608 takes the long-K kernel"""
    cfg = tiny(q_heads=4, kv_heads=2, mlp_t=4608)
    sd = {k: v for k, v in synth.state_dict(cfg, 9).items() if not k.startswith(synth.TOWER) and "mm_projector" not in k}
    x = rnd(torch.randn(1, 21, 256, generator=torch.Generator().manual_seed(2)) * 0.5, dt)
    toks = [11, 3, 250, 17, 99]
    runs = {}
    try:
        for key in (3, 1, 0):
            gpu_lib.omchat_op_set_tuning(14, key)
            e = Engine(cfg, dtype=dt, max_seq=64, max_batch=1, max_tiles=1, vision=False)
            e.load_state_dict(sd)
            e.prefill(x)
            outs = []
            for t in toks:
                nxt, lg = e.decode_step(torch.tensor([t]), want_logits=True)
                outs.append(lg.float().cpu().clone())
                assert int(nxt[0]) == int(torch.argmax(outs[-1][0]))
            sync()
            runs[key] = outs
            e.close()
    finally:
        gpu_lib.omchat_op_set_tuning(14, 3)
    for key in (1, 0):
        for a_, b_ in zip(runs[3], runs[key]):
            assert torch.isfinite(a_).all() and rel(a_, b_) < (1.2e-2 if dt == "bf16" else 2.5e-3), (key, rel(a_, b_))
    import oracle
    sdt = {k: rnd(torch.from_numpy(v), dt) for k, v in sd.items()}
    cache = oracle.KVCache(cfg.text["num_hidden_layers"])
    oracle.qwen2_model(x, sdt, cfg.text, cache)
    for s_, t in enumerate(toks):
        ref = oracle.decode_step(torch.tensor([[t]]), sdt, cfg.text, cache)[0, 0]
        assert rel(runs[3][s_][0], ref) < TOL_DEEP[dt], (s_, rel(runs[3][s_][0], ref))
