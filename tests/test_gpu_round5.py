"""GPU, round 5: launch shapes for the SHARD widths of a tensor-parallel rank's decode GEMVs (tuning key 34; gemv.hip / model.hip).

A TP = 8 rank of Qwen2-7B decodes with hidden 3584, 4 query heads / 1 kv head (qkv 768 rows, o_proj K = 512) and an MLP shard of 2368
(gate|up 4736 rows, down_proj K = 2368 = 37 chunks).  Round 4 sent those widths through the fall-back shapes of kernels tuned for the
full widths; round 5 gives them shapes of their own.  Every new shape is checked (1) at op level against the fp32 product and against the
round-4 shape, (2) inside a decode step of a model with exactly those local widths against the ORACLE (transformers modeling_qwen2.py:269-298
restated in oracle/decoder.py), batch 32 / 16 / 5 / 1."""
import pytest
import torch

pytestmark = pytest.mark.gpu
from gpu_util import DT, CODE, TOL, TOL_DEEP, dev, rnd, rel, ptr, sync, randn
from omchat_amd import synth, _lib
from omchat_amd.config import tiny
from omchat_amd.engine import Engine

DTS = ["bf16", "f16"]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("b,N,K,ks", [(32, 768, 3584, 1), (16, 768, 3584, 1), (5, 1152, 3584, 1),       # qkv shards (TP = 8, 4): x-stationary, one tile per workgroup
                                      (32, 3584, 512, 1), (7, 3584, 896, 1),                            # o_proj shards: one chunk per wave
                                      (32, 3584, 2368, 2), (16, 3584, 2368, 2), (32, 3584, 1792, 1)])   # down_proj shard (37 chunks in two slices), TP = 2 o_proj
def test_shard_width_gemv_shapes_vs_fp32_and_round4_shapes(gpu_lib, dt, b, N, K, ks):
    X = rnd(randn((b, K), 1), dt); W = rnd(randn((N, K), 2, 0.03), dt); bias = rnd(randn((N,), 3, 0.1), dt)
    dX, dW, db = dev(X, dt), dev(W, dt), dev(bias, dt)
    y = X @ W.t()
    code = CODE[dt]
    got = {}
    try:
        for key in (7, 0):
            gpu_lib.omchat_op_set_tuning(34, key)
            out = torch.full((b, N), float("nan"), dtype=DT[dt], device="cuda")
            _lib.check(gpu_lib.omchat_op_gemv_packed(code, ptr(dX), K, ptr(dW), K, ptr(out), N, b, N, K, ptr(db), _lib.EPI_NONE, 0, 1, 1, 0, None))
            part = torch.full((ks, b, N), float("nan"), dtype=torch.float32, device="cuda")
            _lib.check(gpu_lib.omchat_op_gemv_packed(code, ptr(dX), K, ptr(dW), K, ptr(part), N, b, N, K, None, 5, 0, ks, 1, 0, None))
            sync()
            assert rel(out, rnd(y + bias, dt)) < TOL[dt]
            assert torch.isfinite(part).all() and rel(part.sum(0), y) < 1e-4
            got[key] = (out.clone(), part.sum(0))
    finally:
        gpu_lib.omchat_op_set_tuning(34, 7)
    # the two shapes sum K in different orders: equal to fp32 rounding, and to one 16-bit ulp after the output rounding
    assert rel(got[7][1], got[0][1]) < 1e-5
    assert rel(got[7][0], got[0][0]) < TOL[dt]


@pytest.mark.parametrize("dt", DTS)
def test_batch1_short_gate_up_one_pair_per_wave(gpu_lib, dt):
    from test_gpu_ops import _gemm_ref
    N, K = 4736, 3584
    X = rnd(randn((1, K), 4), dt); W = rnd(randn((N, K), 5, 0.03), dt)
    ref = _gemm_ref(X, W, None, None, None, _lib.EPI_SWIGLU, dt)
    outs = {}
    try:
        for key in (7, 3):
            gpu_lib.omchat_op_set_tuning(34, key)
            o = torch.full((1, N // 2), float("nan"), dtype=DT[dt], device="cuda")
            _lib.check(gpu_lib.omchat_op_gemv(CODE[dt], ptr(dev(X, dt)), K, ptr(dev(W, dt)), K, ptr(o), N // 2, 1, N, K, None, None, 0, _lib.EPI_SWIGLU, 0, None))
            sync()
            assert rel(o, ref) < TOL[dt]
            outs[key] = o.clone()
    finally:
        gpu_lib.omchat_op_set_tuning(34, 7)
    assert torch.equal(outs[7], outs[3])          # one row per wave either way: the same sum order, the same bits


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("b", [32, 16, 5, 1])
def test_decode_step_at_the_tp8_rank_widths_vs_oracle(gpu_lib, dt, b):
    """a model whose FULL widths are a TP = 8 rank's local widths (hidden 3584, 4 q / 1 kv heads, MLP 2368, 2 layers) decodes two steps:
    logits against the oracle, with the round-5 shapes and with the round-4 ones; batch 1 with the norm-in-GEMV form off (tuning key 14)
    so that the step takes the residual + RMSNorm launches a tensor-parallel rank takes"""
    import oracle
    cfg = tiny(layers_t=2, q_heads=4, kv_heads=1, hidden_t=3584, mlp_t=2368, vocab=320)
    keep = lambda k: not k.startswith(synth.TOWER) and "mm_projector" not in k
    sd = {k: v for k, v in synth.state_dict(cfg, 5).items() if keep(k)}
    sdt = {k: rnd(torch.from_numpy(v), dt) for k, v in sd.items()}
    S = 70
    x = rnd(torch.randn(b, S, 3584, generator=torch.Generator().manual_seed(b)) * 0.5, dt)
    lens = [S - (i % 5) for i in range(b)]
    toks = torch.arange(b) % 300 + 5
    res = {}
    try:
        gpu_lib.omchat_op_set_tuning(14, 0)
        for key in (7, 0):
            gpu_lib.omchat_op_set_tuning(34, key)
            e = Engine(cfg, dtype=dt, max_seq=S + 8, max_batch=b, max_tiles=1, vision=False)
            e.load_state_dict(sd)
            e.prefill(x, lens)
            nxt, lg = e.decode_step(toks, want_logits=True)
            nxt2, lg2 = e.decode_step(nxt, want_logits=True); sync()
            res[key] = (lg.cpu(), lg2.cpu(), nxt.cpu())
            e.close()
    finally:
        gpu_lib.omchat_op_set_tuning(34, 7)
        gpu_lib.omchat_op_set_tuning(14, 3)
    for i in sorted({0, 1 % b, b // 2, b - 1}):
        cache = oracle.KVCache(cfg.text["num_hidden_layers"])
        oracle.qwen2_model(x[i:i + 1, :lens[i]], sdt, cfg.text, cache)
        o1 = oracle.decode_step(toks[i:i + 1][None].long(), sdt, cfg.text, cache)[0, 0]
        o2 = oracle.decode_step(res[7][2][i:i + 1][None].long(), sdt, cfg.text, cache)[0, 0]
        for key in (7, 0):
            assert rel(res[key][0][i], o1) < TOL_DEEP[dt], (key, i, rel(res[key][0][i], o1))
        assert rel(res[7][1][i], o2) < TOL_DEEP[dt], (i, rel(res[7][1][i], o2))
    assert rel(res[7][0], res[0][0]) < TOL_DEEP[dt]
