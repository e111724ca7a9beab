"""GPU, round 5: launch shapes for the SHARD widths of a tensor-parallel rank's decode GEMVs (tuning key 34; gemv.hip / model.hip).

A TP = 8 rank of Qwen2-7B decodes with hidden 3584, 4 query heads / 1 kv head (qkv 768 rows, o_proj K = 512) and an MLP shard of 2368
(gate|up 4736 rows, down_proj K = 2368 = 37 chunks).  Round 4 sent those widths through the fall-back shapes of kernels tuned for the
full widths; round 5 gives them shapes of their own.  Every new shape is checked (1) at op level against the fp32 product and against the
round-4 shape, (2) inside a decode step of a model with exactly those local widths against the ORACLE (transformers modeling_qwen2.py:269-298
restated in oracle/decoder.py), batch 32 / 16 / 5 / 1."""
import pytest
import torch

pytestmark = pytest.mark.gpu
from gpu_util import DT, CODE, TOL, TOL_DEEP, dev, rnd, rel, ptr, sync, randn, synth_state_dict
from omchat_amd import synth, _lib
from omchat_amd.config import tiny
from omchat_amd.engine import Engine

DEFAULT_KEY39 = 0      # tuning key 39 of the shipped library (0 = long-K GEMV stages x in LDS)

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
        for key in (15, 0):
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
        gpu_lib.omchat_op_set_tuning(34, 15)
    # the two shapes sum K in different orders: equal to fp32 rounding, and to one 16-bit ulp after the output rounding
    assert rel(got[15][1], got[0][1]) < 1e-5
    assert rel(got[15][0], got[0][0]) < TOL[dt]


@pytest.mark.parametrize("dt", DTS)
def test_batch1_short_gate_up_one_pair_per_wave(gpu_lib, dt):
    from test_gpu_ops import _gemm_ref
    N, K = 4736, 3584
    X = rnd(randn((1, K), 4), dt); W = rnd(randn((N, K), 5, 0.03), dt)
    ref = _gemm_ref(X, W, None, None, None, _lib.EPI_SWIGLU, dt)
    outs = {}
    try:
        for key in (15, 11):
            gpu_lib.omchat_op_set_tuning(34, key)
            o = torch.full((1, N // 2), float("nan"), dtype=DT[dt], device="cuda")
            _lib.check(gpu_lib.omchat_op_gemv(CODE[dt], ptr(dev(X, dt)), K, ptr(dev(W, dt)), K, ptr(o), N // 2, 1, N, K, None, None, 0, _lib.EPI_SWIGLU, 0, None))
            sync()
            assert rel(o, ref) < TOL[dt]
            outs[key] = o.clone()
    finally:
        gpu_lib.omchat_op_set_tuning(34, 15)
    assert torch.equal(outs[15], outs[11])          # one row per wave either way: the same sum order, the same bits


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
        for key in (15, 0):
            gpu_lib.omchat_op_set_tuning(34, key)
            e = Engine(cfg, dtype=dt, max_seq=S + 8, max_batch=b, max_tiles=1, vision=False)
            e.load_state_dict(sd)
            e.prefill(x, lens)
            nxt, lg = e.decode_step(toks, want_logits=True)
            nxt2, lg2 = e.decode_step(nxt, want_logits=True); sync()
            res[key] = (lg.cpu(), lg2.cpu(), nxt.cpu())
            e.close()
    finally:
        gpu_lib.omchat_op_set_tuning(34, 15)
        gpu_lib.omchat_op_set_tuning(14, 3)
    for i in sorted({0, 1 % b, b // 2, b - 1}):
        cache = oracle.KVCache(cfg.text["num_hidden_layers"])
        oracle.qwen2_model(x[i:i + 1, :lens[i]], sdt, cfg.text, cache)
        o1 = oracle.decode_step(toks[i:i + 1][None].long(), sdt, cfg.text, cache)[0, 0]
        o2 = oracle.decode_step(res[15][2][i:i + 1][None].long(), sdt, cfg.text, cache)[0, 0]
        for key in (15, 0):
            assert rel(res[key][0][i], o1) < TOL_DEEP[dt], (key, i, rel(res[key][0][i], o1))
        assert rel(res[15][1][i], o2) < TOL_DEEP[dt], (i, rel(res[15][1][i], o2))
    assert rel(res[15][0], res[0][0]) < TOL_DEEP[dt]


# ---------------------------------------------------------------------------------------------------------------------
# padded-batch decode (omchat_arch.py:61-70): device-resident mask / positions, e4m3 cache, full width
# ---------------------------------------------------------------------------------------------------------------------
def _tiny_padded(dt, g, fp8_kv=False):
    import numpy as np
    from omchat_amd.model.omchat_qwen2 import OmChatQwen2ForCausalLM
    cfg = tiny()
    e = Engine(cfg, dtype=dt, max_seq=128, max_batch=2, max_tiles=3)
    e.load_state_dict(synth.state_dict(cfg, int(g["seed"])))
    if fp8_kv:
        e.enable_fp8_kv(True)
    m = OmChatQwen2ForCausalLM(cfg.clone(), e)
    feats = rnd(torch.from_numpy(np.ascontiguousarray(g["feats"])).float(), dt)
    m.encode_images = lambda images: feats.to(DT[dt]).cuda()
    return cfg, e, m


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("side", ["right", "left"])
def test_masked_steps_without_host_data_give_the_bits_of_the_host_mask_steps(gpu_lib, dt, side):
    """omchat_masked_decode_begin + omchat_decode_step_masked_next (mask and positions resident on the device, no synchronisation) against
    omchat_decode_step_masked fed with the reference's own per-step masks / positions (tests/golden/leftpad_decode.npz): identical logits, both
    padding sides; omchat_kv_rewind takes a speculative step back (slots AND positions)."""
    from conftest import golden
    g = golden("leftpad_decode")
    ids, mask = torch.from_numpy(g["ids"]).long(), torch.from_numpy(g["mask"]).long()
    dummy = torch.zeros(3, 3, 56, 56)
    toks = [torch.from_numpy(g[f"{side}_tok_{k}"]).long() for k in range(int(g["steps"]))]
    # (a) the host-mask form
    cfg, e, m = _tiny_padded(dt, g)
    m.config.mm["tokenizer_padding_side"] = side
    kv = m(input_ids=ids, attention_mask=mask, images=dummy, use_cache=True).past_key_values
    want = []
    for k, t in enumerate(toks):
        nxt, lg = e.decode_step_masked(t, torch.from_numpy(g[f"{side}_dec_pos_{k}"]), torch.from_numpy(g[f"{side}_dec_mask_{k}"]), want_logits=True)
        want.append(lg.clone())
    e.close()
    # (b) the device-resident form: begin with the FIRST step's mask / positions only
    cfg, e, m = _tiny_padded(dt, g)
    m.config.mm["tokenizer_padding_side"] = side
    m(input_ids=ids, attention_mask=mask, images=dummy, use_cache=True)
    e.masked_decode_begin(torch.from_numpy(g[f"{side}_dec_pos_0"]), torch.from_numpy(g[f"{side}_dec_mask_0"]))
    got = []
    for k, t in enumerate(toks):
        nxt, lg = e.decode_step_masked_next(t, want_logits=True)
        got.append(lg.clone())
        if k == 1:      # a speculative step that is taken back: the repeated step must give the same logits again
            e.decode_step_masked_next(toks[2])
            e.kv_rewind(2, 1)
    sync()
    for k in range(len(toks)):
        assert torch.equal(got[k], want[k]), (side, k, rel(got[k], want[k]))
    assert e.kv_lengths(2) == [int(g[side + "_S"]) + len(toks)] * 2
    # the per-step mask form after `next` needs no begin; `next` after it does
    e.decode_step_masked(toks[0], torch.from_numpy(g[f"{side}_dec_pos_0"]) + 3, torch.ones(2, int(g[side + "_S"]) + len(toks) + 1, dtype=torch.long))
    with pytest.raises(ValueError, match="omchat_masked_decode_begin"):
        e.decode_step_masked_next(toks[0])
    e.close()


@pytest.mark.parametrize("side", ["right", "left"])
def test_masked_decode_on_the_e4m3_cache(gpu_lib, side):
    """round 4 refused the fp8 KV cache in the masked step; now the new rows are rotated to their own positions, appended AND quantised at the
    common slot (rope_kv slot0) and the split-KV kernel applies the key mask on the e4m3 cache: logits within the e4m3 cache tolerance of the
    16-bit masked steps (which match the reference), every step"""
    from conftest import golden
    g = golden("leftpad_decode")
    dt = "bf16"
    ids, mask = torch.from_numpy(g["ids"]).long(), torch.from_numpy(g["mask"]).long()
    dummy = torch.zeros(3, 3, 56, 56)
    res = {}
    for f8 in (False, True):
        cfg, e, m = _tiny_padded(dt, g, fp8_kv=f8)
        m.config.mm["tokenizer_padding_side"] = side
        m(input_ids=ids, attention_mask=mask, images=dummy, use_cache=True)
        e.masked_decode_begin(torch.from_numpy(g[f"{side}_dec_pos_0"]), torch.from_numpy(g[f"{side}_dec_mask_0"]))
        out = []
        for k in range(int(g["steps"])):
            _, lg = e.decode_step_masked_next(torch.from_numpy(g[f"{side}_tok_{k}"]).long(), want_logits=True)
            out.append(lg.float().cpu())
        sync(); e.close()
        res[f8] = out
    for k in range(int(g["steps"])):
        ref = torch.from_numpy(g[f"{side}_logits_{k}"]).float()
        d8, d16 = rel(res[True][k], ref), rel(res[False][k], ref)
        assert d16 < TOL_DEEP[dt] and 1e-4 < rel(res[True][k], res[False][k]) < 0.08 and d8 < 0.08, (side, k, d8, d16)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("side", ["right", "left"])
def test_full_width_masked_decode_vs_the_oracle_restatement(gpu_lib, dt, side):
    """the padded-batch decode at the REAL widths (Qwen2-7B: 3584 hidden, 28 query / 4 kv heads, MLP 18944; 2 layers, 1024
    image tokens per sentinel): three rows of different spliced length (2 images + 12 ids, 1 image + 30 ids, 0 images + 9 ids), padded prefill,
    three teacher-forced decode steps through generate()'s own path (begin + next) against the oracle's restatement of omchat_arch.py:61-70
    (oracle.splice_inputs / decode_step_inputs / qwen2_model with the mask and position_ids the decode branch returns)."""
    import numpy as np
    import oracle
    from omchat_amd.config import omchat13b
    from omchat_amd.model.omchat_qwen2 import OmChatQwen2ForCausalLM
    cfg = omchat13b()
    cfg.text["num_hidden_layers"] = 2
    cfg.text["vocab_size"] = 2048
    cfg.mm["tokenizer_padding_side"] = side
    keep = lambda k: not k.startswith(synth.TOWER) and "mm_projector" not in k
    sd = synth_state_dict(cfg, 11, keep)          # generated on the device (bit-identical to synth.state_dict, ~100 x faster at these widths)
    sdt = {k: rnd(torch.from_numpy(v), dt) for k, v in sd.items()}
    e = Engine(cfg, dtype=dt, max_seq=2200, max_batch=3, max_tiles=1, vision=False)
    e.load_state_dict(sd)
    m = OmChatQwen2ForCausalLM(cfg.clone(), e)
    m.get_vision_tower = lambda: object()
    # four feature entries: a row WITHOUT a sentinel still consumes one, as a zero-length slice (omchat_arch.py:122-129)
    feats = rnd(torch.randn(4, 1024, 3584, generator=torch.Generator().manual_seed(4)) * 0.3, dt)
    m.encode_images = lambda images: feats.to(DT[dt]).cuda()
    T = 32
    rows = [[5, -200, 6, -200] + list(range(10, 20)), [-200] + list(range(30, 60)), list(range(100, 109))]
    ids = torch.zeros(3, T, dtype=torch.long); mask = torch.zeros(3, T, dtype=torch.long)
    for i, r in enumerate(rows):
        if side == "left":
            ids[i, T - len(r):] = torch.tensor(r); mask[i, T - len(r):] = 1
        else:
            ids[i, :len(r)] = torch.tensor(r); mask[i, :len(r)] = 1
    dummy = torch.zeros(4, 3, 448, 448)
    out = m(input_ids=ids, attention_mask=mask, images=dummy, use_cache=True)
    kv = out.past_key_values
    # oracle: splice, padded prefill, decode branch
    emb_o, mask_sp, lengths = oracle.splice_inputs(ids, mask, [f for f in feats], sdt["model.embed_tokens.weight"], side, None)
    assert lengths == [2 * 1024 + 12, 1024 + 30, 9] and kv.get_seq_length() == emb_o.shape[1] == 2060
    cache = oracle.KVCache(2)
    h = oracle.qwen2_model(emb_o, sdt, cfg.text, cache, None, mask_sp)
    last = [n - 1 for n in lengths] if side == "right" else [emb_o.shape[1] - 1] * 3
    ref0 = torch.stack([oracle.lm_head(h[i:i + 1, last[i]:last[i] + 1], sdt)[0, 0] for i in range(3)])
    for i in range(3):
        assert rel(out.logits[i, 0], ref0[i]) < TOL_DEEP[dt], (side, "prefill", i, rel(out.logits[i, 0], ref0[i]))
    tok = torch.argmax(ref0, dim=-1)
    tok_mask = torch.cat([mask, torch.ones(3, 1, dtype=torch.long)], dim=1)
    _, pos1, mask1, _, _, _ = m.prepare_inputs_labels_for_multimodal(tok[:, None], None, tok_mask, kv, None, dummy)
    e.masked_decode_begin(pos1, mask1)
    for k in range(3):
        mo, po = oracle.decode_step_inputs(tok_mask, cache.get_seq_length())
        if k == 0:
            assert np.array_equal(mo.numpy(), mask1.numpy()) and np.array_equal(po.numpy(), pos1.numpy())
        ho = oracle.qwen2_model(sdt["model.embed_tokens.weight"][tok][:, None], sdt, cfg.text, cache, po, mo)
        ref = oracle.lm_head(ho, sdt)[:, -1]
        nxt, lg = e.decode_step_masked_next(tok, want_logits=True); sync()
        for i in range(3):
            assert rel(lg[i], ref[i]) < TOL_DEEP[dt], (side, k, i, rel(lg[i], ref[i]))
        tok = torch.argmax(ref, dim=-1)
        tok_mask = torch.cat([tok_mask, torch.ones(3, 1, dtype=torch.long)], dim=1)
    assert e.kv_lengths(3) == [2063] * 3
    e.close()


# ---------------------------------------------------------------------------------------------------------------------
# batched decode, seven launches per layer: un-split o_proj with the residual in its epilogue, the post-attention RMSNorm
# in the registers of the x-stationary gate|up GEMV (tuning key 14 bit 2)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("b", [32, 16, 3])
def test_batched_decode_with_the_norm_in_the_gate_up_gemv_vs_eight_launches_and_oracle(gpu_lib, dt, b):
    """one Qwen2-7B-width layer pair (3584, 28 / 4 heads, MLP 18944: the widths at which the x-stationary forms run), ragged batch: two decode
    steps with key 14 = 7 (seven launches) and 14 = 3 (o_proj split-K + residual + RMSNorm launch) against each other and against the
    oracle (transformers modeling_qwen2.py:269-298 restated in oracle/decoder.py)"""
    import oracle
    from omchat_amd.config import omchat13b
    if not gpu_lib.omchat_has_experiments():
        pytest.skip("measured slower than the eight launches (4.57-4.87 vs 4.28 ms per batch-32 step, DESIGN.md section 6): -DOMCHAT_EXPERIMENTS=1 builds only")
    cfg = omchat13b()
    cfg.text["num_hidden_layers"] = 2
    cfg.text["vocab_size"] = 2048
    keep = lambda k: not k.startswith(synth.TOWER) and "mm_projector" not in k
    sd = synth_state_dict(cfg, 9, keep)
    sdt = {k: rnd(torch.from_numpy(v), dt) for k, v in sd.items()}
    S = 40
    x = rnd(torch.randn(b, S, 3584, generator=torch.Generator().manual_seed(b)) * 0.5, dt)
    lens = [S - (i % 7) for i in range(b)]
    toks = torch.arange(b) % 2000 + 5
    res = {}
    try:
        for key in (7, 3):
            gpu_lib.omchat_op_set_tuning(14, key)
            e = Engine(cfg, dtype=dt, max_seq=S + 8, max_batch=b, max_tiles=1, vision=False)
            e.load_state_dict(sd)
            e.prefill(x, lens)
            nxt, lg = e.decode_step(toks, want_logits=True)
            nxt2, lg2 = e.decode_step(nxt, want_logits=True); sync()
            res[key] = (lg.cpu(), lg2.cpu(), nxt.cpu())
            e.close()
    finally:
        gpu_lib.omchat_op_set_tuning(14, 3)
    for i in sorted({0, b // 2, b - 1}):
        cache = oracle.KVCache(2)
        oracle.qwen2_model(x[i:i + 1, :lens[i]], sdt, cfg.text, cache)
        o1 = oracle.decode_step(toks[i:i + 1][None].long(), sdt, cfg.text, cache)[0, 0]
        o2 = oracle.decode_step(res[7][2][i:i + 1][None].long(), sdt, cfg.text, cache)[0, 0]
        for key in (7, 3):
            assert rel(res[key][0][i], o1) < TOL_DEEP[dt], (key, i, rel(res[key][0][i], o1))
        assert rel(res[7][1][i], o2) < TOL_DEEP[dt], (i, rel(res[7][1][i], o2))
    # the two structures sum o_proj's K in different orders (one slice against two) and form the variance in different orders
    assert rel(res[7][0], res[3][0]) < TOL[dt]


# ---------------------------------------------------------------------------------------------------------------------
# GEMM as two launches over disjoint column ranges (tile ids 12 / 13): whole rounds of 256^2 tiles + one round of 192 x 224 tiles
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("tile", [12, 13])
@pytest.mark.parametrize("M,N,K", [(3075, 12800, 192), (3075, 9600, 128), (1025, 3200, 256)])
def test_gemm_split_into_whole_rounds_and_a_tail_launch(gpu_lib, dt, tile, M, N, K):
    """the ViT fc1 shape (650 tiles = 2.54 rounds: 507 tiles of 256^2 + 221 of 192 x 224), the qkv shape, and a shape too small to split (falls back
    to the 256^2 kernel): every epilogue against the fp32 restatement, and bit-identical to the one-launch 256^2 kernel (each output element is
    accumulated over K in the same order by every tile shape)"""
    from test_gpu_ops import _gemm_ref
    A = rnd(randn((M, K), 1), dt); W = rnd(randn((N, K), 2, 0.05), dt)
    bias = rnd(randn((N,), 3, 0.1), dt); ls = rnd(randn((N,), 4, 0.1) + 0.1, dt); resid = rnd(randn((M, N), 5), dt)
    dA, dW, db, dl, dr = dev(A, dt), dev(W, dt), dev(bias, dt), dev(ls, dt), dev(resid, dt)
    for epi in (_lib.EPI_NONE, _lib.EPI_GELU, _lib.EPI_LS_RESID, _lib.EPI_RESID):
        use_bias = epi != _lib.EPI_RESID
        outs = []
        for t in (tile, 2):
            out = torch.full((M, N), float("nan"), dtype=DT[dt], device="cuda")
            _lib.check(gpu_lib.omchat_op_gemm(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(out), N, M, N, K, ptr(db) if use_bias else None,
                                              ptr(dl), ptr(dr), N, epi, t, None))
            sync()
            outs.append(out)
        ref = _gemm_ref(A, W, bias if use_bias else None, ls, resid, epi, dt)
        assert torch.isfinite(outs[0].float()).all(), (epi, "non-finite / unwritten outputs")
        assert rel(outs[0], ref) < TOL[dt], (epi, tile, rel(outs[0], ref))
        assert torch.equal(outs[0], outs[1]), (epi, tile)


# ---------------------------------------------------------------------------------------------------------------------
# ViT attention with the keys split between two wave groups of one workgroup (attn2_kernel KG = 2, tuning key 36)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("b,S,H,D,lens", [(3, 1025, 25, 128, None), (2, 300, 5, 128, None), (8, 1025, 16, 64, None), (1, 129, 9, 64, None), (2, 64, 3, 128, None),
                                        (3, 257, 4, 128, [257, 100, 33]), (1, 2000, 2, 128, None)])
def test_mha_prefill_attention_with_two_key_groups_vs_reference_and_one_group(gpu_lib, dt, b, S, H, D, lens):
    """eight waves on the same 128 queries, each half walking half of the key tiles, merged through LDS: against the fp32 softmax(QK^T)V and
    against the one-group kernel (same values up to where the online softmax is cut: one rescale more per query); odd tile counts (17, 5, 3),
    a single key tile (the second group has nothing), padded key lengths, both head dims, both grid forms (key 33)"""
    if not gpu_lib.omchat_has_experiments():
        pytest.skip("measured slower than two independent four-wave workgroups (73.4 vs 68.0 us at 3 tiles, DESIGN.md section 6): -DOMCHAT_EXPERIMENTS=1 builds only")
    q = rnd(randn((b, S, H, D), 1), dt); k = rnd(randn((b, H, S, D), 2), dt); v = rnd(randn((b, H, S, D), 3), dt)
    dq, dk, dv = dev(q, dt), dev(k, dt), dev(v, dt)
    dl = torch.tensor(lens, dtype=torch.int32, device="cuda") if lens else None
    outs = {}
    try:
        for kg in (2, 1):
            for k33 in (1, 0):
                gpu_lib.omchat_op_set_tuning(36, kg); gpu_lib.omchat_op_set_tuning(33, k33)
                o = torch.full((b, S, H, D), float("nan"), dtype=DT[dt], device="cuda")
                _lib.check(gpu_lib.omchat_op_attn_prefill_d(CODE[dt], ptr(dq), ptr(dk), ptr(dv), ptr(o), b, S, S, H, H, D, ptr(dl), 0, 0, D ** -0.5, None))
                sync()
                assert torch.isfinite(o.float()).all(), (kg, k33)
                outs[(kg, k33)] = o.clone()
    finally:
        gpu_lib.omchat_op_set_tuning(36, 0); gpu_lib.omchat_op_set_tuning(33, 1)
    assert torch.equal(outs[(2, 0)], outs[(2, 1)]) and torch.equal(outs[(1, 0)], outs[(1, 1)])
    sc = torch.einsum("bqhd,bhkd->bhqk", q.float(), k.float()) * D ** -0.5
    if lens:
        for i, n in enumerate(lens):
            sc[i, :, :, n:] = float("-inf")
    ref = torch.einsum("bhqk,bhkd->bqhd", torch.softmax(sc, -1), v.float())
    assert rel(outs[(2, 1)], ref) < TOL[dt] and rel(outs[(1, 1)], ref) < TOL[dt]
    assert rel(outs[(2, 1)], outs[(1, 1)]) < TOL[dt] / 2      # (round 6: the one-group form folds the odd key into its initial state, key 46)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("tile", [2, 9, 10, 1])
@pytest.mark.parametrize("M,N,K,ldc", [(515, 3000, 128, 3000), (300, 1000, 64, 1008), (1025, 3200, 128, 3200), (257, 1004, 64, 1004), (130, 520, 64, 528)])
def test_gemm_epilogue_wide_stores_same_bits_as_narrow(gpu_lib, dt, tile, M, N, K, ldc):
    """16-byte epilogue stores (tuning key 37 = 1: neighbouring column blocks exchanged between lane pairs) against the 8-byte form: the same values to the
    same places, bit for bit, for every epilogue -- N a multiple of 16, N = 8 mod 16 (the last block pair ends inside a block), N % 8 != 0 (falls back),
    padded row stride; nothing written outside [M, N] (the padding columns keep their sentinel)"""
    A = rnd(randn((M, K), 1), dt); W = rnd(randn((N, K), 2, 0.05), dt)
    bias = rnd(randn((N,), 3, 0.1), dt); ls = rnd(randn((N,), 4, 0.1) + 0.1, dt); resid = rnd(randn((M, ldc), 5), dt)
    dA, dW, db, dl, dr = dev(A, dt), dev(W, dt), dev(bias, dt), dev(ls, dt), dev(resid, dt)
    for epi in (_lib.EPI_NONE, _lib.EPI_GELU, _lib.EPI_LS_RESID, _lib.EPI_RESID):
        use_bias = epi != _lib.EPI_RESID
        outs = {}
        try:
            for key in (1, 0):
                gpu_lib.omchat_op_set_tuning(37, key)
                out = torch.full((M, ldc), 77.0, dtype=DT[dt], device="cuda")
                _lib.check(gpu_lib.omchat_op_gemm(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(out), ldc, M, N, K, ptr(db) if use_bias else None,
                                                  ptr(dl), ptr(dr), ldc, epi, tile, None))
                sync()
                outs[key] = out.clone()
        finally:
            gpu_lib.omchat_op_set_tuning(37, 1)
        assert torch.equal(outs[1], outs[0]), (epi, tile)
        assert bool((outs[1][:, N:] == 77.0).all()) and torch.isfinite(outs[1][:, :N].float()).all()


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("N,K", [(37888, 3584), (4736, 3584), (8192 + 64, 2048), (96, 1024 + 64)])
def test_gate_up_norm_gemv_pairs_per_wave_same_bits(gpu_lib, dt, N, K):
    """tuning key 38: the batch-1 gate|up GEMV with its RMSNorm in registers as 1 (default), 2 or 3 (gate, up) pairs per wave, and the loop form (key 16):
    one wave owns a pair's whole dot products in every form, so not a bit may differ; and against the fp32 restatement with Qwen2RMSNorm's rounding points
    (transformers modeling_qwen2.py:247-252 + :46-48)"""
    x = rnd(randn((K,), 1), dt); w = rnd(randn((N, K), 2, 0.05), dt); nw = rnd(randn((K,), 3, 0.1) + 1.0, dt)
    xn = rnd(nw * rnd(x * torch.rsqrt((x * x).mean() + 1e-6), dt), dt)
    blocks = (w @ xn).reshape(N // 32, 2, 16)
    ref = (rnd(torch.nn.functional.silu(rnd(blocks[:, 0], dt)), dt) * rnd(blocks[:, 1], dt)).reshape(N // 2)
    dx, dw, dn = dev(x, dt), dev(w, dt), dev(nw, dt)
    outs = []
    try:
        for k38, k16 in ((1, 0), (2, 0), (3, 0), (1, 1)):
            gpu_lib.omchat_op_set_tuning(38, k38); gpu_lib.omchat_op_set_tuning(16, k16)
            out = torch.full((N // 2,), float("nan"), dtype=DT[dt], device="cuda")
            _lib.check(gpu_lib.omchat_op_gemv_norm(CODE[dt], ptr(dx), ptr(dw), K, ptr(out), N, K, ptr(dn), 1e-6, None, _lib.EPI_SWIGLU, 0, None))
            sync()
            outs.append(out)
    finally:
        gpu_lib.omchat_op_set_tuning(38, 1); gpu_lib.omchat_op_set_tuning(16, 0)
    assert rel(outs[0], ref) < TOL[dt]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("N,K,in_place", [(3584, 18944, True), (256, 4608, True), (96, 8256, False), (7, 32768, True), (3000, 12352, True)])
def test_long_k_gemv_without_the_lds_stage_same_bits(gpu_lib, dt, N, K, in_place):
    """tuning key 39: the batch-1 down_proj GEMV (K in one piece, residual in the epilogue) with x loaded from L2 by every wave instead of staged in LDS,
    as workgroups of 1 / 2 / 4 waves: the same chunk order and dot products as gemv_rows_longk_kernel, so not a bit may differ; ragged K and N; and
    against fp32 torch (Qwen2MLP.down_proj + the residual add, transformers modeling_qwen2.py:41-48, :296-297).  Measured slower than the LDS form (DESIGN.md
    section 6, round 5), so the kernel is only in the -DOMCHAT_EXPERIMENTS=1 build"""
    if not gpu_lib.omchat_has_experiments():
        pytest.skip("gemv_rows_longk_direct_kernel is in the experiments build only")
    x = rnd(randn((K,), 1, 0.5), dt); w = rnd(randn((N, K), 2, 0.05), dt); r = rnd(randn((N,), 3), dt)
    ref = r + rnd(w @ x, dt)
    dx, dw = dev(x, dt), dev(w, dt)
    outs = []
    try:
        for k39 in (0, 1, 2, 4):
            gpu_lib.omchat_op_set_tuning(39, k39)
            dr = dev(r, dt)
            y = dr if in_place else torch.full((N,), float("nan"), dtype=DT[dt], device="cuda")
            _lib.check(gpu_lib.omchat_op_gemv(CODE[dt], ptr(dx), K, ptr(dw), K, ptr(y), N, 1, N, K, None, ptr(dr), N, _lib.EPI_RESID, 0, None))
            sync()
            outs.append(y)
    finally:
        gpu_lib.omchat_op_set_tuning(39, DEFAULT_KEY39)
    assert rel(outs[0], ref) < TOL[dt]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("rows,H", [(3075, 3200), (300, 3584), (257, 4096), (1025, 1024), (259, 72)])
def test_rmsnorm_one_wave_per_row_same_bits(gpu_lib, dt, rows, H):
    """tuning key 40: RMSNorm launches of >= 256 rows (prefill, ViT) as one wave per row -- a lane plays four threads of the workgroup-per-row kernel and
    the partial sums combine in that kernel's order, so not a bit may differ; against the oracle's rounding sequence (InternRMSNorm,
    modeling_intern_vit.py:39-44 == Qwen2RMSNorm, transformers modeling_qwen2.py:247-252); a ragged last workgroup (rows % 4) and H % 512 != 0.
    Measured no faster (these launches run at the chip's copy rate), so the wave kernels are in the -DOMCHAT_EXPERIMENTS=1 build only"""
    import oracle
    if not gpu_lib.omchat_has_experiments():
        pytest.skip("rmsnorm_wave_kernel is in the experiments build only")
    x = rnd(randn((rows, H), 1, 2.0), dt); w = rnd(randn((H,), 2, 0.1) + 1.0, dt)
    dx, dw = dev(x, dt), dev(w, dt)
    outs = []
    try:
        for k40 in (1, 0):
            gpu_lib.omchat_op_set_tuning(40, k40)
            out = torch.full((rows + 1, H), float("nan"), dtype=DT[dt], device="cuda")
            _lib.check(gpu_lib.omchat_op_rmsnorm(CODE[dt], ptr(dx), ptr(dw), ptr(out), rows, H, 1e-6, None))
            sync()
            outs.append(out)
    finally:
        gpu_lib.omchat_op_set_tuning(40, 0)
    assert torch.isnan(outs[0][rows].float()).all()          # nothing beyond the last row
    assert torch.equal(outs[0][:rows], outs[1][:rows])
    ref = oracle.rms_norm(x.to(DT[dt]), w.to(DT[dt]), 1e-6).float()
    assert rel(outs[0][:rows], ref) < 2e-3
    assert (outs[0][:rows].float().cpu() == ref).float().mean() > 0.98


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("rows,C", [(3075, 3200), (258, 1024), (1025, 384)])
def test_vit_qknorm_one_wave_per_row_same_bits(gpu_lib, dt, rows, C):
    """tuning key 40 for the ViT's q / k norm over all heads of a token (modeling_intern_vit.py:143-148), in place on the fused qkv rows: one wave per
    (row, q | k) against one workgroup, bit for bit; v untouched; against the oracle (experiments build only, as above)"""
    import oracle
    if not gpu_lib.omchat_has_experiments():
        pytest.skip("vit_qknorm_wave_kernel is in the experiments build only")
    qkv = rnd(randn((rows, 3 * C), 1), dt); wq = rnd(randn((C,), 2, 0.1) + 1, dt); wk = rnd(randn((C,), 3, 0.1) + 1, dt)
    dwq = dev(wq, dt); dwk = dev(wk, dt)
    scale = 128 ** -0.5
    outs = []
    try:
        for k40 in (1, 0):
            gpu_lib.omchat_op_set_tuning(40, k40)
            d = dev(qkv, dt)
            _lib.check(gpu_lib.omchat_op_vit_qknorm(CODE[dt], ptr(d), 3 * C, ptr(dwq), ptr(dwk), rows, C, C, 1e-6, scale, None))
            sync()
            outs.append(d)
    finally:
        gpu_lib.omchat_op_set_tuning(40, 0)
    assert torch.equal(outs[0], outs[1])
    T = DT[dt]
    q = (oracle.rms_norm(qkv[:, :C].to(T), wq.to(T), 1e-6) * scale).float()
    k = oracle.rms_norm(qkv[:, C:2 * C].to(T), wk.to(T), 1e-6).float()
    assert rel(outs[0][:, :C], q) < 3e-3 and rel(outs[0][:, C:2 * C], k) < 3e-3
    assert torch.equal(outs[0][:, 2 * C:].float().cpu(), qkv[:, 2 * C:])
