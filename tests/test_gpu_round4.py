"""Round-4 GPU tests (all through the C ABI):
  * the fused attention + merge + o_proj launch of the batch-1 decode step (csrc/experiments/fused_decode.hip) gives the BITS of the three launches it
    replaces, at the tiny geometries (ragged K chunk, one row per workgroup) and at the full Qwen2-7B width (28 / 4 heads, 3584 wide, two
    rows per wave), across a key-tile boundary and at 3 k keys; no hand-off timed out
  * padded batches decode as the REFERENCE computes them (omchat_arch.py:61-70): logits of the three steps after the right-
    AND the left-padded prefill of tests/golden/leftpad_decode.npz (captured from the reference, eager CPU attention), through
    omchat_decode_step_masked and through the drop-in generate()
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
from conftest import golden
from gpu_util import DT, TOL_DEEP, rnd, rel, sync
from omchat_amd import synth, _lib
from omchat_amd.config import tiny
from omchat_amd.engine import Engine

DTS = ["bf16", "f16"]
T32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()


def _decoder_sd(cfg, seed):
    return {k: v for k, v in synth.state_dict(cfg, seed).items() if not k.startswith(synth.TOWER) and "mm_projector" not in k}


GEOMS = {
    # name: (config kwargs, prefill length, decode steps, max_seq)
    "tiny_q4kv2": (dict(q_heads=4, kv_heads=2), 21, 6, 64),                       # K = 512: one chunk, one o_proj row per workgroup
    "tiny_q7kv1_tile_edge": (dict(q_heads=7, kv_heads=1), 61, 8, 128),              # K = 896: ragged second chunk; the steps cross key 64
    "qwen2_7b_width": (dict(q_heads=28, kv_heads=4, hidden_t=3584, mlp_t=1024, layers_t=2), 3000, 5, 3072),   # 14 rows per workgroup, 47 splits
    "qwen2_7b_width_short": (dict(q_heads=28, kv_heads=4, hidden_t=3584, mlp_t=1024, layers_t=2), 70, 4, 128),
    "qwen2_7b_layer": (dict(q_heads=28, kv_heads=4, hidden_t=3584, mlp_t=18944, layers_t=1), 3580, 6, 3600),    # the real MLP width, 56 -> 57 splits
    "tiny_long_mlp": (dict(q_heads=4, kv_heads=2, mlp_t=4608), 21, 5, 64),                                    # 9 activation chunks, 2 passes
}


MODES = {"layer": (1, 0), "attn_oproj": (0, 1), "six_launches": (0, 0)}      # tuning keys (23, 22)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("geom", list(GEOMS))
def test_one_launch_layer_and_fused_attention_give_the_bits_of_the_six_launches(gpu_lib, dt, geom):
    """decode_layer.hip (the whole layer as one launch, tuning key 23) and fused_decode.hip (attention + merge + o_proj as one launch, key
    22) against the six launches per layer: the same model, free-running greedy steps, logits compared BIT FOR BIT at every step"""
    if not gpu_lib.omchat_has_experiments():
        pytest.skip("round-4 one-launch experiments are not in the product build (-DOMCHAT_EXPERIMENTS=1)")
    kw, S, steps, max_seq = GEOMS[geom]
    cfg = tiny(**kw)
    sd = _decoder_sd(cfg, 5)
    H = cfg.text["hidden_size"]
    x = rnd(torch.randn(1, S, H, generator=torch.Generator().manual_seed(3)) * 0.5, dt)
    runs = {}
    try:
        for mode, (k23, k22) in MODES.items():
            _lib.check(gpu_lib.omchat_op_set_tuning(23, k23)); _lib.check(gpu_lib.omchat_op_set_tuning(22, k22))
            e = Engine(cfg, dtype=dt, max_seq=max_seq, max_batch=1, max_tiles=1, vision=False)
            e.load_state_dict(sd)
            e.prefill(x)
            tok, outs = torch.tensor([11]), []
            for _ in range(steps):
                nxt, lg = e.decode_step(tok, want_logits=True)
                outs.append(lg.float().cpu().clone())
                tok = nxt.cpu()
            sync()
            n, bits = e.fused_status()
            assert bits == 0, f"a hand-off of the fused launch timed out: {bits:#x}"
            assert n == (steps * cfg.text["num_hidden_layers"] if mode != "six_launches" else 0), (mode, n)
            runs[mode] = outs
            e.close()
    finally:
        _lib.check(gpu_lib.omchat_op_set_tuning(23, 0)); _lib.check(gpu_lib.omchat_op_set_tuning(22, 0))
    for mode in ("layer", "attn_oproj"):
        for s_, (a_, b_) in enumerate(zip(runs[mode], runs["six_launches"])):
            assert torch.isfinite(a_).all()
            assert torch.equal(a_, b_), (mode, geom, s_, rel(a_, b_))


def test_fused_launch_is_repeatable_and_race_screened(gpu_lib):
    """the same 40 decode steps twice at the full width (one-launch layers): every step's logits bit-identical between the runs (a hand-off
    that let a stale granule through, or a read before its sweep, would show up as a difference sooner or later), no time-out bit"""
    if not gpu_lib.omchat_has_experiments():
        pytest.skip("round-4 one-launch experiments are not in the product build (-DOMCHAT_EXPERIMENTS=1)")
    cfg = tiny(q_heads=28, kv_heads=4, hidden_t=3584, mlp_t=2048, layers_t=2)
    sd = _decoder_sd(cfg, 6)
    x = rnd(torch.randn(1, 500, 3584, generator=torch.Generator().manual_seed(4)) * 0.5, "bf16")
    outs = []
    _lib.check(gpu_lib.omchat_op_set_tuning(23, 1))
    try:
        e = Engine(cfg, dtype="bf16", max_seq=1024, max_batch=1, max_tiles=1, vision=False)
        e.load_state_dict(sd)
        for rep in range(2):
            e.prefill(x)
            tok, seq = torch.tensor([7]), []
            for _ in range(40):
                nxt, lg = e.decode_step(tok, want_logits=True)
                seq.append(lg.float().cpu().clone())
                tok = nxt.cpu()
            outs.append(seq)
        sync()
        n, bits = e.fused_status()
        assert bits == 0 and n == 2 * 40 * 2
        for a_, b_ in zip(*outs):
            assert torch.equal(a_, b_)
        e.close()
    finally:
        _lib.check(gpu_lib.omchat_op_set_tuning(23, 0))


# ---------------------------------------------------------------------------------------------------------------------
# padded batches, decoded as the reference does
# ---------------------------------------------------------------------------------------------------------------------
def _padded_model(dt, g):
    from omchat_amd.model.omchat_qwen2 import OmChatQwen2ForCausalLM
    cfg = tiny()
    e = Engine(cfg, dtype=dt, max_seq=128, max_batch=2, max_tiles=3)
    e.load_state_dict(synth.state_dict(cfg, int(g["seed"])))
    m = OmChatQwen2ForCausalLM(cfg.clone(), e)
    feats = rnd(T32(g["feats"]), dt)
    m.encode_images = lambda images: feats.to(DT[dt]).cuda()
    return cfg, e, m


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("side", ["right", "left"])
def test_padded_batch_decode_steps_match_the_reference(gpu_lib, dt, side):
    """omchat_arch.py:61-70 as HF generate drives it: after the padded prefill of two rows of different spliced length (40 and 19) the
    reference appends every row's token at the common cache slot, rotates it to sum(mask) - 1 ([40, 34] at step 0) and masks with the
    token-level mask padded with ones (hides slots 4..9 of row 1, exposes its padded slots).  Teacher-forced on the reference's ids, the
    logits of all three steps must match the reference's (fp32 CPU, eager attention) for BOTH padding sides -- the left one needs the
    prefill's fully masked rows to attend uniformly, as the eager backend does."""
    g = golden("leftpad_decode")
    cfg, e, m = _padded_model(dt, g)
    m.config.mm["tokenizer_padding_side"] = side
    ids, mask = torch.from_numpy(g["ids"]).long(), torch.from_numpy(g["mask"]).long()
    dummy = torch.zeros(3, 3, 56, 56)
    out = m(input_ids=ids, attention_mask=mask, images=dummy, use_cache=True)
    kv = out.past_key_values
    assert kv.get_seq_length() == int(g[side + "_S"])
    tok_mask = mask
    for k in range(int(g["steps"])):
        nxt = torch.from_numpy(g[f"{side}_tok_{k}"]).long()
        tok_mask = torch.cat([tok_mask, torch.ones(2, 1, dtype=torch.long)], dim=1)
        # the host mirror of the decode branch returns the reference's own integers ...
        _, pos_k, mask_k, _, emb_k, _ = m.prepare_inputs_labels_for_multimodal(nxt[:, None], None, tok_mask, kv, None, dummy)
        assert emb_k is None and np.array_equal(pos_k.numpy(), g[f"{side}_dec_pos_{k}"]) and np.array_equal(mask_k.numpy(), g[f"{side}_dec_mask_{k}"])
        # ... and the device step computed with them matches the reference's logits
        o = m(input_ids=nxt[:, None], attention_mask=tok_mask, past_key_values=kv, images=dummy, use_cache=True)
        sync()
        ref = T32(g[f"{side}_logits_{k}"])
        for i in range(2):
            assert rel(o.logits[i, 0], ref[i]) < TOL_DEEP[dt], (side, k, i, rel(o.logits[i, 0], ref[i]))
        assert kv.get_seq_length() == int(g[side + "_S"]) + k + 1
    # the per-sequence step must not be mixed in after masked steps
    with pytest.raises(ValueError, match="omchat_decode_step"):
        e.decode_step(torch.tensor([1, 2]))
    e.close()


@pytest.mark.parametrize("side", ["right", "left"])
def test_generate_on_a_padded_batch_follows_the_reference_calls(gpu_lib, side):
    """the drop-in generate() on the same batch: no refusal any more; its tokens are the argmax of the masked steps it drives (the same
    calls HF generate makes on the reference), and where the reference's own top-1 / top-2 margin is clear of the 16-bit noise they are
    the reference's ids"""
    g = golden("leftpad_decode")
    cfg, e, m = _padded_model("f16", g)
    m.config.mm["tokenizer_padding_side"] = side
    ids, mask = torch.from_numpy(g["ids"]).long(), torch.from_numpy(g["mask"]).long()
    out = m.generate(ids, images=torch.zeros(3, 3, 56, 56), attention_mask=mask, max_new_tokens=4)
    assert out.shape == (2, ids.shape[1] + 4)
    new = out[:, ids.shape[1]:]
    # token k + 1 is the argmax of the reference's step-k logits wherever that argmax is decisive
    for k in range(3):
        ref = T32(g[f"{side}_logits_{k}"])
        top2 = torch.topk(ref, 2, dim=-1).values
        for i in range(2):
            same_path = all(int(new[i, j]) == int(g[f"{side}_tok_{j}"][i]) for j in range(k + 1))      # still teacher-consistent
            if same_path and float(top2[i, 0] - top2[i, 1]) > 0.05:
                assert int(new[i, k + 1]) == int(torch.argmax(ref[i])), (side, k, i)
    e.close()


@pytest.mark.parametrize("dt", DTS)
def test_dynamic_gate_up_gives_the_bits_of_the_equal_share_form(gpu_lib, dt):
    """gemv_rows_norm_dyn_kernel (tuning key 24: the gate|up outputs of a batch-1 step dealt by atomic work counters, because the XCDs do not
    stream at the same rate) against the loop form with equal shares: the same bits at every step, and the counters are back at zero after
    every launch (the second, third ... step would otherwise start from a drained pool and produce garbage or hang)"""
    if not gpu_lib.omchat_has_experiments():
        pytest.skip("round-4 one-launch experiments are not in the product build (-DOMCHAT_EXPERIMENTS=1)")
    cfg = tiny(q_heads=28, kv_heads=4, hidden_t=3584, mlp_t=18944, layers_t=2)
    sd = _decoder_sd(cfg, 8)
    x = rnd(torch.randn(1, 100, 3584, generator=torch.Generator().manual_seed(5)) * 0.5, dt)
    runs = {}
    try:
        for key in (1, 0):
            _lib.check(gpu_lib.omchat_op_set_tuning(24, key))
            e = Engine(cfg, dtype=dt, max_seq=256, max_batch=1, max_tiles=1, vision=False)
            e.load_state_dict(sd)
            e.prefill(x)
            tok, outs = torch.tensor([3]), []
            for _ in range(12):
                nxt, lg = e.decode_step(tok, want_logits=True)
                outs.append(lg.float().cpu().clone())
                tok = nxt.cpu()
            sync()
            runs[key] = outs
            e.close()
    finally:
        _lib.check(gpu_lib.omchat_op_set_tuning(24, 1))
    for s_, (a_, b_) in enumerate(zip(runs[1], runs[0])):
        assert torch.isfinite(a_).all() and torch.equal(a_, b_), (s_, rel(a_, b_))


# ---------------------------------------------------------------------------------------------------------------------
# batched decode attention, LDS-DMA ring form (attention.hip: attn_decode_dma_kernel, tuning keys 25 / 26)
# ---------------------------------------------------------------------------------------------------------------------
def _attn_decode_ref(q, k, v, lens, scale):
    """fp32 restatement of eager_attention_forward (modeling_qwen2.py:150-172) for one new token per sequence, GQA by repetition"""
    b, Hq, _ = q.shape
    rep = Hq // k.shape[1]
    out = torch.zeros(b, Hq, 128)
    for i, n in enumerate(lens):
        kk = k[i, :, :n].float().repeat_interleave(rep, 0); vv = v[i, :, :n].float().repeat_interleave(rep, 0)
        p = torch.softmax(torch.einsum("hd,hnd->hn", q[i].float(), kk) * scale, -1)
        out[i] = torch.einsum("hn,hnd->hd", p, vv)
    return out


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("b,Hq,Hkv,cap,lens", [(3, 8, 2, 1024, [1000, 513, 64]), (2, 28, 4, 4096, [3700, 33]), (5, 7, 1, 512, [1, 31, 32, 97, 512])])
def test_batched_decode_attention_dma_ring_vs_reference_and_register_form(gpu_lib, dt, b, Hq, Hkv, cap, lens):
    """op level: ragged lengths incl. a single key, tile edges (31 / 32 / 33 keys), a poisoned cache tail that must never leak; every split
    size the slot count (key 26) produces; against the fp32 reference and against the register form (key 25 = 0) it replaces"""
    from gpu_util import dev, ptr, randn, CODE, TOL
    q = rnd(randn((b, Hq, 128), 1), dt); k = rnd(randn((b, Hkv, cap, 128), 2), dt); v = rnd(randn((b, Hkv, cap, 128), 3), dt)
    ref = _attn_decode_ref(q, k, v, lens, 128 ** -0.5)
    for i, n in enumerate(lens):
        k[i, :, n:] = float("nan"); v[i, :, n:] = float("nan")
    dq, dk, dv = dev(q, dt), dev(k, dt), dev(v, dt)
    L = max(lens)
    wsb = gpu_lib.omchat_op_attn_decode_ws(b, Hq, L)
    ws = torch.empty(wsb // 4 + 4, dtype=torch.float32, device="cuda")
    dl = torch.tensor(lens, dtype=torch.int32, device="cuda")
    outs = {}
    try:
        gpu_lib.omchat_op_set_tuning(10, 4)
        for dma, slots in ((0, 4), (1, 4), (1, 1), (1, 64)):
            gpu_lib.omchat_op_set_tuning(25, dma); gpu_lib.omchat_op_set_tuning(26, slots)
            out = torch.full((b, Hq, 128), float("nan"), dtype=DT[dt], device="cuda"); ws.fill_(float("nan"))
            _lib.check(gpu_lib.omchat_op_attn_decode(CODE[dt], ptr(dq), ptr(dk), ptr(dv), ptr(out), b, Hq, Hkv, cap, L, ptr(dl), 128 ** -0.5, ptr(ws), wsb, None))
            sync()
            assert torch.isfinite(out.float()).all(), (dma, slots)
            assert rel(out, ref) < TOL[dt], (dma, slots, rel(out, ref))
            outs[(dma, slots)] = out.float().cpu()
        assert rel(outs[(1, 4)], outs[(0, 4)]) < TOL[dt]
    finally:
        gpu_lib.omchat_op_set_tuning(10, 0); gpu_lib.omchat_op_set_tuning(25, 1); gpu_lib.omchat_op_set_tuning(26, 0)


@pytest.mark.parametrize("dt", DTS)
def test_batched_decode_attention_dma_ring_in_the_model(gpu_lib, dt):
    """inside the decode step the split that owns the new position rotates the fresh key row, patches it into its LDS image and appends k / v:
    ragged contexts whose new positions sit on the last row of a 32-key tile, the first row, mid-tile and in a one-tile split; 5 free-running
    steps; logits against the one-tile kernel (key 10 = 1) and the register multi-tile form (key 25 = 0), and the cache rows the steps
    appended must be the same bits in all three (the next step reads them)"""
    cfg = tiny(q_heads=4, kv_heads=2)
    sd = _decoder_sd(cfg, 5)
    b, S = 5, 200
    x = torch.randn(b, S, 256, generator=torch.Generator().manual_seed(11)) * 0.5
    lens = [200, 127, 129, 64, 31]
    runs = {}
    try:
        for name, (tpw, dma, slots) in {"one_tile": (1, 0, 4), "register": (4, 0, 4), "dma": (4, 1, 4), "dma_one_split": (4, 1, 1)}.items():
            gpu_lib.omchat_op_set_tuning(10, tpw); gpu_lib.omchat_op_set_tuning(25, dma); gpu_lib.omchat_op_set_tuning(26, slots)
            e = Engine(cfg, dtype=dt, max_seq=256, max_batch=b, max_tiles=1, vision=False)
            e.load_state_dict(sd)
            e.prefill(x, lens)
            tok = torch.arange(b) % 300 + 7
            seq = []
            for _ in range(5):
                tok, lg = e.decode_step(tok, want_logits=True)
                seq.append((tok.cpu().clone(), lg.float().cpu().clone()))
            sync()
            runs[name] = seq
            e.close()
    finally:
        gpu_lib.omchat_op_set_tuning(10, 0); gpu_lib.omchat_op_set_tuning(25, 1); gpu_lib.omchat_op_set_tuning(26, 0)
    for name in ("register", "dma", "dma_one_split"):
        for step in range(5):
            a_, r_ = runs[name][step][1], runs["one_tile"][step][1]
            assert torch.isfinite(a_).all()
            assert rel(a_, r_) < TOL_DEEP[dt], (name, step, rel(a_, r_))


# ---------------------------------------------------------------------------------------------------------------------
# GQA prefill attention: dispatch order and the head split of the heaviest causal blocks (attention.hip: attn2_kernel, tuning key 30)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("b,S,Hq,Hkv", [(1, 1000, 28, 4), (2, 1500, 14, 2), (1, 3584, 28, 4)])
def test_gqa_prefill_attention_head_split_of_the_heaviest_blocks_same_bits(gpu_lib, dt, b, S, Hq, Hkv):
    """the heaviest causal query blocks of a one-round launch are issued as two workgroups, each running half of the kv group's query heads:
    every (head, query block) is still computed by one wave with the same key tiles in the same order, so the output must not change by a
    bit -- split off (key 30 = 0), the launcher's own choice (-1), half of the ranks (key 30 = 1000), three ranks; ragged S (partial last
    block), two sequences in one launch; and against the fp32 reference"""
    from gpu_util import dev, ptr, randn, CODE, TOL
    import ctypes as C
    q = rnd(randn((b, S, Hq, 128), 1), dt); k = rnd(randn((b, Hkv, S, 128), 2), dt); v = rnd(randn((b, Hkv, S, 128), 3), dt)
    dq, dk, dv = dev(q, dt), dev(k, dt), dev(v, dt)
    outs = {}
    try:
        for hs in (0, -1, 1000, 3):
            ring = 0
            gpu_lib.omchat_op_set_tuning(30, hs)
            o = torch.full((b, S, Hq, 128), float("nan"), dtype=DT[dt], device="cuda")
            _lib.check(gpu_lib.omchat_op_attn_prefill(CODE[dt], ptr(dq), ptr(dk), ptr(dv), ptr(o), b, S, S, Hq, Hkv, None, 1, 0, 128 ** -0.5, None))
            sync()
            assert torch.isfinite(o.float()).all(), (hs, ring)
            outs[(hs, ring)] = o.clone()
    finally:
        gpu_lib.omchat_op_set_tuning(30, -1)
    for key in outs:
        assert torch.equal(outs[(0, 0)], outs[key]), key
    # reference on a slice of the rows (the full S x S fp32 softmax of 28 heads is slow on the host)
    rows = torch.tensor(sorted({0, 1, 31, 32, S // 2, S - 33, S - 1}))
    rep = Hq // Hkv
    for i in range(b):
        kk = k[i].float().repeat_interleave(rep, 0); vv = v[i].float().repeat_interleave(rep, 0)          # [Hq, S, 128]
        sc = torch.einsum("rhd,hnd->hrn", q[i, rows].float(), kk) * 128 ** -0.5
        mask = torch.arange(S)[None, None, :] > rows[None, :, None]
        p = torch.softmax(sc.masked_fill(mask, float("-inf")), -1)
        ref = torch.einsum("hrn,hnd->rhd", p, vv)
        assert rel(outs[(-1, 0)][i, rows.cuda()], ref) < TOL[dt]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("b,S,H,D", [(3, 1025, 25, 128), (2, 300, 5, 128), (8, 1025, 16, 64), (1, 129, 9, 64)])
def test_mha_prefill_attention_one_xcd_grid_same_bits(gpu_lib, dt, b, S, H, D):
    """the ViT attention's one-dimensional grid (query blocks of a (head, tile) pair 8 workgroup ids apart, the last group of eight pairs
    padded with workgroups that exit) against the (query block, head, tile) grid, tuning key 33: the same workgroups compute the same
    tiles, so not a bit may change -- head counts that are and are not multiples of 8, both head dims, a ragged last query block; and
    against the fp32 reference"""
    from gpu_util import dev, ptr, randn, CODE, TOL
    q = rnd(randn((b, S, H, D), 1), dt); k = rnd(randn((b, H, S, D), 2), dt); v = rnd(randn((b, H, S, D), 3), dt)
    dq, dk, dv = dev(q, dt), dev(k, dt), dev(v, dt)
    outs = {}
    try:
        for k33 in (1, 0):
            gpu_lib.omchat_op_set_tuning(33, k33)
            o = torch.full((b, S, H, D), float("nan"), dtype=DT[dt], device="cuda")
            _lib.check(gpu_lib.omchat_op_attn_prefill_d(CODE[dt], ptr(dq), ptr(dk), ptr(dv), ptr(o), b, S, S, H, H, D, None, 0, 0, D ** -0.5, None))
            sync()
            assert torch.isfinite(o.float()).all(), k33
            outs[k33] = o.clone()
    finally:
        gpu_lib.omchat_op_set_tuning(33, 1)
    assert torch.equal(outs[0], outs[1])
    p = torch.softmax(torch.einsum("bqhd,bhkd->bhqk", q.float(), k.float()) * D ** -0.5, -1)
    ref = torch.einsum("bhqk,bhkd->bqhd", p, v.float())
    assert rel(outs[1], ref) < TOL[dt]
