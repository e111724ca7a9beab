"""CPU: the oracle (oracle/) against golden vectors captured from the imported reference (tools/make_golden.py).

This is the pin that makes the oracle trustworthy; the GPU parity tests then compare the HIP path with the oracle."""
import numpy as np
import torch
import pytest
from conftest import golden, rel_err
import oracle
from oracle.pipeline import _sub, TOWER_PFX, PROJ_PFX
from omchat_amd import synth
from omchat_amd.config import tiny

torch.set_grad_enabled(False)
T = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt)
FP32_TOL = 2e-5      # same arithmetic, different summation order (unfold+matmul vs conv2d etc.)


def sd_torch(cfg, seed, prefix=None, dtype=torch.float32):
    return {k: T(v, dtype) for k, v in synth.state_dict(cfg, seed, prefix).items()}


def test_rmsnorm_3200():
    g = golden("rmsnorm_3200")
    y = oracle.rms_norm(T(g["x"]), T(g["w"]), 1e-6)
    assert np.array_equal(y.numpy(), g["y"])                      # identical op sequence -> bit exact
    yh = oracle.rms_norm(T(g["x"]).half(), T(g["w"]), 1e-6)
    assert np.array_equal(yh.float().numpy(), g["y_half"])


def test_vit_tiny_layers():
    g = golden("vit_tiny")
    cfg = tiny()
    w = _sub(sd_torch(cfg, int(g["seed"]), synth.TOWER), TOWER_PFX)
    px = T(g["pixels"])
    assert np.array_equal(px.numpy(), synth.pixels(2, 56, int(g["pixel_seed"])))
    emb = oracle.vit_embeddings(px, w, 14, 56)
    assert rel_err(emb, g["hs0"]) < FP32_TOL
    hs = oracle.vit_encoder(T(g["hs0"]), w, 2, 2)
    assert rel_err(hs[1], g["hs1"]) < FP32_TOL
    assert rel_err(hs[2], g["hs2"]) < FP32_TOL
    n1 = oracle.rms_norm(T(g["hs0"]), w["encoder.layers.0.norm1.weight"])
    assert rel_err(n1, g["l0_norm1"]) < 1e-6
    assert rel_err(oracle.vit_attention(n1, w, "encoder.layers.0.", 2), g["l0_attn"]) < FP32_TOL
    n2 = oracle.rms_norm(T(g["hs0"]), w["encoder.layers.0.norm2.weight"])
    assert rel_err(oracle.vit_mlp(n2, w, "encoder.layers.0."), g["l0_mlp"]) < FP32_TOL


@pytest.mark.parametrize("sel,feat", [(-1, "patch"), (-2, "patch"), (-1, "cls_patch")])
def test_tower_wrapper_select(sel, feat):
    g = golden(f"tower_wrapper_L{sel}_{feat}")
    cfg = tiny()
    w = _sub(sd_torch(cfg, 0, synth.TOWER), TOWER_PFX)
    # fp32 oracle vs the reference's fp16 plumbing: fp16 tolerance
    f32 = oracle.vision_tower_forward(T(g["pixels"]), w, cfg.vision, sel, feat)
    assert f32.shape == g["feats_half"].shape
    assert rel_err(f32, g["feats_half"]) < 3e-3
    # oracle run in fp16 like the reference
    wh = {k: v.half() for k, v in w.items()}
    f16 = oracle.vision_tower_forward(T(g["pixels"]).half(), wh, cfg.vision, sel, feat)
    assert f16.dtype == torch.float16
    assert rel_err(f16.float(), g["feats_half"]) < 2e-3


def test_vit300m_tiny_layers():
    """InternViT-300M variant (LayerNorm, 64-wide heads, no q/k norm) against the reference's intern_vit_300m modeling file"""
    from omchat_amd.config import tiny300m
    import torch.nn.functional as F
    g = golden("vit300m_tiny")
    cfg = tiny300m()
    w = _sub(sd_torch(cfg, int(g["seed"]), synth.TOWER), TOWER_PFX)
    assert "encoder.layers.0.norm1.bias" in w and "encoder.layers.0.attn.q_norm.weight" not in w
    px = T(g["pixels"])
    assert rel_err(oracle.vit_embeddings(px, w, 14, 56), g["hs0"]) < FP32_TOL
    hs = oracle.vit_encoder(T(g["hs0"]), w, 2, cfg.vision["num_attention_heads"])
    assert rel_err(hs[1], g["hs1"]) < FP32_TOL and rel_err(hs[2], g["hs2"]) < FP32_TOL
    n1 = F.layer_norm(T(g["hs0"]), (256,), w["encoder.layers.0.norm1.weight"], w["encoder.layers.0.norm1.bias"], 1e-6)
    assert rel_err(n1, g["l0_norm1"]) < 1e-6
    assert rel_err(oracle.vit_attention(n1, w, "encoder.layers.0.", 4), g["l0_attn"]) < FP32_TOL


@pytest.mark.parametrize("sel,feat", [(-1, "patch"), (-2, "cls_patch")])
def test_tower300m_wrapper_select(sel, feat):
    from omchat_amd.config import tiny300m
    g = golden(f"tower300m_wrapper_L{sel}_{feat}")
    cfg = tiny300m()
    w = _sub(sd_torch(cfg, 0, synth.TOWER), TOWER_PFX)
    f32 = oracle.vision_tower_forward(T(g["pixels"]), w, cfg.vision, sel, feat)
    assert f32.shape == g["feats_half"].shape and rel_err(f32, g["feats_half"]) < 3e-3
    wh = {k: v.half() for k, v in w.items()}
    f16 = oracle.vision_tower_forward(T(g["pixels"]).half(), wh, cfg.vision, sel, feat)
    assert f16.dtype == torch.float16 and rel_err(f16.float(), g["feats_half"]) < 2e-3


def test_vit_embed_bicubic_resize():
    g = golden("vit_embed_resize")
    cfg = tiny(image_size=112)
    w = _sub(sd_torch(cfg, int(g["seed"]), synth.TOWER), TOWER_PFX)
    emb = oracle.vit_embeddings(T(g["pixels"]), w, 14, 112)
    assert rel_err(emb, g["emb"]) < FP32_TOL


def test_vit_attention_full_width_25_heads():
    g = golden("vit_attn_full")
    P = "g.attnfull."
    w = {"attn.qkv.weight": T(synth.uniform(P + "qkv.weight", (9600, 3200))),
         "attn.q_norm.weight": T(synth.uniform(P + "q_norm.weight", (3200,), 0, 0.05, 1.0)),
         "attn.k_norm.weight": T(synth.uniform(P + "k_norm.weight", (3200,), 0, 0.05, 1.0)),
         "attn.proj.weight": T(synth.uniform(P + "proj.weight", (3200, 3200))),
         "attn.proj.bias": T(synth.uniform(P + "proj.bias", (3200,)))}
    y = oracle.vit_attention(T(g["x"]), w, "", 25)
    assert rel_err(y, g["y"]) < FP32_TOL


def test_projector():
    g = golden("projector_tiny")
    w = _sub(sd_torch(tiny(), 0, "model.mm_projector."), PROJ_PFX)
    assert rel_err(oracle.projector_forward(T(g["x"]), w), g["y"]) < FP32_TOL


@pytest.mark.parametrize("name", ["1x3", "2_uneven_right", "2_uneven_left", "noimage_row", "truncate"])
def test_splice(name):
    g = golden("splice_" + name)
    cfg = tiny()
    emb = T(synth.uniform("model.embed_tokens.weight", (cfg.text["vocab_size"], cfg.text["hidden_size"]), int(g["seed"])))
    mask = T(g["mask"], torch.long) if bool(g["has_mask"]) else None
    maxlen = None if int(g["maxlen"]) < 0 else int(g["maxlen"])
    feats = [f for f in T(g["feats"])]
    embeds, mask_out, lengths = oracle.splice_inputs(T(g["ids"], torch.long), mask, feats, emb, str(g["side"]), maxlen)
    assert np.array_equal(embeds.numpy(), g["embeds"])            # pure copy: bit exact
    if mask is None:
        assert mask_out is None and g["mask_out"].size == 0
    else:
        assert mask_out.dtype == torch.long
        assert np.array_equal(mask_out.numpy(), g["mask_out"])


def test_splice_decode_shortcircuit():
    g = golden("splice_decode_shortcircuit")
    m, pos = oracle.decode_step_inputs(T(g["mask_in"], torch.long), int(g["past_len"]))
    assert np.array_equal(m.numpy(), g["mask_out"])
    assert np.array_equal(pos.numpy(), g["position_ids"])


@pytest.mark.parametrize("name,q,kv", [("7q1kv", 7, 1), ("4q2kv", 4, 2)])
def test_decoder_prefill_and_decode(name, q, kv):
    g = golden("decoder_" + name)
    cfg = tiny(q_heads=q, kv_heads=kv)
    sd = sd_torch(cfg, int(g["seed"]))
    cache = oracle.KVCache(cfg.text["num_hidden_layers"])
    h = oracle.qwen2_model(T(g["x"]), sd, cfg.text, cache)
    logits = oracle.lm_head(h, sd)[0]
    assert rel_err(logits, g["prefill_logits"]) < FP32_TOL
    toks = []
    last = logits[-1]
    for s in range(len(g["tokens"])):
        nxt = int(torch.argmax(last.float())); toks.append(nxt)
        step = oracle.decode_step(torch.tensor([[nxt]]), sd, cfg.text, cache)
        assert rel_err(step[0, 0], g["step_logits"][s]) < FP32_TOL
        last = step[0, 0]
    assert toks == [int(t) for t in g["tokens"]]
    # hidden state after the first layer
    cache2 = oracle.KVCache(cfg.text["num_hidden_layers"])
    pos = torch.arange(g["x"].shape[1])[None]
    cos, sin = oracle.rope_cos_sin(pos, 128, 1e6, torch.float32)
    h1 = oracle.qwen2_layer(T(g["x"]), sd, 0, cfg.text, cos, sin, cache2)
    assert rel_err(h1[0], g["hs1"]) < FP32_TOL


def test_e2e_tiny_greedy():
    g = golden("e2e_tiny")
    cfg = tiny()
    sd = sd_torch(cfg, int(g["seed"]))
    px = T(synth.pixels(int(g["n_tiles"]), 56, int(g["pixel_seed"])))
    ids = T(g["ids"], torch.long)
    feats = oracle.encode_images(px, sd, cfg.vision)
    assert rel_err(feats, g["image_features_half"]) < 3e-3        # fp32 oracle vs fp16 reference plumbing
    logits, cache, lengths = oracle.prefill(ids, px, sd, cfg.vision, cfg.text)
    assert logits.shape[1] == int(g["prefill_len"]) == ids.shape[1] - 2 + 2 * cfg.num_image_tokens
    assert rel_err(logits[0, -1], g["prefill_logits_last"]) < 5e-3
    toks, margins = oracle.greedy_generate(ids, px, sd, cfg.vision, cfg.text, len(g["tokens"]))
    # greedy ids must agree wherever the reference's own top-1/top-2 margin is resolvable at fp16 precision
    ref = [int(t) for t in g["tokens"]]
    for i, (a, b, m) in enumerate(zip(toks, ref, g["margins"])):
        if a != b:
            assert m < 2e-2, (i, a, b, m)
            break


def test_int_tables():
    g = golden("int_tables")
    from omchat_amd.mm_utils import select_best_resolution, tokenizer_image_token
    pin = [[448, 896], [896, 448], [896, 896], [1344, 448], [448, 1344], [1344, 1344]]
    for s, b in zip(g["sizes"], g["best"]):
        assert tuple(select_best_resolution(tuple(int(x) for x in s), pin)) == tuple(int(x) for x in b)

    class Tok:
        bos_token_id = None
        def __call__(self, s):
            import types
            return types.SimpleNamespace(input_ids=[1000 + ord(ch) for ch in s])
    ids = tokenizer_image_token("<image>\npatch:<image>\npatch:<image>\nhello", Tok(), -200)
    assert ids == [int(x) for x in g["prompt_ids"]]


@pytest.mark.parametrize("side", ["left", "right"])
def test_padded_batch_prefill_and_decode_steps(side):
    """SURVEY 8 f-4 (omchat_arch.py:61-70,176-193): what the reference computes on the decode steps after a padded batch prefill --
    token-level mask padded with ones to the spliced cache length, position_ids = sum(mask) - 1 -- captured by tools/make_golden_r3.py.
    The oracle restates it exactly (mask / positions bit for bit, logits to fp32 noise), including the LEFT-padded case whose padded
    row is positioned BEFORE the tokens it follows (the evidence behind the product's refusal)."""
    g = golden("leftpad_decode")
    cfg = tiny()
    sd = sd_torch(cfg, int(g["seed"]))
    ids, mask = T(g["ids"], torch.long), T(g["mask"], torch.long)
    feats = [f for f in T(g["feats"])]
    embeds, mask_sp, lengths = oracle.splice_inputs(ids, mask, feats, sd["model.embed_tokens.weight"], side, None)
    assert lengths == [int(x) for x in g["lengths"]] and embeds.shape[1] == int(g[side + "_S"])
    cache = oracle.KVCache(cfg.text["num_hidden_layers"])
    oracle.qwen2_model(embeds, sd, cfg.text, cache, None, mask_sp)
    tok_mask = mask
    for k in range(int(g["steps"])):
        nxt = T(g[f"{side}_tok_{k}"], torch.long)
        tok_mask = torch.cat([tok_mask, torch.ones(2, 1, dtype=torch.long)], dim=1)
        m, pos = oracle.decode_step_inputs(tok_mask, cache.get_seq_length())
        assert np.array_equal(m.numpy(), g[f"{side}_dec_mask_{k}"]) and np.array_equal(pos.numpy(), g[f"{side}_dec_pos_{k}"])
        h = oracle.qwen2_model(sd["model.embed_tokens.weight"][nxt][:, None], sd, cfg.text, cache, pos, m)
        logits = oracle.lm_head(h, sd)[:, -1]
        assert rel_err(logits, g[f"{side}_logits_{k}"]) < FP32_TOL
    if side == "left":
        # the padded row (row 1) is rotated to a position before its own last prompt token, and its mask hides slots 4..9 of the
        # spliced cache while exposing the padded slots around them: there is no consistent computation to be identical to
        assert int(g["left_dec_pos_0"][1, 0]) < int(g["last_prefill_pos"]) < int(g["left_dec_pos_0"][0, 0])
        row1 = g["left_dec_mask_0"][1]
        S, n1 = int(g["left_S"]), int(g["lengths"][1])
        assert row1[:4].all() and not row1[4:10].any() and row1[10:S - n1].all()      # padded slots [0, S - n1) partly visible
