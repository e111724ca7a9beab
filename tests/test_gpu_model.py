"""GPU parity: the model-level C ABI (tower, projector, splice, prefill, decode) against the oracle on the same seeded
inputs and against the golden vectors captured from the reference."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
from conftest import golden
from gpu_util import DT, TOL, TOL_DEEP, dev, rnd, rel, sync, synth_state_dict
from omchat_amd import synth, _lib
from omchat_amd.config import tiny, omchat13b, OmChatConfig
from omchat_amd.engine import Engine
import oracle
from oracle.pipeline import _sub, TOWER_PFX, PROJ_PFX

DTS = ["bf16", "f16"]
T32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()


def sd32(cfg, seed, prefix=None):
    return {k: T32(v) for k, v in synth.state_dict(cfg, seed, prefix).items()}


@pytest.fixture(scope="module")
def tiny_engines(gpu_lib):
    out = {}
    for dt in DTS:
        e = Engine(tiny(), dtype=dt, max_seq=256, max_batch=2, max_tiles=3)
        e.load_state_dict(synth.state_dict(tiny(), 0))
        out[dt] = e
    yield out
    for e in out.values():
        e.close()


@pytest.mark.parametrize("dt", DTS)
def test_device_fill_equals_host_state_dict(gpu_lib, dt):
    """omchat_fill_synthetic (bench path) and load_state_dict(synth.state_dict) give identical model outputs"""
    cfg = tiny()
    a = Engine(cfg, dtype=dt, max_seq=64, max_tiles=2); a.fill_synthetic(0)
    b = Engine(cfg, dtype=dt, max_seq=64, max_tiles=2); b.load_state_dict(synth.state_dict(cfg, 0))
    px = T32(synth.pixels(2, 56, 3))
    fa, fb = a.encode_images(px), b.encode_images(px)
    sync()
    assert torch.equal(fa, fb)
    a.close(); b.close()


@pytest.mark.parametrize("dt", DTS)
def test_vit_tiny_vs_golden_and_oracle(tiny_engines, dt):
    e = tiny_engines[dt]
    g = golden("vit_tiny")
    px = T32(g["pixels"])
    for idx, key in ((0, "hs0"), (1, "hs1"), (2, "hs2")):
        out = e.vit_forward(px, select_layer=idx, select_feature="cls_patch")
        sync()
        assert out.shape == (2, 17, 256)
        assert rel(out, T32(g[key])) < TOL_DEEP[dt], (key, rel(out, T32(g[key])))
    last = e.vit_forward(px, select_layer=-1, select_feature="patch")
    sync()
    assert rel(last, T32(g["hs2"])[:, 1:]) < TOL_DEEP[dt]
    m2 = e.vit_forward(px, select_layer=-2)
    sync()
    assert rel(m2, T32(g["hs1"])[:, 1:]) < TOL_DEEP[dt]
    with pytest.raises(ValueError):
        e.vit_forward(px, select_feature="bogus")
    with pytest.raises(ValueError):
        e.vit_forward(px[0])


@pytest.mark.parametrize("dt", DTS)
def test_vit_batch_chunking(tiny_engines, dt):
    """n_tiles > max_tiles is processed in chunks with identical results"""
    e = tiny_engines[dt]
    px = T32(synth.pixels(7, 56, 11))
    full = e.vit_forward(px); sync()
    one = torch.cat([e.vit_forward(px[i:i + 1]) for i in range(7)]); sync()
    assert torch.equal(full, one)


@pytest.mark.parametrize("dt", DTS)
def test_projector_and_encode_images(tiny_engines, dt):
    e = tiny_engines[dt]
    g = golden("projector_tiny")
    out = e.projector_forward(T32(g["x"])); sync()
    assert rel(out, T32(g["y"])) < TOL_DEEP[dt]
    cfg = tiny()
    px = T32(synth.pixels(2, 56, 7))
    feats = e.encode_images(px); sync()
    ref = oracle.encode_images(px, sd32(cfg, 0), cfg.vision)
    assert feats.shape == (2, 16, 256)
    assert rel(feats, ref) < TOL_DEEP[dt]


@pytest.mark.parametrize("name", ["1x3", "2_uneven_right", "2_uneven_left", "noimage_row", "truncate"])
def test_splice_bit_exact_vs_golden(tiny_engines, name):
    e = tiny_engines["f16"]           # fixtures are exact in fp16
    g = golden("splice_" + name)
    mask = torch.from_numpy(g["mask"]).long() if bool(g["has_mask"]) else None
    maxlen = None if int(g["maxlen"]) < 0 else int(g["maxlen"])
    feats = rnd(T32(g["feats"]), "f16")
    embeds, lengths, valid = e.splice(torch.from_numpy(g["ids"]).long(), mask, dev(feats, "f16"), str(g["side"]), maxlen)
    sync()
    ref = rnd(T32(g["embeds"]), "f16")
    # embed rows are exact in fp16 (synthetic weights); feature rows were rounded to fp16 on both sides
    assert torch.equal(embeds.float().cpu(), ref)
    if mask is not None:
        assert np.array_equal(valid.numpy().astype(np.int64), g["mask_out"])


def test_splice_errors(tiny_engines):
    e = tiny_engines["f16"]
    ids = torch.tensor([[1, -200, 2, -200]])
    with pytest.raises(ValueError):
        e.splice(ids, None, torch.zeros(1, 16, 256))       # two sentinels, one tile


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("name,q,kv", [("7q1kv", 7, 1), ("4q2kv", 4, 2)])
def test_decoder_prefill_and_decode_vs_golden(gpu_lib, dt, name, q, kv):
    g = golden("decoder_" + name)
    cfg = tiny(q_heads=q, kv_heads=kv)
    e = Engine(cfg, dtype=dt, max_seq=64, max_batch=1, vision=False)
    e.load_state_dict(synth.state_dict(cfg, int(g["seed"])))
    x = T32(g["x"])
    logits, hidden = e.prefill(x, want_hidden=True)
    all_logits = e.lm_head(hidden); sync()
    ref = T32(g["prefill_logits"])
    assert rel(all_logits[0], ref) < TOL_DEEP[dt], rel(all_logits[0], ref)
    assert rel(logits[0], ref[-1]) < TOL_DEEP[dt]
    assert e.kv_lengths(1) == [x.shape[1]]
    # greedy decode driven with the GOLDEN tokens (teacher forcing keeps the comparison meaningful after a near-tie)
    toks = [int(t) for t in g["tokens"]]
    got = [int(torch.argmax(logits[0]))]
    for s, t in enumerate(toks):
        nxt, lg = e.decode_step(torch.tensor([t]), want_logits=True); sync()
        assert rel(lg[0], T32(g["step_logits"][s])) < TOL_DEEP[dt], (s, rel(lg[0], T32(g["step_logits"][s])))
        assert int(nxt[0]) == int(torch.argmax(lg[0]))
        got.append(int(nxt[0]))
    # ids must match wherever the fp32 reference margin exceeds the 16-bit noise floor
    step_ref = [ref[-1]] + [T32(g["step_logits"][s]) for s in range(len(toks))]
    for s, t in enumerate(toks):
        top2 = torch.topk(step_ref[s], 2).values
        if float(top2[0] - top2[1]) > (0.05 if dt == "bf16" else 0.01):
            assert got[s] == t, (s, got[s], t)
    assert e.kv_lengths(1) == [x.shape[1] + len(toks)]
    e.close()


@pytest.mark.parametrize("dt", DTS)
def test_prefill_right_padded_batch(gpu_lib, dt):
    cfg = tiny()
    e = Engine(cfg, dtype=dt, max_seq=64, max_batch=2, vision=False)
    e.load_state_dict(synth.state_dict(cfg, 5))
    x = rnd(torch.randn(2, 20, 256, generator=torch.Generator().manual_seed(0)), dt)
    lens = [20, 13]
    logits, _ = e.prefill(x, lengths=lens); sync()
    sd = sd32(cfg, 5)
    for i, n in enumerate(lens):
        h = oracle.qwen2_model(x[i:i + 1, :n], sd, cfg.text, oracle.KVCache(2))
        ref = oracle.lm_head(h, sd)[0, -1]
        assert rel(logits[i], ref) < TOL_DEEP[dt]
    # batched decode continues both sequences at their own positions
    nxt, lg = e.decode_step(torch.tensor([3, 4]), want_logits=True); sync()
    for i, (n, tok) in enumerate(zip(lens, [3, 4])):
        cache = oracle.KVCache(2)
        oracle.qwen2_model(x[i:i + 1, :n], sd, cfg.text, cache)
        ref = oracle.decode_step(torch.tensor([[tok]]), sd, cfg.text, cache)[0, 0]
        assert rel(lg[i], ref) < TOL_DEEP[dt]
    e.close()


@pytest.mark.parametrize("dt", DTS)
def test_e2e_tiny_vs_golden(tiny_engines, dt):
    """encode_images -> splice -> prefill -> greedy decode against the reference's fp16 plumbing run"""
    e = tiny_engines[dt]
    g = golden("e2e_tiny")
    cfg = tiny()
    px = T32(synth.pixels(int(g["n_tiles"]), 56, int(g["pixel_seed"])))
    e2 = Engine(cfg, dtype=dt, max_seq=128, max_batch=1, max_tiles=2)
    e2.load_state_dict(synth.state_dict(cfg, int(g["seed"])))
    feats = e2.encode_images(px); sync()
    assert rel(feats, T32(g["image_features_half"])) < TOL_DEEP[dt]
    embeds, lengths, _ = e2.splice(torch.from_numpy(g["ids"]).long(), None, feats)
    assert lengths == [int(g["prefill_len"])]
    logits, _ = e2.prefill(embeds, lengths); sync()
    assert rel(logits[0], T32(g["prefill_logits_last"])) < TOL_DEEP[dt]
    toks = [int(torch.argmax(logits[0]))]
    ref = [int(t) for t in g["tokens"]]
    for i in range(len(ref) - 1):
        nxt, _ = e2.decode_step(torch.tensor([ref[i]])); sync()
        toks.append(int(nxt[0]))
    thr = 0.05 if dt == "bf16" else 0.02
    for i, (a, b, m) in enumerate(zip(toks, ref, g["margins"])):
        if m > thr:
            assert a == b, (i, a, b, m)
    e2.close()


@pytest.mark.parametrize("dt", DTS)
def test_full_width_vit_layer(gpu_lib, dt):
    """one InternViT-6B-width layer (3200 / 25 heads / 12800), one 448 px tile = 1025 tokens, vs the fp32 oracle"""
    cfg = omchat13b()
    cfg.vision["num_hidden_layers"] = 1
    e = Engine(cfg, dtype=dt, max_tiles=1, text=False)
    sd = synth.state_dict(cfg, 0, synth.TOWER)
    sd.update(synth.state_dict(cfg, 0, "model.mm_projector."))
    e.load_state_dict(sd)
    px = T32(synth.pixels(1, 448, 2))
    out = e.vit_forward(px, select_layer=1, select_feature="cls_patch"); sync()
    w = _sub({k: T32(v) for k, v in sd.items()}, TOWER_PFX)
    emb = oracle.vit_embeddings(px, w, 14, 448)
    ref = oracle.vit_layer(emb, w, 0, 25)
    assert out.shape == (1, 1025, 3200)
    assert rel(out, ref) < TOL_DEEP[dt], rel(out, ref)
    e.close()


@pytest.mark.parametrize("dt", ["bf16"])
def test_full_width_decoder_layer(gpu_lib, dt):
    """one Qwen2-7B-width layer (3584, 28/4 heads, 18944), S = 300 prefill + 2 decode steps, vocab cut to 2048"""
    cfg = omchat13b()
    cfg.text["num_hidden_layers"] = 1
    cfg.text["vocab_size"] = 2048
    e = Engine(cfg, dtype=dt, max_seq=512, max_batch=1, vision=False)
    sd = synth_state_dict(cfg, 0, lambda k: not k.startswith(synth.TOWER) and "mm_projector" not in k)
    e.load_state_dict(sd)
    x = rnd(torch.randn(1, 300, 3584, generator=torch.Generator().manual_seed(1)) * 0.5, dt)
    logits, hidden = e.prefill(x, want_hidden=True); sync()
    sdt = {k: T32(v) for k, v in sd.items()}
    cache = oracle.KVCache(1)
    h = oracle.qwen2_model(x, sdt, cfg.text, cache)
    assert rel(hidden, h) < TOL_DEEP[dt], rel(hidden, h)
    assert rel(logits[0], oracle.lm_head(h, sdt)[0, -1]) < TOL_DEEP[dt]
    for tok in (5, 9):
        nxt, lg = e.decode_step(torch.tensor([tok]), want_logits=True); sync()
        ref = oracle.decode_step(torch.tensor([[tok]]), sdt, cfg.text, cache)[0, 0]
        assert rel(lg[0], ref) < TOL_DEEP[dt]
    e.close()


def test_engine_argument_errors(gpu_lib):
    cfg = tiny()
    e = Engine(cfg, dtype="bf16", max_seq=32, max_batch=1, vision=False)
    with pytest.raises(KeyError):
        e.load_state_dict({"model.norm.weight": np.ones(256, np.float32)})      # strict: missing tensors
    with pytest.raises(ValueError):
        e.load_tensor("model.norm.weight", np.ones(7, np.float32))               # shape mismatch
    with pytest.raises(ValueError):
        e.load_tensor("no.such.tensor", np.ones(7, np.float32))
    e.close()


@pytest.mark.parametrize("dt", DTS)
def test_batched_decode_beyond_16_sequences(gpu_lib, dt):
    """b = 20 right-padded sequences (BASELINE configs[2] decodes 32 at once): one weight pass per step, every row equals
    the oracle run of that sequence alone"""
    cfg = tiny()
    b = 20
    e = Engine(cfg, dtype=dt, max_seq=48, max_batch=b, vision=False)
    e.load_state_dict(synth.state_dict(cfg, 5))
    x = rnd(torch.randn(b, 16, 256, generator=torch.Generator().manual_seed(0)), dt)
    lens = [16 - (i % 5) for i in range(b)]
    logits, _ = e.prefill(x, lengths=lens); sync()
    toks = torch.arange(3, 3 + b)
    nxt, lg = e.decode_step(toks, want_logits=True); sync()
    nxt2, lg2 = e.decode_step(nxt, want_logits=True); sync()
    sd = sd32(cfg, 5)
    for i in (0, 7, 15, 16, 19):
        cache = oracle.KVCache(cfg.text["num_hidden_layers"])
        h = oracle.qwen2_model(x[i:i + 1, :lens[i]], sd, cfg.text, cache)
        assert rel(logits[i], oracle.lm_head(h, sd)[0, -1]) < TOL_DEEP[dt]
        ref = oracle.decode_step(torch.tensor([[int(toks[i])]]), sd, cfg.text, cache)[0, 0]
        assert rel(lg[i], ref) < TOL_DEEP[dt], (i, rel(lg[i], ref))
        assert int(nxt[i]) == int(torch.argmax(lg[i]))
        ref2 = oracle.decode_step(torch.tensor([[int(nxt[i])]]), sd, cfg.text, cache)[0, 0]
        assert rel(lg2[i], ref2) < TOL_DEEP[dt]
    assert e.kv_lengths(b) == [n + 2 for n in lens]
    e.close()


@pytest.mark.parametrize("S", [1, 2, 5, 63, 64, 65, 127, 129, 257])
def test_prefill_ragged_lengths_then_decode(gpu_lib, S):
    """prompt lengths around the 64-key / 128-query / 256-row tile edges (and the degenerate 1-token prompt): prefill logits and
    two decode steps against the oracle"""
    cfg = tiny()
    dt = "bf16"
    e = Engine(cfg, dtype=dt, max_seq=S + 8, max_batch=1, vision=False)
    e.load_state_dict(synth.state_dict(cfg, 9))
    sd = sd32(cfg, 9)
    x = rnd(torch.randn(1, S, 256, generator=torch.Generator().manual_seed(S)) * 0.5, dt)
    logits, hidden = e.prefill(x, want_hidden=True); sync()
    cache = oracle.KVCache(cfg.text["num_hidden_layers"])
    h = oracle.qwen2_model(x, sd, cfg.text, cache)
    assert rel(hidden, h) < TOL_DEEP[dt], rel(hidden, h)
    assert rel(logits[0], oracle.lm_head(h, sd)[0, -1]) < TOL_DEEP[dt]
    for tok in (4, 17):
        nxt, lg = e.decode_step(torch.tensor([tok]), want_logits=True); sync()
        ref = oracle.decode_step(torch.tensor([[tok]]), sd, cfg.text, cache)[0, 0]
        assert rel(lg[0], ref) < TOL_DEEP[dt], rel(lg[0], ref)
    assert e.kv_lengths(1) == [S + 2]
    e.close()


@pytest.mark.parametrize("dt", DTS)
def test_pos_embed_resize_path_vs_reference_golden(gpu_lib, dt):
    """checkpoint trained on a 112 px grid (8 x 8 patches) served at 56 px tiles (4 x 4): the position table is bicubic-resized at
    load (InternVisionEmbeddings._get_pos_embed, modeling_intern_vit.py:82-88); embeddings against the reference's own output"""
    g = golden("vit_embed_resize")
    cfg_ckpt = tiny(image_size=112)
    cfg_run = tiny(image_size=56)
    sd = synth.state_dict(cfg_ckpt, int(g["seed"]))
    assert sd[synth.TOWER + "embeddings.position_embedding"].shape[1] == 65
    e = Engine(cfg_run, dtype=dt, max_seq=32, max_tiles=1, text=False)
    e.load_state_dict(sd)
    emb = e.vit_forward(T32(g["pixels"]), select_layer=0, select_feature="cls_patch"); sync()
    assert emb.shape == (1, 17, 256) and rel(emb, T32(g["emb"])) < TOL[dt], rel(emb, T32(g["emb"]))
    e.close()
