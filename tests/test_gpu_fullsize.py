"""GPU, BASELINE.json full sizes (OmChat-13B geometry: InternViT-6B 45 layers + Qwen2-7B 28 layers, 27 GB of synthetic
weights generated on the device).  The oracle cannot finish these sizes in seconds, so parity is checked through
size-independent properties of the path:
  * determinism: the same inputs give bit-identical features / logits / tokens on repeated runs
  * KV-cache consistency: prefill(S) + decode(token) == last position of prefill(S + 1)   (prefill and decode are different
    kernels: MFMA GEMM + causal flash vs weight-streaming GEMV + split-KV attention with fused RoPE / append)
  * batch independence of the tower: features of tile i do not depend on which other tiles share the launch
  * splice is a pure copy: spliced rows equal the feature / embedding rows bit for bit
  * tensor sanity: finite outputs, softmax-normalised attention implied by bounded activations
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
from gpu_util import rel, sync, synth_state_dict
from omchat_amd import synth
from omchat_amd.config import omchat13b
from omchat_amd.engine import Engine


@pytest.fixture(scope="module", params=["bf16", "f16"])
def eng13b(gpu_lib, request):
    cfg = omchat13b()
    e = Engine(cfg, dtype=request.param, max_seq=1400, max_batch=1, max_tiles=3, max_prefill_rows=1400)
    e.fill_synthetic(0)
    e.dt_name = request.param
    yield e
    e.close()


# prefill and decode accumulate in different orders and round activations to 16 bit after every op of 28 layers: the
# agreement scales with the mantissa (bf16 8 bits, f16 11 bits) -- a logic error would not
CONSIST_TOL = {"bf16": 6e-2, "f16": 1e-2}


def test_full_size_tower_determinism_and_batch_independence(eng13b):
    px = torch.from_numpy(synth.pixels(3, 448, 0))
    a = eng13b.encode_images(px); sync()
    b = eng13b.encode_images(px); sync()
    assert a.shape == (3, 1024, 3584) and torch.isfinite(a.float()).all()
    assert torch.equal(a, b)
    one = eng13b.encode_images(px[1:2]); sync()
    # The same tile in another batch.  The GEMMs tile over rows and the attention over (tile, head), so every dot product is summed in the same
    # order whatever the batch; since round 6 the statistics of the folded norms are the exception: a row's sum of squares is added up in the
    # wave-tile grouping of the tile kernel tuned for the problem SIZE (gemm.hip gemm_epilogue), so rstd may differ in its last fp32 bit
    # between a 1-tile and a 3-tile launch and 45 layers of 16-bit rounding amplify that to the noise of one evaluation (f16 measured 2e-3;
    # the reference's own GEMM library picks kernels per shape in the same way).  A tile leaking into another would be O(1).
    assert rel(one[0], a[1]) < {"f16": 5e-3, "bf16": 4e-2}[eng13b.dt_name]
    # ... and inside ONE launch the tiles are independent bit for bit: the same picture three times gives three identical feature blocks
    same = eng13b.encode_images(px[1:2].repeat(3, 1, 1, 1)); sync()
    assert torch.equal(same[0], same[1]) and torch.equal(same[0], same[2])


def test_full_size_splice_prefill_decode_consistency(eng13b):
    cfg = omchat13b()
    px = torch.from_numpy(synth.pixels(1, 448, 5))
    feats = eng13b.encode_images(px)
    text = synth.token_ids(200, 151643, 3).tolist()
    ids = torch.tensor([[text[0], -200] + text[1:]])
    embeds, lengths, valid = eng13b.splice(ids, None, feats); sync()
    S = lengths[0]
    assert S == 200 + 1024 and bool(valid.all())
    # pure copies (omchat_arch.py:133-158)
    assert torch.equal(embeds[0, 1:1025], feats[0])
    # prefill S tokens, then decode token t  vs  prefill S+1 tokens whose last row is embed(t)
    logits_a, _ = eng13b.prefill(embeds, [S]); sync()
    tok = int(torch.argmax(logits_a[0]))
    nxt, step_logits = eng13b.decode_step(torch.tensor([tok]), want_logits=True); sync()
    ids2 = torch.cat([ids, torch.tensor([[tok]])], dim=1)
    embeds2, lengths2, _ = eng13b.splice(ids2, None, feats)
    assert torch.equal(embeds2[0, :S], embeds[0])
    logits_b, _ = eng13b.prefill(embeds2, [S + 1]); sync()
    assert torch.isfinite(step_logits).all() and torch.isfinite(logits_b).all()
    assert rel(step_logits[0], logits_b[0]) < CONSIST_TOL[eng13b.dt_name], rel(step_logits[0], logits_b[0])
    print('prefill/decode consistency', eng13b.dt_name, rel(step_logits[0], logits_b[0]))
    top2 = torch.topk(logits_b[0], 2).values
    if float(top2[0] - top2[1]) > 0.05:
        assert int(nxt[0]) == int(torch.argmax(logits_b[0]))
    # determinism of the whole decoder
    logits_c, _ = eng13b.prefill(embeds2, [S + 1]); sync()
    assert torch.equal(logits_b, logits_c)


def test_full_size_greedy_run_is_reproducible(eng13b):
    px = torch.from_numpy(synth.pixels(1, 448, 9))
    feats = eng13b.encode_images(px)
    ids = torch.tensor([[5, -200, 7, 8, 9]])
    outs = []
    for _ in range(2):
        embeds, lengths, _ = eng13b.splice(ids, None, feats)
        logits, _ = eng13b.prefill(embeds, lengths)
        tok = eng13b.argmax(logits)
        seq = [int(tok[0])]
        for _ in range(12):
            tok, _ = eng13b.decode_step(tok)
            seq.append(int(tok[0]))
        outs.append(seq)
    assert outs[0] == outs[1]
    assert eng13b.kv_lengths(1) == [4 + 1024 + 12]


def test_full_size_configs1_shape_3_tiles_S3584(gpu_lib):
    """the BENCHMARKED shape of BASELINE configs[1] (3 anyres tiles + 512 text ids -> S = 3584, 45 + 28 layers): tower batch
    independence at 3 tiles, splice = pure copy, prefill(S) + decode(t) == last position of prefill(S + 1), reproducible greedy run"""
    cfg = omchat13b()
    S = 3 * 1024 + 512
    e = Engine(cfg, dtype="bf16", max_seq=S + 40, max_batch=1, max_tiles=3, max_prefill_rows=S + 8)
    e.fill_synthetic(0)
    px = torch.from_numpy(synth.pixels(3, 448, 0))
    feats = e.encode_images(px); sync()
    assert feats.shape == (3, 1024, 3584) and torch.isfinite(feats.float()).all()
    assert rel(e.encode_images(px[2:3])[0], feats[2]) < 1e-6
    text = synth.token_ids(512, 151643, 1).tolist()
    ids = torch.tensor([[-200, text[0], -200, text[1], -200] + text[2:]])
    embeds, lengths, valid = e.splice(ids, None, feats); sync()
    assert lengths == [S] and bool(valid.all())
    assert torch.equal(embeds[0, :1024], feats[0]) and torch.equal(embeds[0, 1025:2049], feats[1]) and torch.equal(embeds[0, 2050:3074], feats[2])
    logits_a, _ = e.prefill(embeds, [S]); sync()
    tok = int(torch.argmax(logits_a[0]))
    nxt, step_logits = e.decode_step(torch.tensor([tok]), want_logits=True); sync()
    seq = [tok, int(nxt[0])]
    t2 = nxt
    for _ in range(6):
        t2, _ = e.decode_step(t2)
        seq.append(int(t2[0]))
    ids2 = torch.cat([ids, torch.tensor([[tok]])], dim=1)
    embeds2, lengths2, _ = e.splice(ids2, None, feats)
    logits_b, _ = e.prefill(embeds2, [S + 1]); sync()
    r = rel(step_logits[0], logits_b[0])
    print("configs1 shape prefill/decode consistency", r)
    assert torch.isfinite(logits_b).all() and r < CONSIST_TOL["bf16"], r
    # the same greedy run again: bit-identical ids (no race anywhere in ~75 k kernel launches)
    logits_c, _ = e.prefill(embeds, [S])
    assert torch.equal(logits_c, logits_a)
    t3 = e.argmax(logits_c)
    seq2 = [int(t3[0])]
    for _ in range(7):
        t3, _ = e.decode_step(t3)
        seq2.append(int(t3[0]))
    assert seq2 == seq
    e.close()


def test_full_size_batch32_rows_equal_single_sequence_runs(gpu_lib):
    """BASELINE configs[2] batch (32 sequences, full OmChat-13B width, ragged lengths, right-padded): every row of the batched prefill
    and of the batched decode steps must equal the run of that sequence alone.  Prefill rows are the SAME arithmetic (the GEMM tiles
    over rows with one K order, attention per (sequence, head)): logits agree to fp32 rounding of the lm_head kernels; decode uses
    the two-tile MFMA GEMV at b = 32 and the whole-row form at b = 1 (different K split): 16-bit rounding tolerance"""
    cfg = omchat13b()
    b = 32
    lens = [1024 + 40 + 13 * i for i in range(b)]                       # 1 tile + 40 .. 443 text ids
    S = max(lens)
    e = Engine(cfg, dtype="bf16", max_seq=S + 16, max_batch=b, max_tiles=8, max_prefill_rows=b * S)
    e.fill_synthetic(0)
    feats = e.encode_images(torch.from_numpy(synth.pixels(4, 448, 2)))     # 4 distinct tiles shared round-robin by the 32 samples
    rows, mask = [], []
    for i, n in enumerate(lens):
        text = synth.token_ids(n - 1024, 151643, 10 + i).tolist()
        r = [text[0], -200] + text[1:]
        rows.append(r + [0] * (S - 1024 + 1 - len(r))); mask.append([1] * len(r) + [0] * (S - 1024 + 1 - len(r)))
    ids, am = torch.tensor(rows), torch.tensor(mask)
    tile_of = [i % 4 for i in range(b)]
    embeds, lengths, valid = e.splice(ids, am, feats[tile_of]); sync()
    assert lengths == lens and embeds.shape == (b, S, 3584)
    logits, _ = e.prefill(embeds, lengths); sync()
    toks = e.argmax(logits)
    steps = []
    t = toks
    for _ in range(3):
        t, lg = e.decode_step(t, want_logits=True)
        steps.append((t.clone(), lg.clone()))
    sync()
    assert e.kv_lengths(b) == [n + 3 for n in lens]
    for i in (0, 7, 19, 31):
        l1, _ = e.prefill(embeds[i:i + 1, :lens[i]].contiguous(), [lens[i]]); sync()
        assert rel(l1[0], logits[i]) < 1e-4, (i, rel(l1[0], logits[i]))
        assert int(torch.argmax(l1[0])) == int(toks[i])
        t1 = toks[i:i + 1]
        for s_, (tb, lb) in enumerate(steps):
            t1n, lg1 = e.decode_step(t1, want_logits=True); sync()
            r = rel(lg1[0], lb[i])
            assert r < CONSIST_TOL["bf16"], (i, s_, r)
            t1 = tb[i:i + 1]                                            # follow the batched run's ids (teacher forcing)
    e.close()


def test_long_context_decode_16k_one_full_width_layer(gpu_lib):
    """BASELINE configs[4] shape on one Qwen2-7B-width layer: 16 k tokens of context in the KV cache (prefill in one pass), then
    decode steps with the 16-bit weights against the oracle (257 split-KV partials per head merged).  The oracle's eager
    attention cannot materialise 16 k x 16 k scores, so its KV cache is built directly (layer 0's K / V depend only on the
    input rows: norm -> k/v projection -> RoPE) and only the decode steps (1 x 16 k scores) go through it."""
    from omchat_amd.config import omchat13b
    from oracle import KVCache, decode_step
    from oracle.decoder import rope_cos_sin, apply_rope
    from oracle.vit import rms_norm
    import numpy as np
    import torch.nn.functional as F
    T32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
    cfg = omchat13b()
    cfg.text["num_hidden_layers"] = 1
    cfg.text["vocab_size"] = 2048
    S = 16400
    e = Engine(cfg, dtype="bf16", max_seq=S + 64, max_batch=1, vision=False)
    sd = {k: T32(v) for k, v in synth_state_dict(cfg, 0, lambda k: not k.startswith(synth.TOWER) and "mm_projector" not in k).items()}
    e.load_state_dict(sd)
    x = (torch.randn(1, S, 3584, generator=torch.Generator().manual_seed(1)) * 0.5).bfloat16().float()
    logits, _ = e.prefill(x); torch.cuda.synchronize()
    assert torch.isfinite(logits).all()
    P = "model.layers.0."
    xn = rms_norm(x, sd[P + "input_layernorm.weight"], 1e-6)
    k = F.linear(xn, sd[P + "self_attn.k_proj.weight"], sd[P + "self_attn.k_proj.bias"]).view(1, S, 4, 128).transpose(1, 2)
    v = F.linear(xn, sd[P + "self_attn.v_proj.weight"], sd[P + "self_attn.v_proj.bias"]).view(1, S, 4, 128).transpose(1, 2)
    cos, sin = rope_cos_sin(torch.arange(S)[None], 128, cfg.text["rope_theta"], torch.float32)
    _, k = apply_rope(k, k, cos, sin)
    cache = KVCache(1)
    cache.update(k, v, 0)
    rel = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())
    for tok in (5, 9):
        nxt, lg = e.decode_step(torch.tensor([tok]), want_logits=True); torch.cuda.synchronize()
        r = decode_step(torch.tensor([[tok]]), sd, cfg.text, cache)[0, 0]
        assert rel(lg[0], r) < 3e-2, rel(lg[0], r)
        assert int(nxt[0]) == int(torch.argmax(lg[0]))
    assert e.kv_lengths(1) == [S + 2] and cache.get_seq_length() == S + 2
    e.close()


def test_full_size_configs4_whole_model_in_the_fp8_modes(gpu_lib):
    """BASELINE configs[4] as a WHOLE model (VERDICT r03 missing #4): one 32-frame clip = 32 tiles through InternViT-6B, 32 sentinels + 512
    text ids -> S = 33 280 prefill positions through all 28 Qwen2-7B layers, decode at 33 k keys -- once with 16-bit operands, once with
    every fp8 mode on (e4m3 decode weights, e4m3 KV cache, fp8 x fp8 MFMA qkv / gate|up prefill GEMMs).  fp8 has no reference
    counterpart, so the bar is: finite, deterministic, KV lengths right, and the quantised run stays within a stated distance of the
    16-bit run of the same context (last-position prefill logits and three teacher-forced decode steps); greedy ids equal wherever the
    16-bit top-1 / top-2 margin exceeds the quantisation noise measured on these very logits."""
    cfg = omchat13b()
    n_tiles, n_text, steps = 32, 512, 3
    S = n_tiles * 1024 + n_text
    e = Engine(cfg, dtype="bf16", max_seq=S + 16, max_batch=1, max_tiles=n_tiles, max_prefill_rows=S)
    e.fill_synthetic(0)
    px = torch.from_numpy(synth.pixels(n_tiles, 448, 0)).to("cuda", torch.bfloat16)
    text = synth.token_ids(n_text, 151643, 1).tolist()
    row = []
    for t in range(n_tiles):
        row += [-200, text[t]]
    ids = torch.tensor([row[:-1] + text[n_tiles - 1:]], dtype=torch.int64)
    feats = e.encode_images(px)
    embeds, lengths, _ = e.splice(ids, None, feats)
    assert lengths == [S]
    del feats

    def run(fp8, forced=None):
        e.enable_fp8_decode(fp8); e.enable_fp8_kv(fp8); e.enable_fp8_prefill(fp8)
        logits, _ = e.prefill(embeds, lengths)
        rows, toks = [logits[0].float().cpu()], [int(torch.argmax(logits[0]))]
        for k in range(steps):
            feed = forced[k] if forced is not None else toks[-1]
            nxt, lg = e.decode_step(torch.tensor([feed]), want_logits=True)
            rows.append(lg[0].float().cpu()); toks.append(int(nxt[0]))
            assert int(nxt[0]) == int(torch.argmax(lg[0]))
        sync()
        assert e.kv_lengths(1) == [S + steps]
        return rows, toks

    ref_rows, ref_toks = run(False)
    q_rows, q_toks = run(True, forced=ref_toks)
    q_rows2, q_toks2 = run(True, forced=ref_toks)
    for a_, b_ in zip(q_rows, q_rows2):
        assert torch.equal(a_, b_)                          # deterministic
    errs = []
    for k, (r, q) in enumerate(zip(ref_rows, q_rows)):
        assert torch.isfinite(r).all() and torch.isfinite(q).all()
        errs.append(rel(q, r))
        # e4m3 operands carry 3 mantissa bits, and these are random synthetic weights (no trained structure to absorb the noise): after 28
        # layers over 33 k positions the quantised logits sit 0.26 (relative Frobenius) from the 16-bit run's at the prefill position and
        # closer at the decode steps (measured, profiles/r04_j_pytest_gpu.txt) -- cosine > 0.96.  A broken scale, a transposed scale
        # vector or a stale replica decorrelates the logits (distance >= 1)
        cos = float(torch.dot(q.double(), r.double()) / (q.double().norm() * r.double().norm()))
        # bound: per-layer quantisation distance 5.5e-2 at 16 k keys, growing as sqrt(layers) (tests/test_gpu_fp8.py::
        # test_fp8_modes_error_per_layer_over_four_full_width_layers_16k_context: 0.055 / 0.078 / 0.095 / 0.110 after 1..4 layers) -> 0.29 after 28
        # layers; measured 0.26; stated bound 0.33 (round 4: 0.4)
        assert errs[-1] < 0.33 and cos > 0.94, (k, errs[-1], cos)
        noise = float((q - r).abs().max())
        top2 = torch.topk(r, 2).values
        if float(top2[0] - top2[1]) > 2.5 * noise:
            assert q_toks[k] == ref_toks[k], (k, q_toks[k], ref_toks[k])
    print("configs4 whole model: fp8 vs 16-bit logit distance per position", [round(x, 4) for x in errs])
    e.enable_fp8_decode(False); e.enable_fp8_kv(False); e.enable_fp8_prefill(False)
    e.close()
