"""CPU: the layer-streamed oracle driver (oracle/stream.py, used by the full-depth GPU parity test) against the whole-dict
oracle (oracle/pipeline.py, itself pinned to the reference's golden vectors by test_oracle_golden.py) on the tiny config."""
import numpy as np
import torch
from conftest import golden, rel_err
import oracle
from oracle import stream
from omchat_amd import synth
from omchat_amd.config import tiny

torch.set_grad_enabled(False)
T = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt)


def test_streamed_run_equals_whole_dict_oracle_and_cached_decode_steps():
    g = golden("e2e_tiny")
    cfg = tiny()
    seed = int(g["seed"])
    specs = {k: (shape, std, off) for k, shape, std, off in synth.tensor_specs(cfg)}
    fetched = []

    def get(key):
        shape, std, off = specs[key]
        fetched.append(key)
        return T(synth.uniform(key, shape, seed, std, off))

    def embed_rows(ids):
        return get("model.embed_tokens.weight")[ids]

    px = T(synth.pixels(int(g["n_tiles"]), 56, int(g["pixel_seed"])))
    ids = T(g["ids"], torch.long)
    sd = {k: T(v) for k, v in synth.state_dict(cfg, seed).items()}
    # whole-dict oracle: prefill, then two cached decode steps on its own greedy ids
    logits, cache, lengths = oracle.prefill(ids, px, sd, cfg.vision, cfg.text)
    want = [logits[0, -1]]
    forced = []
    for _ in range(2):
        forced.append(int(torch.argmax(want[-1])))
        want.append(oracle.decode_step(torch.tensor([[forced[-1]]]), sd, cfg.text, cache)[0, -1])
    r = stream.run_streamed(px, ids, forced, get, embed_rows, cfg.vision, cfg.text)
    assert r["S"] == lengths[0] == int(g["prefill_len"])
    assert torch.equal(r["feats"], oracle.encode_images(px, sd, cfg.vision))
    # position S - 1 + k of the uncached pass == decode step k on the cache (different GEMM shapes: fp32 summation order only)
    for k in range(3):
        assert rel_err(r["logits"][k], want[k]) < 2e-5, k
    # and the golden vector captured from the reference itself
    assert rel_err(r["logits"][0], g["prefill_logits_last"]) < 5e-3
    # every layer tensor was fetched exactly once: one pass over the weights
    per_layer = [k for k in fetched if ".layers." in k]
    assert len(per_layer) == len(set(per_layer))


def test_padded_batch_streamed_equals_the_whole_dict_masked_restatement():
    """oracle/stream.py padded_batch_streamed (the full-depth padded-batch GPU test's oracle: one layer at a time, teacher-forced decode steps on a
    one-layer cache) against the whole-dict restatement of the reference's padded batch -- splice with the attention mask (omchat_arch.py:55-209),
    padded prefill, then the decode branch (:61-70) fed the TEXT-level mask generate() carries -- same logits; and the property that makes the literal
    form matter: a right-padded row's decode step masks cache slots [t_r, T), not its pads, so it does NOT equal the row computed alone"""
    cfg = tiny()
    sd = {k: T(v) for k, v in synth.state_dict(cfg, 5).items()}
    get = lambda key: sd[key]
    embed_rows = lambda ids: sd["model.embed_tokens.weight"][ids]
    feats = oracle.encode_images(T(synth.pixels(3, 56, 2)), sd, cfg.vision)
    rows = [[5, -200, 6, -200, 9, 10, 11], [-200] + list(range(30, 49)), [40, 41, 42]]
    Tn = max(len(r) for r in rows)
    # a row without a sentinel consumes a (zero-length slice of a) feature entry: four entries for three sentinels + one empty row
    feats4 = torch.cat([feats, feats[:1]], dim=0)
    for side in ("right", "left"):
        ids = torch.zeros(3, Tn, dtype=torch.long); mask = torch.zeros(3, Tn, dtype=torch.long)
        for i, r in enumerate(rows):
            sl = slice(0, len(r)) if side == "right" else slice(Tn - len(r), Tn)
            ids[i, sl] = torch.tensor(r); mask[i, sl] = 1
        emb, mask_sp, lengths = oracle.splice_inputs(ids, mask, [f for f in feats4], sd["model.embed_tokens.weight"], side, None)
        cache = oracle.KVCache(cfg.text["num_hidden_layers"])
        h = oracle.qwen2_model(emb, sd, cfg.text, cache, None, mask_sp)
        last = [n - 1 for n in lengths] if side == "right" else [emb.shape[1] - 1] * 3
        want = [torch.stack([oracle.lm_head(h[i:i + 1, last[i]:last[i] + 1], sd)[0, 0] for i in range(3)])]
        tok = torch.argmax(want[0], dim=-1)
        forced = []
        tok_mask = torch.cat([mask, torch.ones(3, 1, dtype=torch.long)], dim=1)
        for k in range(3):
            forced.append(tok)
            mo, po = oracle.decode_step_inputs(tok_mask, cache.get_seq_length())
            ho = oracle.qwen2_model(sd["model.embed_tokens.weight"][tok][:, None], sd, cfg.text, cache, po, mo)
            want.append(oracle.lm_head(ho, sd)[:, -1])
            tok = torch.argmax(want[-1], dim=-1)
            tok_mask = torch.cat([tok_mask, torch.ones(3, 1, dtype=torch.long)], dim=1)
        forced = torch.stack(forced, dim=1)
        lengths2, got = stream.padded_batch_streamed(ids, mask, feats4, forced, get, embed_rows, cfg.text, side)
        assert lengths2 == lengths and len(set(lengths)) == 3 and got.shape[:2] == (3, 4)
        for k in range(4):
            assert rel_err(got[:, k], want[k]) < 2e-5, (side, k)           # other GEMM shapes in the final norm / lm_head: fp32 summation order only
        if side == "right":
            # the shortest row alone (no padding): its prefill logits are the padded batch's, its decode steps are not
            r = 2
            x = torch.cat([emb[r:r + 1, :lengths[r]], embed_rows(forced[r])[None]], dim=1)
            alone = stream.decoder_streamed(x, get, cfg.text, last_n=4)[0]
            assert rel_err(alone[0], want[0][r]) < 2e-5
            assert rel_err(alone[1], want[1][r]) > 1e-3
