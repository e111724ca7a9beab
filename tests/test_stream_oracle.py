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


def test_ragged_rows_unpadded_equal_the_masked_padded_restatement():
    """oracle/stream.py ragged_rows_streamed (the full-depth padded-batch GPU test's oracle: every row alone, no padding) against the LITERAL
    restatement of the reference's padded batch: splice with the attention mask (omchat_arch.py:55-209), padded prefill with the additive mask,
    then the decode branch (:61-70: mask extended with ones, position_ids = sum(mask) - 1) on the shared cache -- the same logits per row up to
    fp32 summation order"""
    cfg = tiny()
    seed = 5
    sd = {k: T(v) for k, v in synth.state_dict(cfg, seed).items()}
    get = lambda key: sd[key]
    embed_rows = lambda ids: sd["model.embed_tokens.weight"][ids]
    px = T(synth.pixels(3, 56, 2))
    feats = oracle.encode_images(px, sd, cfg.vision)
    rows = [[5, -200, 6, -200, 9, 10, 11], [-200] + list(range(30, 49)), [40, 41, 42]]
    n_img = 3
    Tn = max(len(r) for r in rows)
    ids = torch.zeros(3, Tn, dtype=torch.long); mask = torch.zeros(3, Tn, dtype=torch.long)
    for i, r in enumerate(rows):
        ids[i, :len(r)] = torch.tensor(r); mask[i, :len(r)] = 1
    # a row without a sentinel consumes a (zero-length slice of a) feature entry: four entries for three sentinels + one empty row
    feats4 = torch.cat([feats[:n_img], feats[:1]], dim=0)
    # literal padded path
    emb, mask_sp, lengths = oracle.splice_inputs(ids, mask, [f for f in feats4], sd["model.embed_tokens.weight"], "right", None)
    cache = oracle.KVCache(cfg.text["num_hidden_layers"])
    h = oracle.qwen2_model(emb, sd, cfg.text, cache, None, mask_sp)
    want = [[oracle.lm_head(h[i:i + 1, n - 1:n], sd)[0, 0]] for i, n in enumerate(lengths)]
    tok = torch.stack([torch.argmax(w[0]) for w in want])
    forced = [[] for _ in rows]
    tok_mask = torch.cat([mask_sp.to(torch.long), torch.ones(3, 1, dtype=torch.long)], dim=1)
    for k in range(3):
        for i in range(3):
            forced[i].append(int(tok[i]))
        mo, po = oracle.decode_step_inputs(tok_mask, cache.get_seq_length())
        ho = oracle.qwen2_model(sd["model.embed_tokens.weight"][tok][:, None], sd, cfg.text, cache, po, mo)
        lg = oracle.lm_head(ho, sd)[:, -1]
        for i in range(3):
            want[i].append(lg[i])
        tok = torch.argmax(lg, dim=-1)
        tok_mask = torch.cat([tok_mask, torch.ones(3, 1, dtype=torch.long)], dim=1)
    lengths2, got = stream.ragged_rows_streamed(ids, mask, feats4, forced, get, embed_rows, cfg.text)
    assert lengths2 == lengths and len(set(lengths)) == 3
    for i in range(3):
        assert got[i].shape[0] == 4
        for k in range(4):
            assert rel_err(got[i][k], want[i][k]) < 2e-5, (i, k)
