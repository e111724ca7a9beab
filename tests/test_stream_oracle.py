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
