#!/usr/bin/env python3
"""Generate tests/golden/fulldepth_configs1.npz: the fp32 oracle at FULL depth and width on the benchmarked samples, so that the
GPU parity tests (tests/test_gpu_fulldepth.py) no longer depend on a many-core host running the oracle live.

    python tools/make_fulldepth_fixture.py [--steps 32] [--out tests/golden/fulldepth_configs1.npz]

Runs on the CPU of the build container (8 cores, ~64 GB: about 25 minutes) or of any host; needs nothing but this repository -- the
oracle (`oracle/`, pinned to the reference's golden vectors by tests/test_oracle_golden.py and tests/test_stream_oracle.py) and the
counter-based synthetic weights of omchat_amd/synth.py (bit-identical to the device fill the engine uses).

What it computes (reference loops: modeling_intern_vit.py:244-288,317-355; omchat_arch.py:55-209 and :61-70; transformers
modeling_qwen2.py:342-402,462-465):

  1. configs[1]: 3 tiles -> 45-layer InternViT-6B tower -> projector -> splice (S = 3584) -> 28-layer Qwen2-7B prefill with a KV
     cache -> `--steps` greedy decode steps on the ORACLE's own ids (argmax of its fp32 logits, first index wins ties).
  2. the ragged batch (rows of 1064 / 1111 spliced positions, right-padded): padded prefill under the spliced mask, then three
     decode steps through the decode branch (text-level mask extended with ones, position_ids = sum(mask) - 1), oracle-chosen ids.
  3. configs[3]'s tower: 8 tiles through the 24-layer InternViT-300M (LayerNorm, 16 heads x 64) + projector.

What it keeps (tests/fulldepth_sample.py has the digest / compare helpers): per position every 16th logit + the squared norm + the
top 8 (ids, values) and the full logits of the prefill position; of the tower outputs and projected features every 193rd element per
tile + per-tile squared norms.  The decoder weights stay resident as bf16 (exact: the synthetic values are bf16) and are up-cast per
use; the tower's weights are generated per layer and dropped."""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle                                        # noqa: E402
from oracle import stream                            # noqa: E402
from omchat_amd import synth                         # noqa: E402
from omchat_amd.config import omchat13b, omchat8b_21  # noqa: E402
import fulldepth_sample as fs                        # noqa: E402

torch.set_grad_enabled(False)


def fast_uniform(name, shape, seed, std, off, chunk=1 << 20, workers=None):
    """synth.uniform, chunked over a thread pool (numpy releases the GIL in its ufuncs): ~40 M elements / s on 8 cores"""
    n = int(np.prod(shape))
    out = np.empty(n, np.float32)

    def job(s):
        c = min(chunk, n - s)
        out[s:s + c] = synth.uniform_range(name, s, c, seed, std, off)
    with ThreadPoolExecutor(workers or os.cpu_count() or 1) as ex:
        list(ex.map(job, range(0, n, chunk)))
    return torch.from_numpy(out.reshape(shape))


class Weights:
    """dict-like view the whole-dict oracle functions index: fp32 on access.  keep=True tensors stay resident as bf16."""

    def __init__(self, cfg, seed=0):
        self.specs = {k: (shape, std, off) for k, shape, std, off in synth.tensor_specs(cfg)}
        self.seed = seed
        self.store = {}
        self.H = cfg.text["hidden_size"]

    def gen(self, key):
        shape, std, off = self.specs[key]
        return fast_uniform(key, shape, self.seed, std, off)

    def preload(self, keys):
        for k in keys:
            self.store[k] = self.gen(k).to(torch.bfloat16)

    def __contains__(self, key):
        return key in self.specs

    def __getitem__(self, key):
        if key == "model.embed_tokens.weight":
            return _Rows(self)
        if key in self.store:
            return self.store[key].float()
        return self.gen(key)

    def get(self, key, default=None):
        return self[key] if key in self.specs else default


class _Rows:
    """embed_tokens[ids] without the 2.2 GB table: the generator is counter-based, a row costs its own 3584 elements"""
    dtype = torch.float32

    def __init__(self, w):
        self.w = w
        self.shape = w.specs["model.embed_tokens.weight"][0]

    def __getitem__(self, ids):
        ids = torch.as_tensor(ids, dtype=torch.int64)
        H = self.shape[1]
        _, std, off = self.w.specs["model.embed_tokens.weight"]
        flat = [torch.from_numpy(synth.uniform_range("model.embed_tokens.weight", int(i) * H, H, self.w.seed, std, off)) for i in ids.reshape(-1)]
        return torch.stack(flat).reshape(*ids.shape, H) if flat else torch.zeros(*ids.shape, H)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", fs.FIXTURE))
    ap.add_argument("--skip-300m", action="store_true")
    args = ap.parse_args()
    K = args.steps
    cfg = omchat13b()
    tcfg = cfg.text
    timing = dict(threads=torch.get_num_threads(), cpus=os.cpu_count(), torch=torch.__version__)
    out = {}
    T0 = time.time()

    def log(msg):
        print(f"[{time.time() - T0:7.1f} s] {msg}", flush=True)

    # ---- 1. configs[1]: tower + projector (weights generated per layer, dropped)
    W = Weights(cfg)
    px, ids = fs.sample()
    t = time.time()
    gen_s = [0.0]

    def get(key):
        t1 = time.time()
        v = W[key]
        gen_s[0] += time.time() - t1
        return v
    tower, feats = stream.encode_images_streamed(px, get, cfg.vision, cfg.mm["mm_vision_select_layer"],
                                                 progress=lambda s, i: log(f"tower layer {i}") if i % 5 == 4 else None)
    timing["tower_s"] = time.time() - t - gen_s[0]
    timing["tower_weight_gen_s"] = gen_s[0]
    log(f"tower + projector done: {timing['tower_s']:.1f} s of oracle arithmetic, {gen_s[0]:.1f} s of weight generation")
    out["tower_sample"], out["tower_norm2"] = fs.act_digest(tower)
    out["feats_sample"], out["feats_norm2"] = fs.act_digest(feats)

    # ---- decoder weights resident (bf16), final norm + lm_head too
    t = time.time()
    dec_keys = [k for k in W.specs if k.startswith("model.layers.") or k in ("model.norm.weight", "lm_head.weight")]
    W.preload(dec_keys)
    timing["decoder_weight_gen_s"] = time.time() - t
    log(f"decoder weights resident ({sum(v.numel() for v in W.store.values()) * 2 / 2**30:.1f} GiB as bf16)")

    # ---- prefill with a cache, then K greedy steps on the oracle's own ids
    rows = W["model.embed_tokens.weight"]
    embeds, _, lengths = oracle.splice_inputs(ids, None, [f for f in feats], rows)
    S = lengths[0]
    assert S == fs.N_TILES * 1024 + fs.N_TEXT
    cache = oracle.KVCache(tcfg["num_hidden_layers"])
    t = time.time()
    h = oracle.qwen2_model(embeds, W, tcfg, cache, None, None)
    logits = [oracle.lm_head(h[:, -1:], W)[0, 0]]
    timing["prefill_s"] = time.time() - t
    log(f"prefill of {S} positions: {timing['prefill_s']:.1f} s")
    del h
    forced, step_s = [], []
    for k in range(K):
        tok = int(torch.argmax(logits[-1]))
        forced.append(tok)
        t = time.time()
        logits.append(oracle.decode_step(torch.tensor([[tok]]), W, tcfg, cache)[0, -1])
        step_s.append(time.time() - t)
        if k % 8 == 7:
            log(f"decode step {k + 1} / {K}: {step_s[-1]:.2f} s")
    timing["decode_s_per_token"] = float(np.median(step_s)) if step_s else None
    dig = [fs.logit_digest(l) for l in logits]
    out["forced"] = np.array(forced, np.int64)
    out["logit_samples"] = np.stack([d["sample"] for d in dig])
    out["logit_norm2"] = np.array([d["norm2"] for d in dig])
    out["top_ids"] = np.stack([d["top_ids"] for d in dig])
    out["top_vals"] = np.stack([d["top_vals"] for d in dig])
    out["logits0_full"] = logits[0].numpy()
    del cache

    # ---- 2. ragged batch: the literal padded batch, oracle-chosen ids (the whole-dict restatement of tests/test_stream_oracle.py)
    rids, rmask = fs.ragged_sample()
    t = time.time()
    emb, mask_sp, rlen = oracle.splice_inputs(rids, rmask, [f for f in feats[:2]], rows, "right", None)
    cache = oracle.KVCache(tcfg["num_hidden_layers"])
    h = oracle.qwen2_model(emb, W, tcfg, cache, None, mask_sp)
    last = [n - 1 for n in rlen]
    rl = [torch.stack([oracle.lm_head(h[i:i + 1, last[i]:last[i] + 1], W)[0, 0] for i in range(2)])]
    tok_mask = torch.cat([rmask, torch.ones(2, 1, dtype=torch.long)], dim=1)
    rforced = []
    for k in range(3):
        tok = torch.argmax(rl[-1], dim=-1)
        rforced.append(tok)
        mo, po = oracle.decode_step_inputs(tok_mask, cache.get_seq_length())
        ho = oracle.qwen2_model(rows[tok][:, None], W, tcfg, cache, po, mo)
        rl.append(oracle.lm_head(ho, W)[:, -1])
        tok_mask = torch.cat([tok_mask, torch.ones(2, 1, dtype=torch.long)], dim=1)
    timing["ragged_s"] = time.time() - t
    log(f"ragged batch (rows of {rlen}): {timing['ragged_s']:.1f} s")
    rdig = [[fs.logit_digest(rl[k][i]) for k in range(4)] for i in range(2)]
    out["rag_forced"] = torch.stack(rforced, dim=1).numpy().astype(np.int64)
    out["rag_lengths"] = np.array(rlen, np.int64)
    out["rag_logit_samples"] = np.stack([np.stack([d["sample"] for d in r]) for r in rdig])
    out["rag_logit_norm2"] = np.array([[d["norm2"] for d in r] for r in rdig])
    out["rag_top_ids"] = np.stack([np.stack([d["top_ids"] for d in r]) for r in rdig])
    out["rag_top_vals"] = np.stack([np.stack([d["top_vals"] for d in r]) for r in rdig])
    del cache, W

    # ---- 3. configs[3]: InternViT-300M tower + projector on 8 tiles
    if not args.skip_300m:
        cfg3 = omchat8b_21()
        W3 = Weights(cfg3)
        t = time.time()
        t3, f3 = stream.encode_images_streamed(fs.pixels_300m(), lambda k: W3[k], cfg3.vision, cfg3.mm["mm_vision_select_layer"])
        timing["tower300m_s"] = time.time() - t
        log(f"InternViT-300M tower + projector, 8 tiles: {timing['tower300m_s']:.1f} s")
        out["t300_tower_sample"], out["t300_tower_norm2"] = fs.act_digest(t3)
        out["t300_feats_sample"], out["t300_feats_norm2"] = fs.act_digest(f3)

    timing["total_s"] = time.time() - T0
    out["meta"] = np.array(json.dumps(dict(timing=timing, steps=K, logit_stride=fs.LOGIT_STRIDE, act_stride=fs.ACT_STRIDE,
                                           generated=time.strftime("%Y-%m-%d"), tool="tools/make_fulldepth_fixture.py")))
    np.savez(args.out, **out)
    log(f"wrote {args.out} ({os.path.getsize(args.out) / 2**20:.2f} MiB)")
    print(json.dumps(timing, indent=1))


if __name__ == "__main__":
    main()
