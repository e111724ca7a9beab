#!/usr/bin/env python3
"""Generate tests/golden/fp8_per_layer_16k.npz: the ORACLE side of tests/test_gpu_fp8.py::test_fp8_modes_error_per_layer_over_four_full_width_layers_16k_context
(BASELINE configs[4]: fp8 weights, fp8 KV, 16 k tokens of context) -- four Qwen2-7B-width decoder layers over S = 16400 positions run on the
DE-QUANTISED operands (per-row e4m3 weights of q / k / v / gate / up, per-token e4m3 activations behind both RMSNorms), then two decode steps
(tokens 5 and 9) on the de-quantised weights and the de-quantised K / V cache.  Through round 5 that pass ran live inside the GPU test: 155 s on a
128-thread host, and a skip on hosts with < 48 CPUs; now the test compares against this fixture on any host.

    python tools/make_fp8_fixture.py            # ~6 minutes on 8 cores; peak ~12 GB (the prefill is chunked over query rows)

The oracle functions are oracle/decoder.py's (transformers modeling_qwen2.py:150-172,195-234,46-48,269-298); the prefill is fed to them in chunks of
2048 rows through the KV cache -- row-wise the same arithmetic as one pass (each query row sees the same keys under the same causal mask).
Kept: every 769th element + the squared norm of the post-final-norm hidden state after 1..4 layers, and the two decode steps' logits."""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import KVCache, decode_step                               # noqa: E402
from oracle.decoder import rope_cos_sin, qwen2_attention, qwen2_mlp     # noqa: E402
from oracle.vit import rms_norm                                        # noqa: E402
from omchat_amd import synth                                           # noqa: E402
from omchat_amd.config import omchat13b                                # noqa: E402

torch.set_grad_enabled(False)
L, S, CH, STRIDE = 4, 16400, 2048, 769
TOKENS = (5, 9)
OUT = os.path.join(ROOT, "tests", "golden", "fp8_per_layer_16k.npz")


def quant_ref(w):
    """per-row absmax / 448 scale, e4m3 round-to-nearest-even of w / scale (tests/test_gpu_fp8.py quant_ref; bit-exact against the device quantiser)"""
    w = w.float()
    m = w.abs().amax(dim=1)
    s = torch.where(m > 0, m / 448.0, torch.ones_like(m))
    return (w / s[:, None]).to(torch.float8_e4m3fn), s


def dequant_ref(w):
    q, s = quant_ref(w)
    return q.float() * s[:, None]


def fast_uniform(name, shape, seed, std, off, chunk=1 << 20):
    n = int(np.prod(shape))
    out = np.empty(n, np.float32)

    def job(s0):
        c = min(chunk, n - s0)
        out[s0:s0 + c] = synth.uniform_range(name, s0, c, seed, std, off)
    with ThreadPoolExecutor(os.cpu_count() or 1) as ex:
        list(ex.map(job, range(0, n, chunk)))
    return torch.from_numpy(out.reshape(shape))


def main():
    t0 = time.time()
    cfg = omchat13b()
    cfg.text["num_hidden_layers"] = L
    cfg.text["vocab_size"] = 2048
    sd = {k: fast_uniform(k, shape, 0, std, off) for k, shape, std, off in synth.tensor_specs(cfg)
          if not k.startswith(synth.TOWER) and "mm_projector" not in k}
    x = (torch.randn(1, S, 3584, generator=torch.Generator().manual_seed(1)) * 0.5).bfloat16().float()
    Q = lambda t: dequant_ref(t.to(torch.bfloat16).float().reshape(-1, t.shape[-1])).reshape(t.shape)      # per-token e4m3 of the 16-bit rows
    rb = lambda v: v.to(torch.bfloat16).float()
    sdq_pre = dict(sd)
    for k, v in sd.items():
        if k.endswith("weight") and any(s in k for s in ("q_proj", "k_proj", "v_proj", "gate_proj", "up_proj")):
            sdq_pre[k] = dequant_ref(rb(v))
    cos, sin = rope_cos_sin(torch.arange(S)[None], 128, cfg.text["rope_theta"], torch.float32)
    cache = KVCache(L)
    h = x
    out = {}
    for i in range(L):
        P = f"model.layers.{i}."
        parts = []
        for c0 in range(0, S, CH):
            c1 = min(S, c0 + CH)
            xc = Q(rms_norm(h[:, c0:c1], sd[P + "input_layernorm.weight"], 1e-6))
            parts.append(qwen2_attention(xc, sdq_pre, P, cfg.text, cos[:, c0:c1], sin[:, c0:c1], cache, i))
        h = h + torch.cat(parts, dim=1)
        h = h + qwen2_mlp(Q(rms_norm(h, sd[P + "post_attention_layernorm.weight"], 1e-6)), sdq_pre, P)
        ref = rms_norm(h, sd["model.norm.weight"], 1e-6)[0].reshape(-1).double()
        out[f"hid{i + 1}_sample"] = ref[::STRIDE].float().numpy()
        out[f"hid{i + 1}_norm2"] = np.array(float(ref.pow(2).sum()))
        print(f"[{time.time() - t0:6.0f} s] layer {i + 1} / {L}", flush=True)
    # decode on the de-quantised weights (all projections + lm_head) and the de-quantised cache rows
    dq = lambda t: dequant_ref(rb(t).reshape(-1, 128)).reshape(t.shape)
    cq = KVCache(L)
    for i in range(L):
        cq.update(dq(cache.k[i]), dq(cache.v[i]), i)
    sdq = dict(sd)
    for k, v in sd.items():
        if (".self_attn." in k or ".mlp." in k or k == "lm_head.weight") and k.endswith("weight") and "layernorm" not in k:
            sdq[k] = dequant_ref(rb(v))
    logits = [decode_step(torch.tensor([[tok]]), sdq, cfg.text, cq)[0, 0].numpy() for tok in TOKENS]
    out["decode_logits"] = np.stack(logits)
    out["meta"] = np.array(json.dumps(dict(L=L, S=S, stride=STRIDE, tokens=list(TOKENS), dtype="bf16", seconds=time.time() - t0, threads=torch.get_num_threads(),
                                           tool="tools/make_fp8_fixture.py")))
    np.savez(OUT, **out)
    print(f"wrote {OUT} ({os.path.getsize(OUT) / 2**20:.2f} MiB) in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
