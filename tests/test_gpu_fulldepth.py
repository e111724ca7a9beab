"""GPU: FULL-DEPTH oracle parity at the benchmarked geometry (BASELINE configs[1]: OmChat-13B, 3 tiles + 512 text ids ->
S = 3584, InternViT-6B 45 layers + Qwen2-7B 28 layers, the synthetic weights of bench.py).

The oracle (oracle/stream.py: the per-layer oracle functions, pinned to the reference's golden vectors, applied one layer
at a time in fp32) runs ONCE on the host: tower -> projector -> splice -> 28 decoder layers over the prompt followed by the
teacher-forced ids of the first decode steps.  The HIP path runs the same sample through the C ABI in bf16 and in f16 and is
compared at four seams: tower output (hidden_states[-1] without CLS), projected features, last-position prefill logits, and
the logits of three teacher-forced decode steps; the first greedy id must equal the oracle's wherever the oracle's
top-1 / top-2 margin exceeds the 16-bit noise measured at that seam.

Reference loops covered at the depth and width that is timed: modeling_intern_vit.py:244-288,317-355 (encoder / model),
omchat_arch.py:55-209 (splice), transformers modeling_qwen2.py:342-402,462-465 (decoder loop, lm_head).

Tolerances (relative Frobenius error against the fp32 oracle; the measured values are printed and recorded in DESIGN.md §2):
73 layers of 16-bit kernels, every op's output rounded to the 16-bit type as the reference's fp16 / bf16 modules do."""
import ctypes as C
import json
import math
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
from gpu_util import rel, sync, ptr
from omchat_amd import synth, _lib
from omchat_amd.config import omchat13b
from omchat_amd.engine import Engine

torch.set_grad_enabled(False)

N_TILES, N_TEXT, N_FORCED = 3, 512, 3
# stated tolerances per dtype: (tower, projected features, logits)
# measured on MI355X (profiles/r05_a_fulldepth_parity.json): bf16 tower 1.63e-2 / features 1.70e-2 / logits 3.3-3.6e-2,
# f16 2.03e-3 / 2.13e-3 / 4.1-4.5e-3 -- the bounds leave ~1.4-1.8 x for kernel changes that move summation order
TOL = {"bf16": (2.5e-2, 2.5e-2, 5e-2), "f16": (4e-3, 4e-3, 8e-3)}


def _sample(cfg):
    px = torch.from_numpy(synth.pixels(N_TILES, 448, 0))
    text = synth.token_ids(N_TEXT, 151643, 1).tolist()
    ids = torch.tensor([[-200, text[0], -200, text[1], -200] + text[2:]])
    return px, ids


def _gpu_run(dt, forced):
    """HIP path through the C ABI.  forced=None: free-running greedy for the first N_FORCED steps (returns its ids)."""
    cfg = omchat13b()
    S = N_TILES * 1024 + N_TEXT
    e = Engine(cfg, dtype=dt, max_seq=S + 40, max_batch=1, max_tiles=N_TILES, max_prefill_rows=S + 8)
    e.fill_synthetic(0)
    px, ids = _sample(cfg)
    tower = e.vit_forward(px).float().cpu()
    feats = e.encode_images(px)
    embeds, lengths, valid = e.splice(ids, None, feats)
    assert lengths == [S] and bool(valid.all())
    logits, _ = e.prefill(embeds, [S]); sync()
    out = dict(tower=tower, feats=feats.float().cpu(), logits=[logits[0].cpu()], ids=[int(e.argmax(logits)[0])])
    tok = out["ids"][0] if forced is None else forced[0]
    fed = []
    for k in range(N_FORCED):
        fed.append(tok)
        nxt, lg = e.decode_step(torch.tensor([tok]), want_logits=True); sync()
        out["logits"].append(lg[0].cpu())
        out["ids"].append(int(nxt[0]))
        tok = int(nxt[0]) if forced is None or k + 1 >= len(forced) else forced[k + 1]
    out["fed"] = fed
    e.close()
    del e
    torch.cuda.empty_cache()
    return out


def _device_weight_source(cfg, seed=0):
    """get(key) -> fp32 CPU tensor with the values omchat_fill_synthetic gives the engine: the counter-based generator of
    omchat_amd/synth.py evaluated on the device (bit-identical to the numpy generator: test_gpu_ops.py), then copied to the host."""
    lib = _lib.lib()
    specs = {k: (shape, std, off) for k, shape, std, off in synth.tensor_specs(cfg)}

    def dev_tensor(key):
        shape, std, off = specs[key]
        n = int(np.prod(shape))
        t = torch.empty(n, dtype=torch.bfloat16, device="cuda")
        scale = float(np.float32(std)) * math.sqrt(3.0)
        _lib.check(lib.omchat_op_fill_uniform(_lib.BF16, ptr(t), n, (synth.fnv1a64(key) ^ seed) & 0xFFFFFFFFFFFFFFFF, scale, off, None))
        sync()
        return t.view(*shape)

    def get(key):
        return dev_tensor(key).float().cpu()
    table = {}

    def embed_rows(ids):
        if "t" not in table:
            table["t"] = dev_tensor("model.embed_tokens.weight")
        return table["t"][ids.to("cuda")].float().cpu()
    return get, embed_rows


@pytest.fixture(scope="module")
def runs(gpu_lib):
    from oracle import stream
    # the streamed fp32 oracle is ~190 s on the 128 host threads torch uses on the MI355X boxes; a host with a fraction of that would push the
    # whole GPU suite past its time budget -- skipping loudly is the lesser evil there
    if (os.cpu_count() or 1) < 48:
        pytest.skip(f"full-depth oracle pass needs a many-core host ({os.cpu_count()} CPUs here): run tests/test_gpu_fulldepth.py on its own")
    cfg = omchat13b()
    t0 = time.time()
    g16 = _gpu_run("bf16", None)
    forced = g16["fed"]
    h16 = _gpu_run("f16", forced)
    t1 = time.time()
    get, embed_rows = _device_weight_source(cfg)
    # spot check of the weight source against the host generator (the full check is test_fill_uniform_bit_exact_with_host_generator)
    k = "model.layers.27.mlp.down_proj.weight"
    assert torch.equal(get(k)[:2], torch.from_numpy(synth.uniform(k, (3584, 18944), 0, 0.02, 0.0)[:2]))
    px, ids = _sample(cfg)
    marks = []
    o = stream.run_streamed(px, ids, forced, get, embed_rows, cfg.vision, cfg.text,
                            progress=lambda s, i: marks.append((s, i, time.time())))
    t2 = time.time()
    print(f"\nfull-depth parity: HIP runs {t1 - t0:.1f} s, streamed fp32 oracle {t2 - t1:.1f} s on {torch.get_num_threads()} threads")
    return dict(oracle=o, bf16=g16, f16=h16, forced=forced, oracle_s=t2 - t1)


def _report(name, value):
    os.makedirs("gpurun_out", exist_ok=True)
    path = "gpurun_out/fulldepth_parity.json"
    d = json.load(open(path)) if os.path.exists(path) else {}
    d[name] = value
    json.dump(d, open(path, "w"), indent=1, sort_keys=True)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_full_depth_tower_and_projector_vs_streamed_oracle(runs, dt):
    o, g = runs["oracle"], runs[dt]
    assert o["S"] == N_TILES * 1024 + N_TEXT
    e_tower, e_feats = rel(g["tower"], o["tower"]), rel(g["feats"], o["feats"])
    print(f"\n{dt}: 45-layer tower rel err {e_tower:.3e}, projected features {e_feats:.3e}")
    _report(f"{dt}_tower", e_tower); _report(f"{dt}_feats", e_feats)
    assert torch.isfinite(g["tower"]).all() and torch.isfinite(g["feats"]).all()
    assert e_tower < TOL[dt][0], e_tower
    assert e_feats < TOL[dt][1], e_feats
    # per-tile: no tile carries the error of the others (batch independence seen from the oracle's side)
    for t in range(N_TILES):
        assert rel(g["tower"][t], o["tower"][t]) < TOL[dt][0] * 1.5


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_full_depth_prefill_and_teacher_forced_decode_logits_vs_streamed_oracle(runs, dt):
    o, g = runs["oracle"], runs[dt]
    errs = [rel(g["logits"][k], o["logits"][k]) for k in range(1 + N_FORCED)]
    print(f"\n{dt}: prefill logits rel err {errs[0]:.3e}; teacher-forced decode steps {['%.3e' % e for e in errs[1:]]}")
    _report(f"{dt}_logits", errs)
    for k, e in enumerate(errs):
        assert torch.isfinite(g["logits"][k]).all()
        assert e < TOL[dt][2], (k, e)
    # greedy ids: position k's argmax must equal the oracle's wherever the oracle's top-1 / top-2 margin clears the noise of this
    # seam: per-entry error ~ N(0, (rel err x rms(logits))^2), so a swap needs the DIFFERENCE of two entries (sigma x sqrt 2) to
    # exceed the margin -- guarded at margin > 6 x rel err x rms = 4.2 sigma of that difference
    agree, guarded = 0, 0
    for k in range(1 + N_FORCED):
        ol = o["logits"][k].double()
        top2 = torch.topk(ol, 2).values
        margin = float(top2[0] - top2[1])
        sigma = errs[k] * float(ol.pow(2).mean().sqrt())
        same = int(torch.argmax(ol)) == g["ids"][k]
        agree += same
        if margin > 6.0 * sigma:
            guarded += 1
            assert same, (k, margin, sigma, int(torch.argmax(ol)), g["ids"][k])
    print(f"{dt}: greedy ids equal to the oracle's at {agree} / {1 + N_FORCED} positions ({guarded} above the margin guard)")
    _report(f"{dt}_ids", dict(agree=agree, guarded=guarded, total=1 + N_FORCED))


def test_full_depth_first_greedy_id_bf16_run_is_what_was_forced(runs):
    # the bf16 run was free-running: its step-k id was fed at step k + 1, so the f16 run and the oracle saw the same ids
    g = runs["bf16"]
    assert g["fed"] == runs["forced"] == g["ids"][:N_FORCED]
    assert runs["f16"]["fed"] == runs["forced"]


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_full_depth_internvit300m_tower_and_projector_vs_streamed_oracle(gpu_lib, dt):
    """BASELINE configs[3] (OmChat-2.1-8B): the InternViT-300M tower at FULL depth (24 layers, LayerNorm, 16 heads x 64, no q/k norm) on the
    8 tiles of the benchmarked sample + the projector, against the layer-streamed fp32 oracle (intern_vit_300m/modeling_intern_vit.py:205-222,
    internVIT300m_encoder.py:45-56, multimodal_projector/builder.py:54-61)."""
    from oracle import stream
    from omchat_amd.config import omchat8b_21
    cfg = omchat8b_21()
    n = 8
    e = Engine(cfg, dtype=dt, max_seq=64, max_batch=1, max_tiles=n, text=False)
    e.fill_synthetic(0)
    px = torch.from_numpy(synth.pixels(n, 448, 3))
    tower = e.vit_forward(px).float().cpu()
    feats = e.encode_images(px).float().cpu(); sync()
    e.close()
    get, _ = _device_weight_source(cfg)
    o_tower, o_feats = stream.encode_images_streamed(px, get, cfg.vision, cfg.mm["mm_vision_select_layer"])
    e_t, e_f = rel(tower, o_tower), rel(feats, o_feats)
    print(f"\n{dt}: InternViT-300M 24-layer tower rel err {e_t:.3e}, projected features {e_f:.3e}")
    _report(f"{dt}_tower300m", e_t); _report(f"{dt}_feats300m", e_f)
    assert torch.isfinite(tower).all() and torch.isfinite(feats).all()
    assert e_t < TOL[dt][0] and e_f < TOL[dt][1], (e_t, e_f)


# ---------------------------------------------------------------------------------------------------------------------
# FULL-DEPTH padded batch (round 5): two rows of different spliced length through the reference's batch path -- splice with the attention mask
# (omchat_arch.py:55-209), right-padded prefill, then the decode branch (:61-70) through generate()'s own entries (masked_decode_begin + masked_next)
# -- 45 tower layers + 28 decoder layers, the full vocabulary, against the layer-streamed fp32 oracle of the LITERAL padded batch
# (oracle/stream.py padded_batch_streamed; tests/test_stream_oracle.py pins it to the whole-dict masked restatement).  Literal matters: the decode
# branch is fed the TEXT-level mask, so the shorter row's steps mask cache slots [t_r, T) and see its padded slots -- not the row computed alone.
# ---------------------------------------------------------------------------------------------------------------------
RAGGED_TEXT = (40, 87)          # text ids per row; one <image> sentinel each: spliced lengths 1064 and 1111


def _ragged_sample():
    text = synth.token_ids(sum(RAGGED_TEXT), 151643, 7).tolist()
    a, b = text[:RAGGED_TEXT[0]], text[RAGGED_TEXT[0]:]
    rows = [[-200] + a, b[:7] + [-200] + b[7:]]
    T = max(len(r) for r in rows)
    ids = torch.zeros(2, T, dtype=torch.long); mask = torch.zeros(2, T, dtype=torch.long)
    for i, r in enumerate(rows):
        ids[i, :len(r)] = torch.tensor(r); mask[i, :len(r)] = 1
    return ids, mask


def _gpu_ragged_run(dt, forced):
    """forced = None: free-running greedy (the ids every other run is then forced to); else [b][k] ids to feed"""
    from omchat_amd.model.omchat_qwen2 import OmChatQwen2ForCausalLM
    cfg = omchat13b()
    e = Engine(cfg, dtype=dt, max_seq=1160, max_batch=2, max_tiles=2)
    e.fill_synthetic(0)
    px, _ = _sample(cfg)
    px2 = px[:2]
    m = OmChatQwen2ForCausalLM(cfg.clone(), e)
    m.get_vision_tower = lambda: object()
    feats = e.encode_images(px2)
    m.encode_images = lambda images: feats
    ids, mask = _ragged_sample()
    dummy = torch.zeros(2, 3, 448, 448)
    out = m(input_ids=ids, attention_mask=mask, images=dummy, use_cache=True)
    kv = out.past_key_values
    logits = [out.logits[:, 0].float().cpu()]
    tok = torch.argmax(logits[0], dim=-1) if forced is None else torch.tensor([f[0] for f in forced])
    tok_mask = torch.cat([mask, torch.ones(2, 1, dtype=torch.long)], dim=1)
    _, pos1, mask1, _, _, _ = m.prepare_inputs_labels_for_multimodal(tok[:, None], None, tok_mask, kv, None, dummy)
    e.masked_decode_begin(pos1, mask1)
    fed = [[], []]
    for k in range(N_FORCED):
        for i in range(2):
            fed[i].append(int(tok[i]))
        nxt, lg = e.decode_step_masked_next(tok, want_logits=True); sync()
        logits.append(lg.float().cpu())
        tok = nxt.cpu().long() if forced is None or k + 1 >= N_FORCED else torch.tensor([f[k + 1] for f in forced])
    out = dict(logits=logits, fed=fed, feats=feats.float().cpu(), kv_len=kv.get_seq_length())
    e.close()
    del e, m
    torch.cuda.empty_cache()
    return out


@pytest.fixture(scope="module")
def ragged_runs(runs):
    from oracle import stream
    cfg = omchat13b()
    g16 = _gpu_ragged_run("bf16", None)
    h16 = _gpu_ragged_run("f16", g16["fed"])
    get, embed_rows = _device_weight_source(cfg)
    ids, mask = _ragged_sample()
    t0 = time.time()
    # the oracle's OWN projected features of tiles 0 and 1 (the tower treats tiles independently: same pixels as the batch-1 sample's first two tiles)
    lengths, ol = stream.padded_batch_streamed(ids, mask, runs["oracle"]["feats"][:2], g16["fed"], get, embed_rows, cfg.text, "right")
    print(f"\nfull-depth ragged batch: streamed fp32 oracle of the padded batch, rows of {lengths} positions: {time.time() - t0:.1f} s")
    return dict(oracle=ol, lengths=lengths, bf16=g16, f16=h16)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_full_depth_ragged_batch_prefill_and_masked_decode_vs_padded_oracle(ragged_runs, dt):
    o, g = ragged_runs["oracle"], ragged_runs[dt]
    assert ragged_runs["lengths"] == [1024 + RAGGED_TEXT[0], 1024 + RAGGED_TEXT[1]] and g["kv_len"] == max(ragged_runs["lengths"]) + N_FORCED      # the common cache slot: Lmax + k for every row
    assert g["fed"] == ragged_runs["bf16"]["fed"]
    errs = [[rel(g["logits"][k][i], o[i, k]) for k in range(1 + N_FORCED)] for i in range(2)]
    print(f"\n{dt}: ragged batch, rel err of the logits per row (prefill, then {N_FORCED} masked decode steps): " + "; ".join(str(["%.3e" % e for e in r]) for r in errs))
    _report(f"{dt}_ragged_logits", errs)
    for i in range(2):
        for k in range(1 + N_FORCED):
            assert torch.isfinite(g["logits"][k][i]).all()
            assert errs[i][k] < TOL[dt][2], (i, k, errs[i][k])
