"""Oracle: the whole hot path at FULL depth and width, layer-streamed.  Test infrastructure only.

`oracle.pipeline.prefill` wants the whole state dict at once (OmChat-13B in fp32 = 52 GB).  This driver walks the same
functions one layer at a time: `get(key)` hands over ONE tensor (fp32, reference key names of SURVEY.md Appendix B), the
layer is applied and its weights are dropped, so the peak is one layer's weights plus the activations.  It calls the SAME
per-layer oracle functions as the whole-dict path (`tests/test_stream_oracle.py` checks that both give the same bits on
the tiny config), i.e. the loops restated here are only

  * InternVisionEncoder.forward     omchat/model/multimodal_encoder/intern_vit_6b/modeling_intern_vit.py:244-288
  * InternVisionModel.forward       :317-355 (embeddings -> encoder), feature_select internVIT_encoder.py:35-56
  * prepare_inputs_labels_for_multimodal  omchat/model/omchat_arch.py:55-209 (oracle.splice)
  * Qwen2Model.forward              transformers modeling_qwen2.py:342-402, lm_head :462-465

Teacher-forced decode steps: a causal decoder's logits at position S - 1 + k of a prefill over the prompt followed by k
forced tokens ARE the logits of decode step k after a prefill of the prompt (same keys, same mask, exact arithmetic), so
one streamed pass over `S + k` positions yields the prefill logits and the k decode-step logits without streaming the
weights k + 1 times.
"""
import torch

from .vit import vit_embeddings, vit_layer, projector_forward
from .decoder import qwen2_layer, rope_cos_sin, rms_norm, KVCache
from .splice import splice_inputs, decode_step_inputs

TOWER = "model.vision_tower.vision_tower."


def _layer_keys_vit(vcfg, j):
    P = f"encoder.layers.{j}."
    keys = [P + "ls1", P + "ls2", P + "norm1.weight", P + "norm2.weight", P + "attn.qkv.weight",
            P + "attn.proj.weight", P + "attn.proj.bias", P + "mlp.fc1.weight", P + "mlp.fc1.bias",
            P + "mlp.fc2.weight", P + "mlp.fc2.bias"]
    if vcfg.get("norm_type", "rms_norm") == "layer_norm":
        keys += [P + "norm1.bias", P + "norm2.bias"]
    if vcfg.get("qk_normalization", True):
        keys += [P + "attn.q_norm.weight", P + "attn.k_norm.weight"]
    return keys


def _layer_keys_dec(i):
    P = f"model.layers.{i}."
    return [P + n for n in ("self_attn.q_proj.weight", "self_attn.q_proj.bias", "self_attn.k_proj.weight",
                            "self_attn.k_proj.bias", "self_attn.v_proj.weight", "self_attn.v_proj.bias",
                            "self_attn.o_proj.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight",
                            "mlp.down_proj.weight", "input_layernorm.weight", "post_attention_layernorm.weight")]


def tower_streamed(pixels, get, vcfg, select_layer=-1, select_feature="patch", progress=None):
    """vision_tower_forward (oracle/vit.py) with the layer weights fetched one layer at a time.
    pixels fp32 [n, 3, S, S]; returns fp32 [n, tokens, C]."""
    L = vcfg["num_hidden_layers"]
    idx = select_layer if select_layer >= 0 else L + 1 + select_layer
    w = {k: get(TOWER + k) for k in ("embeddings.class_embedding", "embeddings.position_embedding",
                                    "embeddings.patch_embedding.weight", "embeddings.patch_embedding.bias")}
    h = vit_embeddings(pixels, w, vcfg["patch_size"], vcfg["image_size"])
    for j in range(idx):                                                     # hidden_states[idx] = output of layer idx - 1
        w = {k: get(TOWER + k) for k in _layer_keys_vit(vcfg, j)}
        h = vit_layer(h, w, j, vcfg["num_attention_heads"], vcfg.get("layer_norm_eps", 1e-6))
        del w
        if progress:
            progress("vit", j)
    if select_feature == "patch":
        h = h[:, 1:]
    elif select_feature != "cls_patch":
        raise ValueError(f"Unexpected select feature: {select_feature}")
    return h


def encode_images_streamed(pixels, get, vcfg, select_layer=-1, progress=None):
    """encode_images (omchat_arch.py:50-53): tower then mm_projector.  Returns (tower features, projected features)."""
    feats = tower_streamed(pixels, get, vcfg, select_layer, "patch", progress)
    pw = {k: get("model.mm_projector." + k) for k in ("0.weight", "0.bias", "2.weight", "2.bias")}
    return feats, projector_forward(feats, pw)


def decoder_streamed(embeds, get, tcfg, last_n=1, progress=None):
    """Qwen2Model.forward without a cache over embeds [b, S, H] (fp32), then the final norm and lm_head on the last
    `last_n` positions only.  Returns logits fp32 [b, last_n, vocab]."""
    b, S, _ = embeds.shape
    nh = tcfg["num_attention_heads"]
    d = tcfg.get("head_dim") or tcfg["hidden_size"] // nh
    pos = torch.arange(S)[None, :].expand(b, S)
    cos, sin = rope_cos_sin(pos, d, tcfg.get("rope_theta", 1e6), embeds.dtype)
    h = embeds
    for i in range(tcfg["num_hidden_layers"]):
        w = {k: get(k) for k in _layer_keys_dec(i)}
        h = qwen2_layer(h, w, i, tcfg, cos, sin, None, None)
        del w
        if progress:
            progress("dec", i)
    h = rms_norm(h[:, S - last_n:], get("model.norm.weight"), tcfg.get("rms_norm_eps", 1e-6))
    return torch.nn.functional.linear(h, get("lm_head.weight"))


def run_streamed(pixels, input_ids, forced_tokens, get, embed_rows, vcfg, tcfg, select_layer=-1, progress=None):
    """The path of BASELINE configs[1] for ONE sequence: tiles -> tower -> projector -> splice -> decoder, with
    `forced_tokens` (list of ids) appended after the prompt.  `embed_rows(ids)` returns the fp32 embedding rows of
    int64 `ids` (the table itself is 2.2 GB in fp32 and only ~500 rows are needed).

    Returns dict(tower, feats, embeds [1, S, H], logits [1 + len(forced), vocab]): logits[0] are the prefill's
    last-position logits, logits[k] those of the decode step that consumes forced_tokens[k - 1]."""
    tower, feats = encode_images_streamed(pixels, get, vcfg, select_layer, progress)

    class _Rows:                                                              # embed_tokens[ids] for splice_inputs
        dtype = torch.float32
        shape = (tcfg["vocab_size"], tcfg["hidden_size"])

        def __getitem__(self, ids):
            return embed_rows(ids)
    embeds, _, lengths = splice_inputs(input_ids, None, [f for f in feats], _Rows())
    S = lengths[0]
    k = len(forced_tokens)
    x = embeds
    if k:
        x = torch.cat([embeds, embed_rows(torch.tensor(forced_tokens, dtype=torch.int64))[None]], dim=1)
    logits = decoder_streamed(x, get, tcfg, last_n=k + 1, progress=progress)
    return dict(tower=tower, feats=feats, embeds=embeds, logits=logits[0], S=S)


def padded_batch_streamed(input_ids, attention_mask, feats, forced_tokens, get, embed_rows, tcfg, padding_side="right", progress=None):
    """A padded batch as the reference computes it, one decoder layer at a time: splice with the TEXT-level attention mask
    (omchat_arch.py:55-209), padded prefill under the spliced mask (positions arange, padded keys masked), then k teacher-forced
    decode steps through the decode branch (omchat_arch.py:61-70): the mask generate() carries is the TEXT-level one extended by a
    one per step, the branch extends it with ones to cache length + 1 and takes position_ids = sum(mask) - 1 -- so a decode step
    masks cache slots [t_r, T) of a right-padded row (t_r text ids of T), NOT its padded slots; this literal behaviour is what the
    HIP path mirrors.  Teacher forcing makes the steps' layer-i inputs known once layer i - 1 is done, so every layer's weights are
    fetched once and only one layer's cache is alive.  forced_tokens: int64 [b, k].
    Returns (lengths, logits fp32 [b, 1 + k, vocab]): [:, 0] at each row's last valid (right) / last (left) prefill position."""
    class _Rows:
        dtype = torch.float32
        shape = (tcfg["vocab_size"], tcfg["hidden_size"])

        def __getitem__(self, ids):
            return embed_rows(ids)
    embeds, mask_sp, lengths = splice_inputs(input_ids, attention_mask, [f for f in feats], _Rows(), padding_side, None)
    b, L0, _ = embeds.shape
    forced = torch.as_tensor(forced_tokens, dtype=torch.int64).reshape(b, -1)
    k = forced.shape[1]
    nh = tcfg["num_attention_heads"]
    d = tcfg.get("head_dim") or tcfg["hidden_size"] // nh
    theta = tcfg.get("rope_theta", 1e6)
    cos0, sin0 = rope_cos_sin(torch.arange(L0)[None, :].expand(b, L0), d, theta, embeds.dtype)
    steps = []
    tok_mask = torch.cat([attention_mask, torch.ones((b, 1), dtype=attention_mask.dtype)], dim=1)
    for j in range(k):
        mo, po = decode_step_inputs(tok_mask, L0 + j)
        steps.append((mo, rope_cos_sin(po, d, theta, embeds.dtype)))
        tok_mask = torch.cat([tok_mask, torch.ones((b, 1), dtype=attention_mask.dtype)], dim=1)
    h0 = embeds
    hd = [embed_rows(forced[:, j])[:, None] for j in range(k)]
    for i in range(tcfg["num_hidden_layers"]):
        w = {key: get(key) for key in _layer_keys_dec(i)}
        cache = KVCache(i + 1)
        h0 = qwen2_layer(h0, w, i, tcfg, cos0, sin0, cache, mask_sp)
        for j in range(k):
            mo, (cj, sj) = steps[j]
            hd[j] = qwen2_layer(hd[j], w, i, tcfg, cj, sj, cache, mo)
        del w, cache
        if progress:
            progress("dec", i)
    last = [n - 1 for n in lengths] if padding_side == "right" else [L0 - 1] * b
    rows = torch.cat([torch.stack([h0[r, last[r]] for r in range(b)])[:, None]] + hd, dim=1)          # [b, 1 + k, H]
    rows = rms_norm(rows, get("model.norm.weight"), tcfg.get("rms_norm_eps", 1e-6))
    return lengths, torch.nn.functional.linear(rows, get("lm_head.weight"))
