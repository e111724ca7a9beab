"""Oracle: Qwen2 decoder (torch CPU).  Test infrastructure only.

Restated from `transformers/models/qwen2/modeling_qwen2.py` (third-party; the reference inherits it at
omchat/model/language_model/omchat_qwen2.py:7,22,29 -- pinned transformers==4.41.2 in pyproject.toml:22,
5.15.0 installed here; the formulas below are unchanged between the two).  Weight keys follow the
reference checkpoint (`model.layers.{i}.self_attn.q_proj.weight`, ..., `model.norm.weight`, `lm_head.weight`).
"""
import torch
import torch.nn.functional as F
from .vit import rms_norm


class KVCache:
    """DynamicCache.update semantics (modeling_qwen2.py:211-214): per layer, post-RoPE K and raw V,
    [b, n_kv, L, d], appended along L."""

    def __init__(self, num_layers):
        self.k = [None] * num_layers
        self.v = [None] * num_layers

    def update(self, k, v, layer):
        if self.k[layer] is None:
            self.k[layer], self.v[layer] = k, v
        else:
            self.k[layer] = torch.cat([self.k[layer], k], dim=2)
            self.v[layer] = torch.cat([self.v[layer], v], dim=2)
        return self.k[layer], self.v[layer]

    def get_seq_length(self):
        return 0 if self.k[0] is None else self.k[0].shape[2]


def rope_cos_sin(position_ids, head_dim, theta, dtype):
    """Qwen2RotaryEmbedding (modeling_qwen2.py:64-102): inv_freq = theta^(-2i/d) and cos/sin in fp32, then cast
    to the activation dtype.  position_ids [b, S] -> cos, sin [b, S, d]."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    freqs = position_ids[:, :, None].float() * inv_freq[None, None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def _rotate_half(x):
    """rotate_half (modeling_qwen2.py:105-109)."""
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(q, k, cos, sin):
    """apply_rotary_pos_emb (modeling_qwen2.py:113-135), q/k [b, heads, S, d], cos/sin [b, S, d]."""
    cos = cos.unsqueeze(1)
    sin = sin.unsqueeze(1)
    return (q * cos) + (_rotate_half(q) * sin), (k * cos) + (_rotate_half(k) * sin)


def _additive_mask(b, S, L, dtype, attention_mask=None):
    """Causal mask as HF builds it for eager attention: query i (absolute position L-S+i) sees keys <= L-S+i;
    padded keys (attention_mask == 0) are masked; masked entries are finfo(dtype).min."""
    neg = torch.finfo(dtype).min
    qpos = torch.arange(L - S, L)[:, None]
    kpos = torch.arange(L)[None, :]
    allowed = (kpos <= qpos)[None, None].expand(b, 1, S, L)
    if attention_mask is not None:
        allowed = allowed & attention_mask.bool()[:, None, None, :L]
    m = torch.zeros(b, 1, S, L, dtype=dtype)
    return m.masked_fill(~allowed, neg)


def qwen2_attention(x, w, pfx, cfg, cos, sin, cache, layer, attention_mask=None):
    """Qwen2Attention.forward + eager_attention_forward (modeling_qwen2.py:150-172,195-234): q/k/v proj with
    bias, RoPE, cache append, repeat_kv, scores*scaling + mask, fp32 softmax cast back, @ v, o_proj (no bias)."""
    b, S, _ = x.shape
    nh, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    d = cfg.get("head_dim") or cfg["hidden_size"] // nh
    q = F.linear(x, w[pfx + "self_attn.q_proj.weight"], w[pfx + "self_attn.q_proj.bias"]).view(b, S, nh, d).transpose(1, 2)
    k = F.linear(x, w[pfx + "self_attn.k_proj.weight"], w[pfx + "self_attn.k_proj.bias"]).view(b, S, nkv, d).transpose(1, 2)
    v = F.linear(x, w[pfx + "self_attn.v_proj.weight"], w[pfx + "self_attn.v_proj.bias"]).view(b, S, nkv, d).transpose(1, 2)
    q, k = apply_rope(q, k, cos, sin)
    if cache is not None:
        k, v = cache.update(k, v, layer)
    L = k.shape[2]
    rep = nh // nkv
    kk = k[:, :, None].expand(b, nkv, rep, L, d).reshape(b, nh, L, d)
    vv = v[:, :, None].expand(b, nkv, rep, L, d).reshape(b, nh, L, d)
    att = torch.matmul(q, kk.transpose(2, 3)) * (d ** -0.5)
    att = att + _additive_mask(b, S, L, att.dtype, attention_mask)
    att = F.softmax(att, dim=-1, dtype=torch.float32).to(q.dtype)
    out = torch.matmul(att, vv).transpose(1, 2).reshape(b, S, nh * d)
    return F.linear(out, w[pfx + "self_attn.o_proj.weight"])


def qwen2_mlp(x, w, pfx):
    """Qwen2MLP.forward (modeling_qwen2.py:46-48): down(silu(gate(x)) * up(x)), no bias."""
    g = F.linear(x, w[pfx + "mlp.gate_proj.weight"])
    u = F.linear(x, w[pfx + "mlp.up_proj.weight"])
    return F.linear(F.silu(g) * u, w[pfx + "mlp.down_proj.weight"])


def qwen2_layer(x, w, i, cfg, cos, sin, cache, attention_mask=None):
    """Qwen2DecoderLayer.forward (modeling_qwen2.py:269-298): pre-norm residual x2."""
    pfx = f"model.layers.{i}."
    eps = cfg.get("rms_norm_eps", 1e-6)
    h = x + qwen2_attention(rms_norm(x, w[pfx + "input_layernorm.weight"], eps), w, pfx, cfg, cos, sin, cache, i, attention_mask)
    return h + qwen2_mlp(rms_norm(h, w[pfx + "post_attention_layernorm.weight"], eps), w, pfx)


def qwen2_model(inputs_embeds, w, cfg, cache, position_ids=None, attention_mask=None):
    """Qwen2Model.forward (modeling_qwen2.py:342-402): positions default to past_len + arange(S)
    (cache_position), shared cos/sin for all layers, layer loop, final norm."""
    b, S, _ = inputs_embeds.shape
    past = cache.get_seq_length() if cache is not None else 0
    if position_ids is None:
        position_ids = (past + torch.arange(S))[None, :].expand(b, S)
    nh = cfg["num_attention_heads"]
    d = cfg.get("head_dim") or cfg["hidden_size"] // nh
    cos, sin = rope_cos_sin(position_ids, d, cfg.get("rope_theta", 1e6), inputs_embeds.dtype)
    h = inputs_embeds
    for i in range(cfg["num_hidden_layers"]):
        h = qwen2_layer(h, w, i, cfg, cos, sin, cache, attention_mask)
    return rms_norm(h, w["model.norm.weight"], cfg.get("rms_norm_eps", 1e-6))


def lm_head(h, w):
    """Qwen2ForCausalLM.forward lm_head (modeling_qwen2.py:462-465), untied, no bias."""
    return F.linear(h, w["lm_head.weight"])
