"""Deterministic synthetic weights / inputs (no checkpoint or dataset is reachable offline).

One counter-based generator, bit-identical on the host (numpy, here) and on the device
(`omchat_fill_uniform` in csrc/fill.hip):  for element i of tensor `name`

    h   = splitmix64(fnv1a64(name) ^ seed  +  i * 0x9E3779B97F4A7C15)
    v   = float32(int32(h >> 40) - 2^23) * float32(scale / 2^23)      # uniform in [-scale, scale)
    out = v rounded to bf16 (RNE), flushed to 0 when not exactly representable in fp16

so every value is exact in fp32, bf16 AND fp16: the oracle (fp32), the fp16 reference plumbing and the
HIP path (fp16 or bf16) all start from the same numbers.  scale = std * sqrt(3).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for c in name.encode():
        h ^= c
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def _round_bf16_fp16_exact(v):
    u = v.view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)) << np.uint32(16)
    f = r.view(np.float32)
    ok = f.astype(np.float16).astype(np.float32) == f
    return np.where(ok, f, np.float32(0.0)).astype(np.float32)


def uniform_range(name, start, count, seed=0, std=0.02, offset=0.0):
    """Elements [start, start + count) of the flattened tensor `name` (float32, 1-D): the generator is counter-based, so any slice --
    a few embedding rows of a 545 M element table, a chunk handed to a worker thread -- costs only its own elements."""
    key = np.uint64((fnv1a64(name) ^ (seed & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = key + (np.uint64(start) + np.arange(count, dtype=np.uint64)) * np.uint64(0x9E3779B97F4A7C15)
    h = _splitmix64(ctr)
    iv = (h >> np.uint64(40)).astype(np.int64) - (1 << 23)
    mul = np.float32(np.float32(std * np.sqrt(3.0)) / np.float32(1 << 23))
    v = iv.astype(np.float32) * mul
    if offset:
        v = v + np.float32(offset)
    return _round_bf16_fp16_exact(v.astype(np.float32))


def uniform(name, shape, seed=0, std=0.02, offset=0.0):
    """float32 array of `shape`; values exact in bf16 and fp16.  `offset` is added before rounding (e.g. 1.0 for
    norm weights)."""
    return uniform_range(name, 0, int(np.prod(shape)), seed, std, offset).reshape(shape)


# ---------------------------------------------------------------------------------------------------------
# tensor inventory with the reference's omchat-native key names
# ---------------------------------------------------------------------------------------------------------
TOWER = "model.vision_tower.vision_tower."


def tensor_specs(cfg):
    """[(key, shape, std, offset)] for the whole OmChat state dict."""
    v, t = cfg.vision, cfg.text
    C, I, p = v["hidden_size"], v["intermediate_size"], v["patch_size"]
    ntok = (v["image_size"] // p) ** 2 + 1
    specs = [
        (TOWER + "embeddings.class_embedding", (1, 1, C), 0.02, 0.0),
        (TOWER + "embeddings.position_embedding", (1, ntok, C), 0.02, 0.0),
        (TOWER + "embeddings.patch_embedding.weight", (C, 3, p, p), 0.02, 0.0),
        (TOWER + "embeddings.patch_embedding.bias", (C,), 0.02, 0.0),
    ]
    for j in range(v["num_hidden_layers"]):
        P = TOWER + f"encoder.layers.{j}."
        specs += [
            (P + "ls1", (C,), 0.02, 0.1), (P + "ls2", (C,), 0.02, 0.1),
            (P + "norm1.weight", (C,), 0.05, 1.0), (P + "norm2.weight", (C,), 0.05, 1.0),
            (P + "attn.qkv.weight", (3 * C, C), 0.02, 0.0)]
        if v.get("norm_type", "rms_norm") == "layer_norm":
            specs += [(P + "norm1.bias", (C,), 0.02, 0.0), (P + "norm2.bias", (C,), 0.02, 0.0)]
        if v.get("qk_normalization", True):
            specs += [(P + "attn.q_norm.weight", (C,), 0.05, 1.0), (P + "attn.k_norm.weight", (C,), 0.05, 1.0)]
        specs += [
            (P + "attn.proj.weight", (C, C), 0.02, 0.0), (P + "attn.proj.bias", (C,), 0.02, 0.0),
            (P + "mlp.fc1.weight", (I, C), 0.02, 0.0), (P + "mlp.fc1.bias", (I,), 0.02, 0.0),
            (P + "mlp.fc2.weight", (C, I), 0.02, 0.0), (P + "mlp.fc2.bias", (C,), 0.02, 0.0),
        ]
    H, V, It = t["hidden_size"], t["vocab_size"], t["intermediate_size"]
    nh, nkv, d = t["num_attention_heads"], t["num_key_value_heads"], t["head_dim"]
    specs += [
        ("model.mm_projector.0.weight", (H, C), 0.02, 0.0), ("model.mm_projector.0.bias", (H,), 0.02, 0.0),
        ("model.mm_projector.2.weight", (H, H), 0.02, 0.0), ("model.mm_projector.2.bias", (H,), 0.02, 0.0),
        ("model.embed_tokens.weight", (V, H), 0.02, 0.0),
    ]
    for i in range(t["num_hidden_layers"]):
        P = f"model.layers.{i}."
        specs += [
            (P + "self_attn.q_proj.weight", (nh * d, H), 0.02, 0.0), (P + "self_attn.q_proj.bias", (nh * d,), 0.02, 0.0),
            (P + "self_attn.k_proj.weight", (nkv * d, H), 0.02, 0.0), (P + "self_attn.k_proj.bias", (nkv * d,), 0.02, 0.0),
            (P + "self_attn.v_proj.weight", (nkv * d, H), 0.02, 0.0), (P + "self_attn.v_proj.bias", (nkv * d,), 0.02, 0.0),
            (P + "self_attn.o_proj.weight", (H, nh * d), 0.02, 0.0),
            (P + "mlp.gate_proj.weight", (It, H), 0.02, 0.0), (P + "mlp.up_proj.weight", (It, H), 0.02, 0.0),
            (P + "mlp.down_proj.weight", (H, It), 0.02, 0.0),
            (P + "input_layernorm.weight", (H,), 0.05, 1.0), (P + "post_attention_layernorm.weight", (H,), 0.05, 1.0),
        ]
    specs += [("model.norm.weight", (H,), 0.05, 1.0), ("lm_head.weight", (V, H), 0.02, 0.0)]
    return specs


def state_dict(cfg, seed=0, only_prefix=None):
    """numpy fp32 state dict (host generation; use the device fill for 13B-scale benches)."""
    out = {}
    for key, shape, std, off in tensor_specs(cfg):
        if only_prefix is not None and not key.startswith(only_prefix):
            continue
        out[key] = uniform(key, shape, seed, std, off)
    return out


def pixels(n_tiles, image_size, seed=0):
    """Synthetic tiles: uniform uint8 RGB, ImageNet-normalised (internVIT_encoder.py:26-29), then made exact in
    bf16/fp16 like the weights.  Returns float32 [n, 3, S, S]."""
    n = n_tiles * 3 * image_size * image_size
    key = np.uint64((fnv1a64("pixels") ^ seed) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        h = _splitmix64(key + np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
    u8 = (h >> np.uint64(56)).astype(np.float32).reshape(n_tiles, 3, image_size, image_size)
    mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 3, 1, 1)
    return _round_bf16_fp16_exact(((u8 / np.float32(255.0) - mean) / std).astype(np.float32))


def token_ids(n, vocab, seed=1):
    key = np.uint64((fnv1a64("token_ids") ^ seed) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        h = _splitmix64(key + np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
    return (h % np.uint64(vocab)).astype(np.int64)
