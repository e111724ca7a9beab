"""Oracle: InternViT-6B vision tower + mlp2x_gelu projector (torch CPU).  Test infrastructure only.

Weights are passed as a flat dict with the reference's omchat-native checkpoint key names *relative to the
tower* (`embeddings.class_embedding`, `encoder.layers.{j}.attn.qkv.weight`, ...).
All functions compute in the dtype of their inputs exactly where the reference does (fp32 inside RMSNorm,
everything else in the activation dtype), so running them on fp16 tensors reproduces the reference's
fp16-on-CPU plumbing and running them on fp32 tensors gives the fp32 oracle.
"""
import math
import torch
import torch.nn.functional as F

REF = "omchat/model/multimodal_encoder/intern_vit_6b/modeling_intern_vit.py"


def rms_norm(x, weight, eps=1e-6):
    """InternRMSNorm.forward (modeling_intern_vit.py:39-44) == Qwen2RMSNorm.forward (modeling_qwen2.py:247-252):
    upcast to fp32, x * rsqrt(mean(x^2) + eps), downcast to the input dtype, THEN multiply by weight."""
    in_dtype = x.dtype
    h = x.to(torch.float32)
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + eps)
    return weight * h.to(in_dtype)


def _pos_embed(pos, grid_src, H, W):
    """_get_pos_embed (modeling_intern_vit.py:82-88): fp32 bicubic resize of the patch position table."""
    dt = pos.dtype
    p = pos.float().reshape(1, grid_src, grid_src, -1).permute(0, 3, 1, 2)
    p = F.interpolate(p, size=(H, W), mode="bicubic", align_corners=False)
    return p.reshape(1, -1, H * W).permute(0, 2, 1).to(dt)


def vit_embeddings(pixels, w, patch=14, image_size=448):
    """InternVisionEmbeddings.forward (modeling_intern_vit.py:90-102).
    Conv2d(k=patch, s=patch) restated as unfold + matmul (identical arithmetic up to summation order)."""
    Wc = w["embeddings.patch_embedding.weight"]            # [C, 3, p, p]
    bc = w["embeddings.patch_embedding.bias"]
    B, _, Hp, Wp = pixels.shape
    gh, gw = Hp // patch, Wp // patch
    cols = F.unfold(pixels, kernel_size=patch, stride=patch)            # [B, 3*p*p, gh*gw]
    pe = cols.transpose(1, 2) @ Wc.reshape(Wc.shape[0], -1).t() + bc    # [B, gh*gw, C]
    cls = w["embeddings.class_embedding"].expand(B, 1, -1).to(pe.dtype)
    emb = torch.cat([cls, pe], dim=1)
    pos = w["embeddings.position_embedding"]
    pos = torch.cat([pos[:, :1, :], _pos_embed(pos[:, 1:, :], image_size // patch, gh, gw)], dim=1)
    return emb + pos.to(pe.dtype)


def vit_attention(x, w, pfx, num_heads, eps=1e-6):
    """InternAttention._naive_attn (modeling_intern_vit.py:138-155): fused qkv (no bias), joint-head RMSNorm
    on q and k over all C channels (:143-146), (q*scale) @ k^T, softmax in the input dtype, @ v, proj(+bias)."""
    B, N, C = x.shape
    D = C // num_heads
    qkv = F.linear(x, w[pfx + "attn.qkv.weight"], w.get(pfx + "attn.qkv.bias"))
    qkv = qkv.reshape(B, N, 3, num_heads, D).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)                                            # [B, H, N, D]
    if (pfx + "attn.q_norm.weight") in w:
        q = rms_norm(q.transpose(1, 2).flatten(-2, -1), w[pfx + "attn.q_norm.weight"], eps).view(B, N, num_heads, D).transpose(1, 2)
        k = rms_norm(k.transpose(1, 2).flatten(-2, -1), w[pfx + "attn.k_norm.weight"], eps).view(B, N, num_heads, D).transpose(1, 2)
    attn = (q * (D ** -0.5)) @ k.transpose(-2, -1)
    attn = attn.softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(out, w[pfx + "attn.proj.weight"], w[pfx + "attn.proj.bias"])


def vit_mlp(x, w, pfx):
    """InternMLP.forward (modeling_intern_vit.py:187-191): fc1(+b) -> exact erf GELU -> fc2(+b)."""
    h = F.linear(x, w[pfx + "mlp.fc1.weight"], w[pfx + "mlp.fc1.bias"])
    h = F.gelu(h)
    return F.linear(h, w[pfx + "mlp.fc2.weight"], w[pfx + "mlp.fc2.bias"])


def vit_layer(x, w, j, num_heads, eps=1e-6):
    """InternVisionEncoderLayer.forward (modeling_intern_vit.py:210-222); drop_path is Identity at rate 0."""
    pfx = f"encoder.layers.{j}."

    def norm(h, name):
        # NORM2FN (intern_vit_300m/modeling_intern_vit.py:61-64,209-210): nn.LayerNorm when the checkpoint carries a norm bias
        if (pfx + name + ".bias") in w:
            return F.layer_norm(h, (h.shape[-1],), w[pfx + name + ".weight"], w[pfx + name + ".bias"], eps)
        return rms_norm(h, w[pfx + name + ".weight"], eps)
    x = x + vit_attention(norm(x, "norm1"), w, pfx, num_heads, eps) * w[pfx + "ls1"]
    x = x + vit_mlp(norm(x, "norm2"), w, pfx) * w[pfx + "ls2"]
    return x


def vit_encoder(emb, w, num_layers, num_heads, eps=1e-6, upto=None):
    """InternVisionEncoder.forward (modeling_intern_vit.py:268-282): returns the tuple of hidden states
    (index 0 = embeddings, index L = output of the last layer)."""
    states = [emb]
    h = emb
    n = num_layers if upto is None else upto
    for j in range(n):
        h = vit_layer(h, w, j, num_heads, eps)
        states.append(h)
    return states


def vision_tower_forward(pixels, w, cfg, select_layer=-1, select_feature="patch", tower_dtype=None):
    """InternVITVisionTower.forward + feature_select (internVIT_encoder.py:35-56): the reference casts pixels
    to fp16 (:53) -- here `tower_dtype` (None = keep the weights' dtype), runs the tower with
    output_hidden_states=True, picks hidden_states[select_layer], drops CLS for 'patch', casts back."""
    in_dtype = pixels.dtype
    wd = w["embeddings.patch_embedding.weight"].dtype if tower_dtype is None else tower_dtype
    L = cfg["num_hidden_layers"]
    # only the layers that feed hidden_states[select_layer] are needed
    idx = select_layer if select_layer >= 0 else L + 1 + select_layer
    emb = vit_embeddings(pixels.to(wd), w, cfg["patch_size"], cfg["image_size"])
    states = vit_encoder(emb, w, L, cfg["num_attention_heads"], cfg.get("layer_norm_eps", 1e-6), upto=idx)
    feats = states[idx]
    if select_feature == "patch":
        feats = feats[:, 1:]
    elif select_feature != "cls_patch":
        raise ValueError(f"Unexpected select feature: {select_feature}")
    return feats.to(in_dtype)


def projector_forward(x, w):
    """build_vision_projector('mlp2x_gelu') (omchat/model/multimodal_projector/builder.py:54-61):
    Linear(mm_hidden->hidden)+b, exact GELU, Linear(hidden->hidden)+b.  Keys `0.weight`, `0.bias`, `2.weight`, `2.bias`."""
    h = F.linear(x, w["0.weight"], w["0.bias"])
    h = F.gelu(h)
    return F.linear(h, w["2.weight"], w["2.bias"])
