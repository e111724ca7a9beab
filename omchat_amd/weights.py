"""Checkpoint key layouts and host-side weight preparation.

Two on-disk layouts exist (convert_omchat_to_hf.py:26-35): omchat-native (`model.vision_tower.vision_tower.*`,
`model.mm_projector.{0,2}.*`, `model.layers.*`, `lm_head.weight`) and HF-format (`vision_tower.*`,
`multi_modal_projector.linear_{1,2}.*`, `language_model.*`).  The C ABI speaks omchat-native; HF keys are mapped
back here (the inverse of the reference's KEYS_TO_MODIFY_MAPPING)."""
import numpy as np

TOWER = "model.vision_tower.vision_tower."
_HF_TO_NATIVE = [
    ("language_model.model.", "model."),
    ("language_model.lm_head.", "lm_head."),
    ("multi_modal_projector.linear_1.", "model.mm_projector.0."),
    ("multi_modal_projector.linear_2.", "model.mm_projector.2."),
    ("vision_tower.", TOWER),
]


def to_native_key(k):
    if k.startswith("model.") or k.startswith("lm_head."):
        return k
    for a, b in _HF_TO_NATIVE:
        if k.startswith(a):
            return b + k[len(a):]
    return k


def to_hf_key(k):
    """convert_omchat_to_hf.py:26-35,49-59 (key rewrite only)."""
    for hf, nat in [(a, b) for a, b in _HF_TO_NATIVE if b != "model."]:
        if k.startswith(nat):
            return hf + k[len(nat):]
    if k.startswith("model."):
        return "language_model.model." + k[len("model."):]
    return k


def resize_pos_embed(pos, grid_src, grid_dst):
    """_get_pos_embed (modeling_intern_vit.py:82-88): fp32 bicubic resize of the patch rows (CLS row untouched).
    Identity at the native 448 px grid (N9), so it is folded into the loaded constant once."""
    import torch
    import torch.nn.functional as F
    p = torch.as_tensor(np.asarray(pos, dtype=np.float32)) if not torch.is_tensor(pos) else pos.float()
    p = p.reshape(1, -1, p.shape[-1])
    if grid_src == grid_dst:
        return p
    cls, patch = p[:, :1], p[:, 1:]
    patch = patch.reshape(1, grid_src, grid_src, -1).permute(0, 3, 1, 2)
    patch = F.interpolate(patch, size=(grid_dst, grid_dst), mode="bicubic", align_corners=False)
    patch = patch.reshape(1, -1, grid_dst * grid_dst).permute(0, 2, 1)
    return torch.cat([cls, patch], dim=1)


def prepare_state_dict(sd, cfg, vision=True, text=True):
    """Normalise keys to omchat-native, drop rotary buffers, keep only what the context holds, and bring the
    position embedding to the context's patch grid."""
    out = {}
    for k, v in sd.items():
        if k.endswith("inv_freq"):
            continue                                    # convert_omchat_to_hf.py:52-53
        k = to_native_key(k)
        is_v = k.startswith(TOWER) or k.startswith("model.mm_projector.")
        if (is_v and not vision) or (not is_v and not text):
            continue
        out[k] = v
    pk = TOWER + "embeddings.position_embedding"
    if pk in out and len(out[pk].shape) >= 2 and int(np.prod(out[pk].shape[:-1])) > 1:
        g_dst = cfg.vision["image_size"] // cfg.vision["patch_size"]
        n = int(np.prod(out[pk].shape[:-1])) - 1
        g_src = int(round(n ** 0.5))
        if g_src != g_dst:
            out[pk] = resize_pos_embed(out[pk], g_src, g_dst)
    return out
