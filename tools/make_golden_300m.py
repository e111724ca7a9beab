#!/usr/bin/env python3
"""Golden vectors for the InternViT-300M tower variant, captured from the reference itself
(omchat/model/multimodal_encoder/intern_vit_300m/modeling_intern_vit.py, internVIT300m_encoder.py) imported in the build
container with the same stubs as tools/make_golden.py.  Writes tests/golden/vit300m_tiny.npz and
tests/golden/tower300m_wrapper_*.npz.  Run from the repo root: python tools/make_golden_300m.py"""
import os, sys, types
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import import_reference, T, save  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    import_reference()
    from omchat_amd import synth
    from omchat_amd.config import tiny300m
    from omchat.model.multimodal_encoder.intern_vit_300m.modeling_intern_vit import InternVisionModel
    from omchat.model.multimodal_encoder.intern_vit_300m.configuration_intern_vit import InternVisionConfig
    import omchat.model.multimodal_encoder.internVIT300m_encoder as enc_mod
    TOW = synth.TOWER
    cfg = tiny300m()
    keys = ("hidden_size", "num_attention_heads", "intermediate_size", "num_hidden_layers", "patch_size", "image_size",
            "layer_norm_eps", "qk_normalization", "qkv_bias", "norm_type")
    vc = InternVisionConfig(**{**{k: cfg.vision[k] for k in keys}, "use_flash_attn": False})
    tower = InternVisionModel(vc).eval()
    sd = synth.state_dict(cfg, seed=0, only_prefix=TOW)
    tower.load_state_dict({k[len(TOW):]: T(v) for k, v in sd.items()}, strict=True)
    px = T(synth.pixels(2, cfg.vision["image_size"], seed=3))
    out = tower(px, output_hidden_states=True, return_dict=True)
    lay0 = tower.encoder.layers[0]
    h0 = out.hidden_states[0]
    n1 = lay0.norm1(h0)
    save("vit300m_tiny", pixels=px, seed=0, pixel_seed=3, hs0=out.hidden_states[0], hs1=out.hidden_states[1], hs2=out.hidden_states[2],
         l0_norm1=n1, l0_attn=lay0.attn(n1), l0_mlp=lay0.mlp(lay0.norm2(h0)))

    orig = enc_mod.InternVisionConfig
    enc_mod.InternVisionConfig = lambda *a, **k: vc
    try:
        for sel_layer, sel_feat in ((-1, "patch"), (-2, "cls_patch")):
            args = types.SimpleNamespace(mm_vision_select_layer=sel_layer, mm_vision_select_feature=sel_feat)
            tw = enc_mod.InternVIT300mVisionTower("internvit-300m-448px", args, delay_load=False)
            tw.vision_tower.load_state_dict({k[len(TOW):]: T(v) for k, v in sd.items()}, strict=True)
            tw.vision_tower.half()
            feats = tw(px.half())
            save(f"tower300m_wrapper_L{sel_layer}_{sel_feat}", pixels=px, feats_half=feats.float(), select_layer=sel_layer, select_feature=sel_feat)
    finally:
        enc_mod.InternVisionConfig = orig


if __name__ == "__main__":
    main()
