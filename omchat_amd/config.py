"""Configuration of the hot path: vision tower, projector, decoder.

Field names follow the reference so that a checkpoint's `config.json` maps 1:1:
  vision -- InternVisionConfig (omchat/model/multimodal_encoder/intern_vit_6b/configuration_intern_vit.py:63-83)
  text   -- Qwen2Config as stored in the OmChat checkpoint (hidden 3584, 28 layers, 28/4 heads, ...)
  mm     -- the ad-hoc attributes the reference reads via getattr (omchat_arch.py:25-28,161,176;
            internVIT_encoder.py:16-17; multimodal_projector/builder.py:40; single_inference.py:46)
"""
import copy

DEFAULT_PINPOINTS = [[448, 896], [896, 448], [896, 896], [1344, 448], [448, 1344], [1344, 1344]]  # hf/image_processing_omchat.py:195-199


class OmChatConfig:
    def __init__(self, vision, text, mm=None):
        self.vision = dict(vision)
        self.text = dict(text)
        self.mm = dict(mm or {})
        self.mm.setdefault("mm_vision_tower", "internvit-6b-448px")
        self.mm.setdefault("mm_projector_type", "mlp2x_gelu")
        self.mm.setdefault("mm_vision_select_layer", -1)
        self.mm.setdefault("mm_vision_select_feature", "patch")
        self.mm.setdefault("mm_hidden_size", self.vision["hidden_size"])
        self.mm.setdefault("image_grid_pinpoints", DEFAULT_PINPOINTS)
        self.mm.setdefault("tokenizer_padding_side", "right")
        self.mm.setdefault("tokenizer_model_max_length", None)
        t = self.text
        t.setdefault("head_dim", t["hidden_size"] // t["num_attention_heads"])
        t.setdefault("rms_norm_eps", 1e-6)
        t.setdefault("rope_theta", 1e6)
        v = self.vision
        v.setdefault("patch_size", 14)
        v.setdefault("layer_norm_eps", 1e-6)
        v.setdefault("qk_normalization", True)
        v.setdefault("qkv_bias", False)
        v.setdefault("norm_type", "rms_norm")                  # 'layer_norm' for InternViT-300M (NORM2FN, intern_vit_300m/modeling_intern_vit.py:61-64)
        v.setdefault("head_dim", v["hidden_size"] // v["num_attention_heads"])
        if v["head_dim"] not in (64, 128) or v["norm_type"] not in ("rms_norm", "layer_norm") or v["qkv_bias"]:
            raise ValueError(f"unsupported vision tower geometry: head_dim={v['head_dim']} norm_type={v['norm_type']} qkv_bias={v['qkv_bias']}")

    # reference-style attribute access (model.config.image_grid_pinpoints, single_inference.py:46)
    def __getattr__(self, k):
        for d in ("mm", "text", "vision"):
            dd = self.__dict__.get(d, {})
            if k in dd:
                return dd[k]
        raise AttributeError(k)

    @property
    def num_image_tokens(self):
        g = self.vision["image_size"] // self.vision["patch_size"]
        return g * g

    def clone(self):
        return OmChatConfig(copy.deepcopy(self.vision), copy.deepcopy(self.text), copy.deepcopy(self.mm))


def omchat13b():
    """OmChat2.0-13B = InternViT-6B (45 layers as the reference instantiates it) + Qwen2-7B."""
    vision = dict(hidden_size=3200, num_attention_heads=25, intermediate_size=12800, num_hidden_layers=45,
                  patch_size=14, image_size=448, layer_norm_eps=1e-6, qk_normalization=True, qkv_bias=False)
    text = dict(hidden_size=3584, num_hidden_layers=28, num_attention_heads=28, num_key_value_heads=4,
                intermediate_size=18944, vocab_size=152064, rope_theta=1e6, rms_norm_eps=1e-6, head_dim=128)
    return OmChatConfig(vision, text)


def tiny(layers_v=2, layers_t=2, heads_v=2, q_heads=7, kv_heads=1, image_size=56, vocab=320,
         hidden_t=256, mlp_v=512, mlp_t=512):
    """Small config with the production head_dim (128) so the HIP kernels run unchanged."""
    vision = dict(hidden_size=128 * heads_v, num_attention_heads=heads_v, intermediate_size=mlp_v,
                  num_hidden_layers=layers_v, patch_size=14, image_size=image_size)
    text = dict(hidden_size=hidden_t, num_hidden_layers=layers_t, num_attention_heads=q_heads,
                num_key_value_heads=kv_heads, intermediate_size=mlp_t, vocab_size=vocab, head_dim=128)
    return OmChatConfig(vision, text)


def omchat8b_21():
    """OmChat-2.1-8B shape (BASELINE configs[3]): InternViT-300M (intern_vit_300m/configuration_intern_vit.py:60-80: 1024 hidden,
    16 heads x 64, MLP 4096, 24 layers, LayerNorm, no q/k norm) + Qwen2-7B."""
    vision = dict(hidden_size=1024, num_attention_heads=16, intermediate_size=4096, num_hidden_layers=24,
                  patch_size=14, image_size=448, layer_norm_eps=1e-6, qk_normalization=False, qkv_bias=False, norm_type="layer_norm")
    text = dict(hidden_size=3584, num_hidden_layers=28, num_attention_heads=28, num_key_value_heads=4,
                intermediate_size=18944, vocab_size=152064, rope_theta=1e6, rms_norm_eps=1e-6, head_dim=128)
    return OmChatConfig(vision, text, {"mm_vision_tower": "internvit-300m-448px"})


def tiny300m(layers_v=2, layers_t=2, heads_v=4, image_size=56, **kw):
    """tiny() with the InternViT-300M tower variant: head_dim 64, LayerNorm, no q/k norm."""
    c = tiny(layers_v=layers_v, layers_t=layers_t, heads_v=1, image_size=image_size, **kw)
    c.vision.update(hidden_size=64 * heads_v, num_attention_heads=heads_v, head_dim=64, norm_type="layer_norm", qk_normalization=False)
    c.mm["mm_vision_tower"] = "internvit-300m-448px"
    c.mm["mm_hidden_size"] = c.vision["hidden_size"]
    return c
