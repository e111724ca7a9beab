"""Counterpart of omchat/model/multimodal_encoder/internVIT_encoder.py + builder.py + base_encoder.py."""
import torch


class InternVITVisionTower:
    """InternVITVisionTower(vision_tower, args, delay_load) (internVIT_encoder.py:9-84): owns the CLIP-style image
    processor (448 px, ImageNet mean/std, :25-29) and runs the tower on the HIP engine.  `forward(images)` returns
    `hidden_states[select_layer]` without CLS for 'patch' in the dtype of `images` (:35-56)."""

    def __init__(self, vision_tower, args, delay_load=False, engine=None):
        self.is_loaded = False
        self.vision_tower_name = vision_tower
        self.select_layer = args.mm_vision_select_layer
        self.select_feature = getattr(args, "mm_vision_select_feature", "patch")
        self.engine = engine
        self._cfg = engine.cfg.vision if engine is not None else None
        if not delay_load:
            self.load_model()

    def load_model(self, is_train=False):
        from ..image_processing import HipImageProcessor
        crop = 448 if "448" in self.vision_tower_name else 336
        # same parameters as the reference's CLIPImageProcessor (:25-29); the arithmetic runs in csrc/preproc.hip
        self.image_processor = HipImageProcessor(crop_size=crop, image_mean=[0.485, 0.456, 0.406], image_std=[0.229, 0.224, 0.225],
                                                 device=self.engine.device if self.engine is not None else None)
        self.is_loaded = True

    def feature_select(self, hidden_state):
        if self.select_feature == "patch":
            return hidden_state[:, 1:]
        if self.select_feature == "cls_patch":
            return hidden_state
        raise ValueError(f"Unexpected select feature: {self.select_feature}")

    def forward(self, images):
        if self.engine is None:
            raise RuntimeError("InternVITVisionTower has no HIP engine attached (no CPU fallback)")
        if type(images) is list:                      # :46-51 per-image list branch
            return [self.engine.vit_forward(im.unsqueeze(0), self.select_layer, self.select_feature).to(im.dtype) for im in images]
        return self.engine.vit_forward(images, self.select_layer, self.select_feature).to(images.dtype)

    __call__ = forward

    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)

    @property
    def dtype(self):
        return torch.float                              # reference quirk kept (:62-64)

    @property
    def device(self):
        return self.engine.device

    @property
    def config(self):
        return self._cfg

    @property
    def hidden_size(self):
        return self._cfg["hidden_size"]

    @property
    def num_patches(self):
        return (self._cfg["image_size"] // self._cfg["patch_size"]) ** 2


class InternVIT300mVisionTower(InternVITVisionTower):
    """InternVIT300mVisionTower (internVIT300m_encoder.py:10-84): same wrapper contract as the 6B tower (CLIP-style processor,
    hidden_states[select_layer], CLS dropped for 'patch', fp16 cast of the pixels :52); the tower behind it is the
    LayerNorm / 16 x 64-head / no-q-k-norm variant, selected by the engine's vision config."""

    def __init__(self, vision_tower, args, delay_load=False, engine=None):
        if engine is not None and (engine.cfg.vision.get("norm_type") != "layer_norm" or engine.cfg.vision.get("head_dim") != 64):
            raise ValueError("InternVIT300mVisionTower needs an engine built with the InternViT-300M vision config")
        super().__init__(vision_tower, args, delay_load=delay_load, engine=engine)


def build_vision_tower(vision_tower_cfg, engine=None, **kwargs):
    """multimodal_encoder/builder.py:7-18: dispatch on the tower name (300m before 6b, as the reference); the CLIP / SigLIP
    towers are outside the hot path."""
    name = getattr(vision_tower_cfg, "mm_vision_tower", getattr(vision_tower_cfg, "vision_tower", None))
    if name is not None and "internvit-300m" in name.lower():
        return InternVIT300mVisionTower(name, args=vision_tower_cfg, engine=engine, **kwargs)
    if name is not None and "internvit-6b" in name.lower():
        return InternVITVisionTower(name, args=vision_tower_cfg, engine=engine, **kwargs)
    raise ValueError(f"Unknown vision tower: {name} (this build implements the internvit-6b and internvit-300m towers)")
