"""HF-format front end: counterparts of omchat/hf/image_processing_omchat.py::OmChatImageProcessor and
omchat/hf/processing_omchat.py::OmChatProcessor (:143-257), as driven by hf_example.py:
    inputs = processor(text=prompt, images=image, return_tensors="pt"); model.generate(**inputs, ...)
The pixels come from the device front end (csrc/preproc.hip); the prompt layout is the reference's make_context."""
from .image_processing import HipImageProcessor
from .make_context import make_context
from .config import DEFAULT_PINPOINTS


class OmChatImageProcessor(HipImageProcessor):
    """OmChatImageProcessor (hf/image_processing_omchat.py:195-199 defaults, :466-528 patching, :569-733 preprocess): anyres tiles
    per image, thumbnail first.  Unlike omchat/mm_utils.py, the HF processor reads `image_grid_pinpoints` as (height, width)
    pairs -- the candidate set is the same but ties break differently (a square picture becomes 896 wide x 448 high here,
    448 x 896 there); verified against the imported reference in tests/golden/hf_image_processor.json."""

    def __init__(self, crop_size=448, image_grid_pinpoints=None, **kw):
        super().__init__(crop_size=crop_size, **kw)
        self.image_grid_pinpoints = [list(p) for p in (image_grid_pinpoints or DEFAULT_PINPOINTS)]

    def _pins_wh(self):
        return [(p[1], p[0]) for p in self.image_grid_pinpoints]

    def __call__(self, images, return_tensors="pt", dtype=None, **kw):
        """{"pixel_values": [n_images, max_patches, 3, H, W] (zero padded), "num_patches": int64 [n_images]} on the device."""
        import torch
        imgs = images if isinstance(images, (list, tuple)) else [images]
        tiles = [self.process_anyres(im, self._pins_wh(), dtype=dtype) for im in imgs]
        n = torch.tensor([t.shape[0] for t in tiles], dtype=torch.int64)
        pv = torch.zeros(len(tiles), int(n.max()), *tiles[0].shape[1:], dtype=tiles[0].dtype, device=tiles[0].device)
        for i, t in enumerate(tiles):
            pv[i, :t.shape[0]] = t
        return {"pixel_values": pv, "num_patches": n}

    preprocess = __call__


def _batch_feature(data):
    """BatchFeature like the reference returns (hf/processing_omchat.py:253-257): attribute access (`inputs.input_ids`,
    hf_example.py:17), `.to("cuda")` (:12), `**inputs` into generate (:15)."""
    from transformers import BatchFeature
    return BatchFeature(data=data)


class OmChatProcessor:
    """OmChatProcessor.__call__ (hf/processing_omchat.py:171-253): one sample; `images` = one picture or a list; returns a
    BatchFeature {"input_ids": int64 [1, T] with one -200 per tile, "images": [sum(tiles), 3, H, W]}.  Text-only prompts return
    {"input_ids"} (the reference builds a bare tensor there and then fails in BatchFeature(**tensor); a mapping is the evident intent)."""

    def __init__(self, image_processor=None, tokenizer=None, **kw):
        self.image_processor = image_processor
        self.tokenizer = tokenizer

    @classmethod
    def from_pretrained(cls, path, trust_remote_code=None, **kw):
        """AutoProcessor.from_pretrained(path, trust_remote_code=True) (hf_example.py:8): tokenizer from the checkpoint directory,
        image geometry from its preprocessor_config.json / config.json when present, the reference's defaults otherwise."""
        import json, os
        from transformers import AutoTokenizer
        try:
            tok = AutoTokenizer.from_pretrained(path, use_fast=False)
        except Exception:
            tok = AutoTokenizer.from_pretrained(path)
        crop, pins = 448, None
        for fn in ("preprocessor_config.json", "config.json"):
            f = os.path.join(path, fn)
            if os.path.exists(f):
                j = json.load(open(f))
                pins = pins or j.get("image_grid_pinpoints")
                cs = j.get("crop_size")
                if isinstance(cs, dict):
                    crop = int(cs.get("height", crop))
                elif isinstance(cs, int):
                    crop = cs
                elif fn == "config.json" and isinstance(j.get("vision_config"), dict):
                    crop = int(j["vision_config"].get("image_size", crop))
        return cls(OmChatImageProcessor(crop_size=crop, image_grid_pinpoints=pins), tok)

    def __call__(self, text, images=None, padding=False, truncation=None, max_length=None, return_tensors="pt"):
        import torch
        system = "You are a helpful assistant."
        if images is None:
            _, ids = make_context(self.tokenizer, text.replace("<image>", "").strip(), None, system)
            return _batch_feature({"input_ids": torch.tensor([ids])})
        out = self.image_processor(images, return_tensors=return_tensors)
        n_per = out["num_patches"].tolist()
        tiles = [out["pixel_values"][i, :n] for i, n in enumerate(n_per)]           # split_tensor (:133-141)
        patch_block = lambda n: "<image>\n" + "\n".join(["patch:<image>"] * (n - 1))
        if len(tiles) == 1:
            query = patch_block(n_per[0]) + "\n" + text.replace("<image>", "").strip()
        else:                                                                      # :235-241: one block per picture, text pieces between
            parts = text.split("<image>")
            query = parts[0]
            for i, n in enumerate(n_per):
                query += patch_block(n)
                if i + 1 < len(parts):
                    query += parts[i + 1]
            query = query.strip()
        _, ids = make_context(self.tokenizer, query, None, system)
        return _batch_feature({"input_ids": torch.tensor([ids]), "images": torch.cat(tiles, dim=0)})

    def batch_decode(self, *a, **k):
        return self.tokenizer.batch_decode(*a, **k)

    def decode(self, *a, **k):
        return self.tokenizer.decode(*a, **k)
