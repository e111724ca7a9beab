#!/usr/bin/env python3
"""Golden fixture for the HF image processor (omchat/hf/image_processing_omchat.py::OmChatImageProcessor), captured from the
reference imported in the build container: per seeded random picture the tile count, the canvas the tiles come from and the
sha256 of the fp32 pixel bytes; plus token layouts of OmChatProcessor for a stub tokenizer.  -> tests/golden/hf_image_processor.json"""
import hashlib, json, os, sys, types
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import import_reference  # noqa: E402

CASES = [(570, 380, 0), (333, 999, 1), (448, 448, 2), (1000, 1000, 3), (700, 1500, 4), (1344, 448, 5), (100, 37, 6)]


class Tok:      # the same stub tokenizer as tests/test_host_cpu.py
    bos_token_id = None
    pad_token_id = 0

    def __call__(self, s):
        return types.SimpleNamespace(input_ids=[1000 + ord(ch) for ch in s])

    def encode(self, s):
        return [1000 + ord(ch) for ch in s]


def main():
    import_reference()
    from PIL import Image
    from omchat.hf.image_processing_omchat import OmChatImageProcessor
    from omchat.hf.processing_omchat import OmChatProcessor
    ip = OmChatImageProcessor()
    out = {"pinpoints": ip.image_grid_pinpoints, "cases": [], "prompts": []}
    for w, h, seed in CASES:
        a = np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)
        r = ip(Image.fromarray(a), return_tensors="pt")
        n = int(r["num_patches"][0])
        pv = np.ascontiguousarray(r["pixel_values"][0, :n].numpy())
        out["cases"].append({"w": w, "h": h, "seed": seed, "n": n, "sha256": hashlib.sha256(pv.tobytes()).hexdigest()})
    proc = OmChatProcessor.__new__(OmChatProcessor)          # ProcessorMixin.__init__ type-checks the tokenizer class; bypass it
    proc.image_processor, proc.tokenizer = ip, Tok()
    imgs = [Image.fromarray(np.random.default_rng(s).integers(0, 256, (h, w, 3), dtype=np.uint8)) for w, h, s in CASES[:2]]
    for text, images in (("What is this?", imgs[0]), ("first <image> then <image> compare", imgs)):
        r = OmChatProcessor.__call__(proc, text=text, images=images)
        out["prompts"].append({"text": text, "n_images": 1 if not isinstance(images, list) else len(images),
                               "input_ids": r["input_ids"][0].tolist(), "images_shape": list(r["images"].shape)})
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hf_image_processor.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
