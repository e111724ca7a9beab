"""Host-side (CPU, integer) helpers on the caller side of the hot path; counterparts of the reference's
omchat/mm_utils.py.  They decide the ViT batch shape (tile count + order) and the `-200` sentinel layout.  Pixel work (resize /
pad / tile / normalise) is NOT here: it runs on the device (image_processing.py -> csrc/preproc.hip)."""
import math
from .constants import IMAGE_TOKEN_INDEX


def select_best_resolution(original_size, possible_resolutions):
    """Counterpart of mm_utils.py:12-39: maximise the effective (downscaled) resolution, then minimise waste;
    first candidate wins ties."""
    ow, oh = original_size
    best, best_eff, best_waste = None, 0, float("inf")
    for w, h in possible_resolutions:
        s = min(w / ow, h / oh)
        dw, dh = int(ow * s), int(oh * s)
        eff = min(dw * dh, ow * oh)
        waste = w * h - eff
        if eff > best_eff or (eff == best_eff and waste < best_waste):
            best, best_eff, best_waste = (w, h), eff, waste
    return best


def padded_size(original_size, target_resolution):
    """Size of the aspect-preserving resize inside resize_and_pad_image (mm_utils.py:42-74)."""
    ow, oh = original_size
    tw, th = target_resolution
    sw, sh = tw / ow, th / oh
    if sw < sh:
        return tw, min(math.ceil(oh * sw), th)
    return min(math.ceil(ow * sh), tw), th


def anyres_tile_count(image_size, grid_pinpoints, tile=448):
    w, h = select_best_resolution(image_size, grid_pinpoints)
    return 1 + (w // tile) * (h // tile)


def process_anyres_image(image, processor, grid_pinpoints, return_type_list=False, return_best_res=False):
    """mm_utils.py:119-158: [thumbnail] + tiles of the resized / padded canvas, normalised.  The pixels are produced on the device
    by `processor` (image_processing.HipImageProcessor -> csrc/preproc.hip, bit-identical to the reference's PIL + CLIPImageProcessor
    result); there is no CPU path in the product -- the PIL restatement lives in oracle/preproc.py for the tests."""
    if not hasattr(processor, "process_anyres"):
        raise TypeError("process_anyres_image needs the device image processor (omchat_amd.image_processing.HipImageProcessor); "
                        "the HIP path has no CPU fallback")
    out, best = processor.process_anyres(image, grid_pinpoints, return_best_res=True)
    out = list(out) if return_type_list else out
    return (out, best) if return_best_res else out


def find_closest_aspect_ratio(aspect_ratio, target_ratios, width, height, image_size):
    """mm_utils.py:326-339."""
    best_diff, best = float("inf"), (1, 1)
    area = width * height
    for r in target_ratios:
        diff = abs(aspect_ratio - r[0] / r[1])
        if diff < best_diff:
            best_diff, best = diff, r
        elif diff == best_diff and area > 0.5 * image_size * image_size * r[0] * r[1]:
            best = r
    return best


def dynamic_grid(size, min_num=1, max_num=6, image_size=448):
    """Grid (cols, rows) that dynamic_preprocess (mm_utils.py:276-292) resizes a (width, height) picture to."""
    w, h = size
    ratios = set((i, j) for n in range(min_num, max_num + 1) for i in range(1, n + 1) for j in range(1, n + 1)
                 if min_num <= i * j <= max_num)
    ratios = sorted(ratios, key=lambda x: x[0] * x[1])
    return find_closest_aspect_ratio(w / h, ratios, w, h, image_size)


def process_dynamic_image(image, processor, max_num=6, image_size=336, grid_pinpoints=None, return_type_list=False, return_best_res=False):
    """mm_utils.py:315-323 (dynamic_preprocess :276-312 + per-tile preprocess) on the device; no CPU path in the product."""
    if not hasattr(processor, "process_dynamic"):
        raise TypeError("process_dynamic_image needs the device image processor (omchat_amd.image_processing.HipImageProcessor)")
    out = processor.process_dynamic(image, max_num=max_num, image_size=image_size)
    out = list(out) if return_type_list else out
    return (out, None) if return_best_res else out


def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, return_tensors=None):
    """mm_utils.py:197-230 (the `<image>` branch): tokenise the chunks between `<image>` markers and join them
    with the sentinel; a leading BOS (none for Qwen2) is kept once."""
    chunks = [tokenizer(c).input_ids for c in prompt.split("<image>")]
    ids, offset = [], 0
    if chunks and chunks[0] and chunks[0][0] == getattr(tokenizer, "bos_token_id", None):
        offset = 1
        ids.append(chunks[0][0])
    for i, c in enumerate(chunks):
        ids.extend(c[offset:])
        if i < len(chunks) - 1:
            ids.append(image_token_index)
    if return_tensors == "pt":
        import torch
        return torch.tensor(ids, dtype=torch.long)
    if return_tensors is not None:
        raise ValueError(f"Unsupported tensor type: {return_tensors}")
    return ids


def get_model_name_from_path(model_path):
    """mm_utils.py:233-239."""
    parts = model_path.strip("/").split("/")
    return parts[-2] + "_" + parts[-1] if parts[-1].startswith("checkpoint-") else parts[-1]


class KeywordsStoppingCriteria:
    """mm_utils.py:242-274: stop when the tail of the generated ids equals a keyword's ids, or a keyword appears in the decoded
    tail.  Callable as HF StoppingCriteria: criteria(output_ids [b, T], scores) -> bool (all sequences hit)."""

    def __init__(self, keywords, tokenizer, input_ids):
        import torch
        self.keywords = keywords
        self.keyword_ids = []
        self.max_keyword_len = 0
        for keyword in keywords:
            ids = tokenizer(keyword).input_ids
            if len(ids) > 1 and ids[0] == tokenizer.bos_token_id:
                ids = ids[1:]
            self.max_keyword_len = max(self.max_keyword_len, len(ids))
            self.keyword_ids.append(torch.tensor(ids))
        self.tokenizer = tokenizer
        self.start_len = input_ids.shape[1]

    def call_for_batch(self, output_ids, scores, **kwargs):
        import torch
        offset = min(output_ids.shape[1] - self.start_len, self.max_keyword_len)
        for kid in self.keyword_ids:
            if torch.equal(output_ids[0, -kid.shape[0]:].cpu(), kid):
                return True
        outputs = self.tokenizer.batch_decode(output_ids[:, -offset:], skip_special_tokens=True)[0]
        return any(k in outputs for k in self.keywords)

    def __call__(self, output_ids, scores=None, **kwargs):
        return all(self.call_for_batch(output_ids[i].unsqueeze(0), scores) for i in range(output_ids.shape[0]))
