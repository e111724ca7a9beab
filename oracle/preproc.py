"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the image front-end that feeds the ViT.

The reference does this step with Pillow (`Image.resize`, default BICUBIC) + transformers' CLIPImageProcessor
(omchat/mm_utils.py:42-74,119-158; internVIT_encoder.py:25-29).  Pillow is a third-party dependency that is not vendored
under /root/reference (this image: Pillow 12.2.0); its 8-bit resampler (src/libImaging/Resample.c: precompute_coeffs,
normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc) is restated here in numpy and pinned in
tests/test_preproc_cpu.py against Pillow itself on seeded images (bit-exact)."""
import math
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bicubic_filter(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size, support0=2.0, filt=bicubic_filter):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the full-image box.  Returns (ksize, bounds[out,2], kk[out,ksize] int32)."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ww = 0.0
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [0.0] * ksize
        for x in range(xmax):
            w = filt((x + xmin - center + 0.5) * ss)
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        bounds[xx] = (xmin, xmax)
        for x in range(ksize):
            v = k[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if k[x] < 0 else int(0.5 + v)
    return ksize, bounds, kk


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resample_axis(img, out_size, axis):
    """One 8bpc pass along `axis` (1 = horizontal, 0 = vertical) of an [H, W, C] uint8 array."""
    in_size = img.shape[axis]
    _, bounds, kk = precompute_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, xmax = bounds[xx]
        acc = np.tensordot(kk[xx, :xmax].astype(np.int64), src[xmin:xmin + xmax], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[xx] = _clip8(acc)
    return np.moveaxis(out, 0, axis)


def resize_bicubic(img, out_w, out_h):
    """PIL Image.resize((out_w, out_h)) for an RGB uint8 array [H, W, 3]: horizontal pass, then vertical; a pass whose size
    does not change is skipped (Resample.c ImagingResampleInner need_horizontal / need_vertical)."""
    h, w = img.shape[:2]
    if w != out_w:
        img = resample_axis(img, out_w, 1)
    if h != out_h:
        img = resample_axis(img, out_h, 0)
    return img


def normalize_lut(mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """CLIPImageProcessor rescale + normalize for every (channel, byte) pair, with transformers' rounding order:
    float32(float64(u) * (1/255)) then (x - mean32) / std32 in float32 (image_transforms.py rescale / normalize)."""
    u = np.arange(256, dtype=np.float64)
    x = (u * (1 / 255)).astype(np.float32)
    m = np.asarray(mean, np.float32)[:, None]
    s = np.asarray(std, np.float32)[:, None]
    return ((x[None, :] - m) / s).astype(np.float32)


def anyres_tiles(img, best, tile=448, lut=None):
    """process_anyres_image (mm_utils.py:119-158) on an RGB uint8 array: returns fp32 [1 + n, 3, tile, tile] (thumbnail
    first, then row-major tiles of the aspect-preserving resize pasted centred on a black canvas)."""
    from omchat_amd.mm_utils import padded_size
    lut = normalize_lut() if lut is None else lut
    h, w = img.shape[:2]
    tw, th = best
    nw, nh = padded_size((w, h), best)
    canvas = np.zeros((th, tw, 3), np.uint8)
    x0, y0 = (tw - nw) // 2, (th - nh) // 2
    canvas[y0:y0 + nh, x0:x0 + nw] = resize_bicubic(img, nw, nh)
    tiles = [resize_bicubic(img, tile, tile)]
    for i in range(0, th, tile):
        for j in range(0, tw, tile):
            tiles.append(canvas[i:i + tile, j:j + tile])
    out = np.empty((len(tiles), 3, tile, tile), np.float32)
    for t, a in enumerate(tiles):
        for c in range(3):
            out[t, c] = lut[c][a[:, :, c]]
    return out


def dynamic_tiles(img, grid, tile=448, thumbnail=True, lut=None):
    """dynamic_preprocess + process_dynamic_image (mm_utils.py:276-323) on an RGB uint8 array: plain resize to the grid
    (cols, rows) x tile, row-major tiles, thumbnail first when the grid has more than one block."""
    lut = normalize_lut() if lut is None else lut
    gw, gh = grid
    r = resize_bicubic(img, gw * tile, gh * tile)
    tiles = [r[(i // gw) * tile:(i // gw + 1) * tile, (i % gw) * tile:(i % gw + 1) * tile] for i in range(gw * gh)]
    if thumbnail and len(tiles) != 1:
        tiles.insert(0, resize_bicubic(img, tile, tile))
    out = np.empty((len(tiles), 3, tile, tile), np.float32)
    for t, a in enumerate(tiles):
        for c in range(3):
            out[t, c] = lut[c][a[:, :, c]]
    return out


# ---- the reference's own PIL pipelines, restated with PIL (the pin for the numpy functions above and the CPU baseline of bench.py)
def pil_resize_and_pad(image, target_resolution):
    """resize_and_pad_image (mm_utils.py:42-74): aspect-preserving Image.resize, pasted centred on a black canvas."""
    from PIL import Image
    from omchat_amd.mm_utils import padded_size
    tw, th = target_resolution
    nw, nh = padded_size(image.size, target_resolution)
    canvas = Image.new("RGB", (tw, th), (0, 0, 0))
    canvas.paste(image.resize((nw, nh)), ((tw - nw) // 2, (th - nh) // 2))
    return canvas


def pil_process_anyres_image(image, processor, grid_pinpoints, return_best_res=False):
    """process_anyres_image (mm_utils.py:119-158) with a CPU `processor` (the reference's CLIPImageProcessor): thumbnail + tiles."""
    import torch
    from omchat_amd.mm_utils import select_best_resolution
    best = select_best_resolution(image.size, grid_pinpoints)
    padded = pil_resize_and_pad(image, best)
    edge = processor.crop_size["height"]
    w, h = padded.size
    patches = [padded.crop((j, i, j + edge, i + edge)) for i in range(0, h, edge) for j in range(0, w, edge)]      # divide_to_patches :77-96
    thumb = image.resize((edge, edge))
    tiles = torch.stack([processor.preprocess(p, return_tensors="pt")["pixel_values"][0] for p in [thumb] + patches], dim=0)
    return (tiles, best) if return_best_res else tiles


def pil_dynamic_preprocess(image, min_num=1, max_num=6, image_size=448, use_thumbnail=False):
    """dynamic_preprocess (mm_utils.py:276-312): plain resize to the grid, row-major crops, thumbnail first when > 1 block."""
    from omchat_amd.mm_utils import dynamic_grid
    gw, gh = dynamic_grid(image.size, min_num, max_num, image_size)
    resized = image.resize((image_size * gw, image_size * gh))
    out = [resized.crop(((i % gw) * image_size, (i // gw) * image_size, (i % gw + 1) * image_size, (i // gw + 1) * image_size))
           for i in range(gw * gh)]
    if use_thumbnail and len(out) != 1:
        out.insert(0, image.resize((image_size, image_size)))
    return out


def pil_process_dynamic_image(image, processor, max_num=6, image_size=448):
    """process_dynamic_image (mm_utils.py:315-323) with a CPU processor."""
    import torch
    return torch.stack([processor.preprocess(p, return_tensors="pt")["pixel_values"][0]
                        for p in pil_dynamic_preprocess(image, max_num=max_num, image_size=image_size, use_thumbnail=True)], dim=0)
