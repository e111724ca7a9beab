"""Device image front-end: the HIP counterpart of CLIPImageProcessor(size=448, crop=448, ImageNet mean/std)
(internVIT_encoder.py:25-29) + process_anyres_image (omchat/mm_utils.py:119-158).  Results are bit-identical to the
reference's PIL + transformers pipeline (tests/test_gpu_preproc.py); the arithmetic runs in libomchat_hip.so
(csrc/preproc.hip), there is no CPU fallback."""
import ctypes as C
import numpy as np

from . import _lib
from ._lib import check, ptr, cur_stream


def _rgb_array(image):
    """PIL.Image / ndarray / tensor -> (array-like uint8 [H, W, 3], on_device)."""
    import torch
    if isinstance(image, torch.Tensor):
        if image.dtype != torch.uint8 or image.dim() != 3 or image.shape[2] != 3:
            raise ValueError(f"expected a uint8 [H, W, 3] image, got {image.dtype} {tuple(image.shape)}")
        return image.contiguous(), image.is_cuda
    if hasattr(image, "convert"):
        image = np.asarray(image.convert("RGB"))
    a = np.ascontiguousarray(image)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError(f"expected a uint8 [H, W, 3] image, got {a.dtype} {a.shape}")
    return a, False


class HipImageProcessor:
    """Attribute-compatible with the CLIPImageProcessor the reference builds (crop_size, size, image_mean, image_std,
    preprocess(...)["pixel_values"]) so that `process_anyres_image(image, processor, grid_pinpoints)` call sites are unchanged."""

    def __init__(self, crop_size=448, image_mean=(0.485, 0.456, 0.406), image_std=(0.229, 0.224, 0.225), device=None):
        self.crop_size = {"height": crop_size, "width": crop_size}
        self.size = {"shortest_edge": crop_size}
        self.image_mean, self.image_std = list(image_mean), list(image_std)
        self.do_resize = self.do_center_crop = self.do_normalize = self.do_rescale = True
        self.device = device

    # ------------------------------------------------------------------ plan (host integers)
    def plan(self, size, grid_pinpoints):
        """select_best_resolution + tile count: ((best_w, best_h), n_tiles)."""
        lib = _lib.lib()
        pins = np.asarray(grid_pinpoints, np.int32).reshape(-1, 2)
        bw, bh, n = C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib.omchat_preproc_plan(int(size[0]), int(size[1]), pins.ctypes.data_as(C.c_void_p), len(pins), self.crop_size["height"],
                                      C.byref(bw), C.byref(bh), C.byref(n)))
        return (bw.value, bh.value), n.value

    # ------------------------------------------------------------------ anyres
    def process_anyres(self, image, grid_pinpoints, dtype=None, return_best_res=False):
        """[1 + n, 3, tile, tile] CUDA tensor (thumbnail first); dtype torch.float32 (reference behaviour) / float16 / bfloat16."""
        import torch
        if not torch.cuda.is_available():
            raise _lib.OmchatError("HipImageProcessor needs a HIP device (no CPU fallback)")
        dtype = dtype or torch.float32
        a, on_dev = _rgb_array(image)
        H, W = int(a.shape[0]), int(a.shape[1])
        best, n = self.plan((W, H), grid_pinpoints)
        tile = self.crop_size["height"]
        dev = torch.device(self.device if self.device is not None else (a.device if on_dev else f"cuda:{torch.cuda.current_device()}"))
        out = torch.empty(n, 3, tile, tile, dtype=dtype, device=dev)
        mean = (C.c_float * 3)(*self.image_mean)
        std = (C.c_float * 3)(*self.image_std)
        src = ptr(a) if isinstance(a, torch.Tensor) else a.ctypes.data_as(C.c_void_p)
        with torch.cuda.device(dev):
            check(_lib.lib().omchat_preproc_anyres(_lib.dtype_code(dtype), src, int(on_dev), W, H, best[0], best[1], tile, mean, std, ptr(out), cur_stream()))
        return (out, best) if return_best_res else out

    # ------------------------------------------------------------------ dynamic tiling (OmChat-2.1)
    def process_dynamic(self, image, max_num=6, image_size=None, min_num=1, use_thumbnail=True, dtype=None):
        """dynamic_preprocess + per-tile preprocess (mm_utils.py:276-323) on the device: [thumbnail? + cols*rows, 3, tile, tile]."""
        import torch
        from .mm_utils import dynamic_grid
        if not torch.cuda.is_available():
            raise _lib.OmchatError("HipImageProcessor needs a HIP device (no CPU fallback)")
        tile = self.crop_size["height"]
        if image_size is not None and image_size != tile:
            raise ValueError(f"image_size {image_size} differs from the processor's tile edge {tile}")
        dtype = dtype or torch.float32
        a, on_dev = _rgb_array(image)
        H, W = int(a.shape[0]), int(a.shape[1])
        gw, gh = dynamic_grid((W, H), min_num, max_num, tile)
        thumb = 1 if (use_thumbnail and gw * gh != 1) else 0
        dev = torch.device(self.device if self.device is not None else (a.device if on_dev else f"cuda:{torch.cuda.current_device()}"))
        out = torch.empty(thumb + gw * gh, 3, tile, tile, dtype=dtype, device=dev)
        mean = (C.c_float * 3)(*self.image_mean)
        std = (C.c_float * 3)(*self.image_std)
        src = ptr(a) if isinstance(a, torch.Tensor) else a.ctypes.data_as(C.c_void_p)
        with torch.cuda.device(dev):
            check(_lib.lib().omchat_preproc_dynamic(_lib.dtype_code(dtype), src, int(on_dev), W, H, gw, gh, tile, thumb, mean, std, ptr(out), cur_stream()))
        return out

    # ------------------------------------------------------------------ CLIPImageProcessor.preprocess for one tile-sized image
    def preprocess(self, images, return_tensors="pt", **kw):
        """One image (or a list) of exactly crop_size x crop_size: rescale + normalize + CHW (resize / centre-crop are
        identities there, which is the only way the reference calls it: mm_utils.py:146-147)."""
        import torch
        imgs = images if isinstance(images, (list, tuple)) else [images]
        tile = self.crop_size["height"]
        outs = []
        for im in imgs:
            a, _ = _rgb_array(im)
            if a.shape[0] != tile or a.shape[1] != tile:
                raise NotImplementedError(f"HipImageProcessor.preprocess takes {tile}x{tile} tiles (got {a.shape[1]}x{a.shape[0]}); "
                                          "use process_anyres for whole pictures")
            outs.append(self.process_anyres(a, [(tile, tile)])[0])          # best = (tile, tile): tile 0 (thumbnail) is the image itself
        pv = torch.stack(outs, 0)
        return {"pixel_values": pv if return_tensors == "pt" else pv.cpu().numpy()}
