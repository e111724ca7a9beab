"""Image front-end, CPU side: the numpy restatement of Pillow's 8-bit BICUBIC resampler is pinned
against Pillow itself (the reference's dependency; Image.resize is what omchat/mm_utils.py:42-74,144 call), and the
host pieces of libomchat_hip.so (fixed-point taps, normalisation table, resolution plan) are compared with it.  No GPU."""
import ctypes as C
import numpy as np
import pytest

from oracle import preproc as opp
from omchat_amd import _lib
from omchat_amd.mm_utils import select_best_resolution

PINS = [[448, 896], [896, 448], [896, 896], [1344, 448], [448, 1344], [1344, 1344]]
SIZES = [(300, 500, 448, 448), (1000, 1500, 448, 896), (448, 448, 448, 448), (97, 61, 448, 448), (1344, 700, 231, 448),
         (500, 300, 600, 301), (448, 900, 448, 448), (2, 3, 5, 7)]


@pytest.mark.parametrize("h,w,oh,ow", SIZES)
def test_oracle_resize_is_pillow(h, w, oh, ow):
    from PIL import Image
    a = np.random.default_rng(h * 7 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    assert np.array_equal(opp.resize_bicubic(a, ow, oh), np.asarray(Image.fromarray(a).resize((ow, oh))))


def test_oracle_resize_extremes_are_pillow():
    from PIL import Image
    for a in (np.zeros((40, 30, 3), np.uint8), np.full((40, 30, 3), 255, np.uint8),
              np.tile(np.array([[0, 255]], np.uint8).repeat(3).reshape(1, 2, 3), (33, 17, 1))):      # overshoot -> clip8
        assert np.array_equal(opp.resize_bicubic(a, 77, 19), np.asarray(Image.fromarray(a).resize((77, 19))))


@pytest.mark.parametrize("n_in,n_out", [(500, 448), (61, 448), (1500, 896), (448, 448), (3000, 448), (7, 3), (3, 7)])
def test_library_taps_match_oracle(n_in, n_out):
    lib = _lib.lib()
    ks, bounds, kk = opp.precompute_coeffs(n_in, n_out)
    k = C.c_int(0)
    b = np.zeros((n_out, 2), np.int32)
    t = np.zeros((n_out, ks), np.int32)
    _lib.check(lib.omchat_resample_coeffs(n_in, n_out, C.byref(k), b.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), t.size))
    assert k.value == ks and np.array_equal(b, bounds) and np.array_equal(t, kk)
    with pytest.raises(ValueError):
        _lib.check(lib.omchat_resample_coeffs(n_in, n_out, C.byref(k), None, t.ctypes.data_as(C.c_void_p), 1))


def test_normalize_table_matches_transformers():
    from transformers import CLIPImageProcessor
    lib = _lib.lib()
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    lut = np.zeros((3, 256), np.float32)
    _lib.check(lib.omchat_normalize_lut((C.c_float * 3)(*mean), (C.c_float * 3)(*std), lut.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(lut, opp.normalize_lut(mean, std))
    # the reference's processor (internVIT_encoder.py:25-29) on an image holding every byte value in every channel
    proc = CLIPImageProcessor(crop_size=448, do_center_crop=True, do_normalize=True, do_resize=True, image_mean=mean, image_std=std, size=448)
    img = np.zeros((448, 448, 3), np.uint8)
    img[:256, 0, :] = np.arange(256, dtype=np.uint8)[:, None]
    from PIL import Image
    pv = proc.preprocess(Image.fromarray(img), return_tensors="np")["pixel_values"][0]
    assert np.array_equal(pv[:, :256, 0], lut)


def test_plan_matches_select_best_resolution():
    lib = _lib.lib()
    pins = np.asarray(PINS, np.int32)
    rng = np.random.default_rng(3)
    sizes = [(448, 448), (570, 380), (1000, 667), (1344, 448), (1, 1), (5000, 300), (300, 5000)] + [tuple(int(x) for x in rng.integers(1, 4000, 2)) for _ in range(200)]
    for w, h in sizes:
        bw, bh, n = C.c_int(0), C.c_int(0), C.c_int(0)
        _lib.check(lib.omchat_preproc_plan(w, h, pins.ctypes.data_as(C.c_void_p), len(pins), 448, C.byref(bw), C.byref(bh), C.byref(n)))
        best = select_best_resolution((w, h), [tuple(p) for p in PINS])
        assert (bw.value, bh.value) == tuple(best) and n.value == 1 + (best[0] // 448) * (best[1] // 448), (w, h)
    with pytest.raises(ValueError):
        _lib.check(lib.omchat_preproc_plan(10, 10, np.asarray([[100, 100]], np.int32).ctypes.data_as(C.c_void_p), 1, 448, C.byref(bw), C.byref(bh), C.byref(n)))


@pytest.mark.parametrize("w,h", [(570, 380), (333, 999), (448, 448)])
def test_oracle_anyres_is_the_reference_pipeline(w, h):
    """oracle.anyres_tiles (numpy) == the reference recipe run with PIL + the reference's CLIPImageProcessor (resize / paste / crop + transformers)"""
    from PIL import Image
    from transformers import CLIPImageProcessor
    proc = CLIPImageProcessor(crop_size=448, do_center_crop=True, do_normalize=True, do_resize=True,
                              image_mean=[0.485, 0.456, 0.406], image_std=[0.229, 0.224, 0.225], size=448)
    a = np.random.default_rng(w + h).integers(0, 256, (h, w, 3), dtype=np.uint8)
    tiles, best = opp.pil_process_anyres_image(Image.fromarray(a), proc, [tuple(p) for p in PINS], return_best_res=True)
    got = opp.anyres_tiles(a, best)
    assert got.shape == tuple(tiles.shape) and np.array_equal(got, tiles.numpy())


def test_dynamic_grid_and_oracle_match_the_reference_recipe():
    """dynamic_preprocess (mm_utils.py:276-312) restated: grid choice on a sweep of sizes, and oracle.dynamic_tiles == the PIL
    pipeline (resize -> crop -> CLIPImageProcessor) with the thumbnail first"""
    from PIL import Image
    from transformers import CLIPImageProcessor
    from omchat_amd.mm_utils import dynamic_grid
    assert [dynamic_grid(s, max_num=6) for s in [(448, 448), (900, 448), (448, 1400), (1000, 700), (3000, 1000), (100, 100)]] == \
        [(1, 1), (2, 1), (1, 3), (3, 2), (3, 1), (1, 1)]
    # expected grids captured from the reference's find_closest_aspect_ratio (imported in the build container); the tie rule
    # (:336-338) moves large pictures to the larger grid of the same aspect
    assert dynamic_grid((1000, 1000), max_num=6) == (2, 2) and dynamic_grid((1000, 1000), max_num=12) == (3, 3)
    assert dynamic_grid((2000, 1000), max_num=6) == (2, 1) and dynamic_grid((2000, 1000), max_num=12) == (4, 2)
    proc = CLIPImageProcessor(crop_size=448, do_center_crop=True, do_normalize=True, do_resize=True,
                              image_mean=[0.485, 0.456, 0.406], image_std=[0.229, 0.224, 0.225], size=448)
    for w, h in [(700, 400), (448, 448), (300, 900)]:
        a = np.random.default_rng(w + 3 * h).integers(0, 256, (h, w, 3), dtype=np.uint8)
        img = Image.fromarray(a)
        grid = dynamic_grid((w, h), max_num=6, image_size=448)
        pil = opp.pil_dynamic_preprocess(img, max_num=6, image_size=448, use_thumbnail=True)
        assert len(pil) == grid[0] * grid[1] + (grid[0] * grid[1] != 1) and all(p.size == (448, 448) for p in pil)
        tiles = opp.pil_process_dynamic_image(img, proc, max_num=6, image_size=448)
        assert np.array_equal(opp.dynamic_tiles(a, grid), tiles.numpy())


def _hf_fixture():
    import json, os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "hf_image_processor.json")))


def test_oracle_matches_the_reference_hf_image_processor():
    """tests/golden/hf_image_processor.json holds sha256 of the pixel bytes produced by the reference's OmChatImageProcessor
    (tools/make_golden_hfproc.py); the oracle reproduces them with the pinpoints read as (height, width) pairs"""
    import hashlib
    fx = _hf_fixture()
    pins_wh = [(p[1], p[0]) for p in fx["pinpoints"]]
    for c in fx["cases"]:
        a = np.random.default_rng(c["seed"]).integers(0, 256, (c["h"], c["w"], 3), dtype=np.uint8)
        best = select_best_resolution((c["w"], c["h"]), pins_wh)
        t = opp.anyres_tiles(a, best)
        assert t.shape[0] == c["n"] and hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest() == c["sha256"], c
    sq = next(c for c in fx["cases"] if c["w"] == c["h"] == 448)       # the tie case that separates the HF and the mm_utils conventions
    assert select_best_resolution((448, 448), pins_wh) == (896, 448) and select_best_resolution((448, 448), [tuple(p) for p in fx["pinpoints"]]) == (448, 896)


def test_hf_processor_prompt_layout_matches_reference():
    """OmChatProcessor.__call__ (hf/processing_omchat.py:171-253): single- and multi-image prompt -> ids, against ids captured from
    the reference with the same stub tokenizer; the pixel side is faked here (no GPU): only the tile counts matter for the layout"""
    import types
    import torch
    from omchat_amd.processing import OmChatProcessor
    fx = _hf_fixture()

    class Tok:
        bos_token_id = None
        pad_token_id = 0
        def __call__(self, s):
            return types.SimpleNamespace(input_ids=[1000 + ord(ch) for ch in s])
        def encode(self, s):
            return [1000 + ord(ch) for ch in s]

    counts = {c["seed"]: c["n"] for c in fx["cases"]}

    class FakeIP:
        def __call__(self, images, return_tensors="pt"):
            imgs = images if isinstance(images, list) else [images]
            n = torch.tensor([counts[i] for i in imgs])
            return {"pixel_values": torch.zeros(len(imgs), int(n.max()), 3, 2, 2), "num_patches": n}
    proc = OmChatProcessor(FakeIP(), Tok())
    for pr in fx["prompts"]:
        images = 0 if pr["n_images"] == 1 else [0, 1]                 # stand-ins keyed by the fixture's seeds
        out = proc(text=pr["text"], images=images)
        assert out["input_ids"][0].tolist() == pr["input_ids"]
        assert out["images"].shape[0] == pr["images_shape"][0] == pr["input_ids"].count(-200)
    assert proc(text="hello <image>")["input_ids"].shape[0] == 1 and "images" not in proc(text="hello")
