"""Image front-end on the device (csrc/preproc.hip) against the oracle (itself pinned to Pillow + transformers in
tests/test_preproc_cpu.py): byte work -> bit-exact, including the dtype cast."""
import numpy as np
import pytest
import torch

from oracle import preproc as opp

pytestmark = pytest.mark.gpu
PINS = [(448, 896), (896, 448), (896, 896), (1344, 448), (448, 1344), (1344, 1344)]


def _proc():
    from omchat_amd.image_processing import HipImageProcessor
    return HipImageProcessor(crop_size=448)


@pytest.mark.parametrize("w,h", [(570, 380), (333, 999), (448, 448), (61, 97), (3000, 2000), (1344, 1344), (449, 447), (2000, 90)])
def test_anyres_matches_oracle_bit_exact(w, h):
    a = np.random.default_rng(w * 5 + h).integers(0, 256, (h, w, 3), dtype=np.uint8)
    proc = _proc()
    out, best = proc.process_anyres(a, PINS, return_best_res=True)
    ref = opp.anyres_tiles(a, best)
    assert out.dtype == torch.float32 and tuple(out.shape) == ref.shape
    assert np.array_equal(out.cpu().numpy(), ref)


def test_anyres_dtypes_and_device_input():
    a = np.random.default_rng(11).integers(0, 256, (380, 570, 3), dtype=np.uint8)
    proc = _proc()
    f32 = proc.process_anyres(a, PINS)
    for dt in (torch.float16, torch.bfloat16):
        got = proc.process_anyres(a, PINS, dtype=dt)
        assert got.dtype == dt and torch.equal(got, f32.to(dt))               # fused cast == the reference's .half() / .to(bf16)
    dev = proc.process_anyres(torch.from_numpy(a).cuda(), PINS)
    assert torch.equal(dev, f32)
    from PIL import Image
    assert torch.equal(proc.process_anyres(Image.fromarray(a), PINS), f32)


def test_extreme_images():
    proc = _proc()
    for a in (np.zeros((100, 300, 3), np.uint8), np.full((700, 200, 3), 255, np.uint8),
              np.tile(np.array([[0, 255]], np.uint8).repeat(3).reshape(1, 2, 3), (64, 256, 1))):
        out, best = proc.process_anyres(a, PINS, return_best_res=True)
        assert np.array_equal(out.cpu().numpy(), opp.anyres_tiles(np.ascontiguousarray(a), best))


def test_preprocess_tile_and_errors():
    proc = _proc()
    a = np.random.default_rng(5).integers(0, 256, (448, 448, 3), dtype=np.uint8)
    pv = proc.preprocess(a, return_tensors="pt")["pixel_values"]
    lut = opp.normalize_lut()
    ref = np.stack([lut[c][a[:, :, c]] for c in range(3)])
    assert tuple(pv.shape) == (1, 3, 448, 448) and np.array_equal(pv[0].cpu().numpy(), ref)
    with pytest.raises(NotImplementedError):
        proc.preprocess(np.zeros((10, 10, 3), np.uint8))
    with pytest.raises(ValueError):
        proc.process_anyres(np.zeros((10, 10), np.uint8), PINS)


def test_mm_utils_and_get_context_use_the_device_front_end():
    """process_anyres_image / get_context call sites (mm_utils.py:119-158, make_context.py:14-43) with the device processor"""
    from PIL import Image
    from omchat_amd.mm_utils import process_anyres_image
    from omchat_amd.make_context import get_context
    a = np.random.default_rng(2).integers(0, 256, (380, 570, 3), dtype=np.uint8)
    img = Image.fromarray(a)
    proc = _proc()
    tiles, best = process_anyres_image(img, proc, PINS, True, return_best_res=True)
    ref = opp.anyres_tiles(a, best)
    assert best == (896, 448) and len(tiles) == 3 and all(t.is_cuda for t in tiles)
    assert np.array_equal(torch.stack(tiles).cpu().numpy(), ref)

    import types

    class _Tok:
        bos_token_id = None
        pad_token_id = 0

        def __call__(self, s):
            return types.SimpleNamespace(input_ids=[1000 + ord(ch) for ch in s])

        def encode(self, s):
            return [1000 + ord(ch) for ch in s]
    _, ids, image_tensor = get_context("hi", _Tok(), image=img, image_processor=proc, image_grid_pinpoints=PINS)
    assert ids.count(-200) == 3 and image_tensor.dtype == torch.float16 and image_tensor.is_cuda and tuple(image_tensor.shape) == (3, 3, 448, 448)
    assert torch.equal(image_tensor.cpu(), torch.from_numpy(ref).half())      # make_context.py:25 `.half()`


@pytest.mark.parametrize("w,h", [(700, 400), (448, 448), (300, 900), (2000, 600)])
def test_dynamic_tiling_matches_oracle_bit_exact(w, h):
    """dynamic_preprocess + process_dynamic_image (mm_utils.py:276-323; OmChat-2.1 tiling) on the device"""
    from omchat_amd.mm_utils import dynamic_grid, process_dynamic_image
    a = np.random.default_rng(w + 7 * h).integers(0, 256, (h, w, 3), dtype=np.uint8)
    proc = _proc()
    out = proc.process_dynamic(a, max_num=6)
    grid = dynamic_grid((w, h), max_num=6, image_size=448)
    ref = opp.dynamic_tiles(a, grid)
    assert tuple(out.shape) == ref.shape and np.array_equal(out.cpu().numpy(), ref)
    from PIL import Image
    via = process_dynamic_image(Image.fromarray(a), proc, max_num=6, image_size=448)
    assert torch.equal(via, out)
    with pytest.raises(ValueError):
        proc.process_dynamic(a, image_size=336)


def test_hf_image_processor_and_processor_on_device():
    """omchat_amd.processing.OmChatImageProcessor / OmChatProcessor (hf_example.py flow) against the bytes and ids captured from the
    reference's HF classes (tests/golden/hf_image_processor.json)"""
    import hashlib, json, os, types
    from omchat_amd.processing import OmChatImageProcessor, OmChatProcessor
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "hf_image_processor.json")))
    ip = OmChatImageProcessor(image_grid_pinpoints=fx["pinpoints"])
    arrays = {}
    for c in fx["cases"]:
        a = np.random.default_rng(c["seed"]).integers(0, 256, (c["h"], c["w"], 3), dtype=np.uint8)
        arrays[c["seed"]] = a
        r = ip(a)
        n = int(r["num_patches"][0])
        pv = np.ascontiguousarray(r["pixel_values"][0, :n].cpu().numpy())
        assert n == c["n"] and pv.dtype == np.float32 and hashlib.sha256(pv.tobytes()).hexdigest() == c["sha256"], c

    class Tok:
        bos_token_id = None
        pad_token_id = 0
        def __call__(self, s):
            return types.SimpleNamespace(input_ids=[1000 + ord(ch) for ch in s])
        def encode(self, s):
            return [1000 + ord(ch) for ch in s]
    proc = OmChatProcessor(ip, Tok())
    for pr in fx["prompts"]:
        images = arrays[0] if pr["n_images"] == 1 else [arrays[0], arrays[1]]
        out = proc(text=pr["text"], images=images)
        assert out["input_ids"][0].tolist() == pr["input_ids"] and list(out["images"].shape) == pr["images_shape"] and out["images"].is_cuda
