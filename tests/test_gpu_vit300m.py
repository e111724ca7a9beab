"""InternViT-300M tower variant (intern_vit_300m/modeling_intern_vit.py): head_dim-64 flash attention,
LayerNorm, the tower against the golden vectors captured from the reference and against the oracle, the wrapper class,
and one full-width (1024 / 16 x 64 / 4096) layer."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
from conftest import golden
from gpu_util import DT, CODE, TOL, TOL_DEEP, dev, rnd, rel, sync, ptr, randn
from test_gpu_ops import _attn_ref
from omchat_amd import synth, _lib
from omchat_amd.config import tiny300m, omchat8b_21
from omchat_amd.engine import Engine
import oracle
from oracle.pipeline import _sub, TOWER_PFX

DTS = ["bf16", "f16"]
T32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("b,Sq,Skv,Hq,Hkv,causal,lens", [
    (2, 1025, 1025, 4, 4, 0, None),          # ViT-300M shape: ragged 1024+1 tail
    (1, 300, 300, 8, 2, 1, None),            # GQA + causal also work at D = 64
    (2, 200, 200, 4, 2, 1, [200, 77]),
    (1, 17, 17, 2, 2, 0, None),
    (1, 64, 64, 16, 16, 0, None),
])
def test_attn_prefill_head_dim_64(gpu_lib, dt, b, Sq, Skv, Hq, Hkv, causal, lens):
    D = 64
    q = rnd(randn((b, Sq, Hq, D), 1), dt); k = rnd(randn((b, Hkv, Skv, D), 2), dt); v = rnd(randn((b, Hkv, Skv, D), 3), dt)
    dq, dk, dv = dev(q, dt), dev(k, dt), dev(v, dt)
    out = torch.full((b, Sq, Hq, D), float("nan"), dtype=DT[dt], device="cuda")
    dl = None if lens is None else torch.tensor(lens, dtype=torch.int32, device="cuda")
    _lib.check(gpu_lib.omchat_op_attn_prefill_d(CODE[dt], ptr(dq), ptr(dk), ptr(dv), ptr(out), b, Sq, Skv, Hq, Hkv, D, ptr(dl), causal, 0, 0.125, None))
    sync()
    ref = _attn_ref(q, k, v, 0.125, causal, 0, lens or [Skv] * b)
    for i in range(b):
        n = Sq if lens is None else lens[i]
        assert torch.isfinite(out[i, :n].float()).all()
        assert rel(out[i, :n], ref[i, :n]) < TOL[dt], rel(out[i, :n], ref[i, :n])
    with pytest.raises(ValueError):
        _lib.check(gpu_lib.omchat_op_attn_prefill_d(CODE[dt], ptr(dq), ptr(dk), ptr(dv), ptr(out), b, Sq, Skv, Hq, Hkv, 96, ptr(dl), causal, 0, 0.1, None))


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("rows,H", [(5, 256), (1025, 1024), (3, 3200)])
def test_layernorm(gpu_lib, dt, rows, H):
    x = rnd(randn((rows, H), 1, 2.0) + 0.3, dt); w = rnd(randn((H,), 2, 0.05) + 1, dt); b = rnd(randn((H,), 3, 0.02), dt)
    y = torch.empty(rows, H, dtype=DT[dt], device="cuda")
    dx, dw, db = dev(x, dt), dev(w, dt), dev(b, dt)          # named: a temporary would be freed (and reused) before the launch
    _lib.check(gpu_lib.omchat_op_layernorm(CODE[dt], ptr(dx), ptr(dw), ptr(db), ptr(y), rows, H, 1e-6, None)); sync()
    ref = torch.nn.functional.layer_norm(x, (H,), w, b, 1e-6)
    assert rel(y, ref) < TOL[dt]
    half = torch.nn.functional.layer_norm(x.to(DT[dt]), (H,), w.to(DT[dt]), b.to(DT[dt]), 1e-6)      # ATen on the 16-bit tensors
    assert (y.cpu().float() - half.float()).abs().max() <= 2 * torch.finfo(DT[dt]).eps * ref.abs().max()


@pytest.fixture(scope="module")
def engines(gpu_lib):
    out = {}
    for dt in DTS:
        e = Engine(tiny300m(), dtype=dt, max_seq=128, max_batch=1, max_tiles=3)
        e.load_state_dict(synth.state_dict(tiny300m(), 0))
        out[dt] = e
    yield out
    for e in out.values():
        e.close()


@pytest.mark.parametrize("dt", DTS)
def test_vit300m_tiny_vs_golden_and_oracle(engines, dt):
    e = engines[dt]
    g = golden("vit300m_tiny")
    px = T32(g["pixels"])
    for idx, key in ((0, "hs0"), (1, "hs1"), (2, "hs2")):
        out = e.vit_forward(px, select_layer=idx, select_feature="cls_patch"); sync()
        assert out.shape == (2, 17, 256)
        assert rel(out, T32(g[key])) < TOL_DEEP[dt], (key, rel(out, T32(g[key])))
    cfg = tiny300m()
    w = _sub({k: T32(v) for k, v in synth.state_dict(cfg, 0, synth.TOWER).items()}, TOWER_PFX)
    ref = oracle.vision_tower_forward(px, w, cfg.vision, -1, "patch")
    got = e.vit_forward(px, select_layer=-1, select_feature="patch"); sync()
    assert got.shape == (2, 16, 256) and rel(got, ref) < TOL_DEEP[dt]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("sel,feat", [(-1, "patch"), (-2, "cls_patch")])
def test_tower300m_wrapper_vs_reference_fp16_run(engines, dt, sel, feat):
    """InternVIT300mVisionTower(...)(images) against the reference wrapper's own fp16 run (internVIT300m_encoder.py:45-56)"""
    import types
    from omchat_amd.model.vision_tower import build_vision_tower, InternVIT300mVisionTower
    g = golden(f"tower300m_wrapper_L{sel}_{feat}")
    args = types.SimpleNamespace(mm_vision_tower="internvit-300m-448px", mm_vision_select_layer=sel, mm_vision_select_feature=feat)
    tw = build_vision_tower(args, engine=engines[dt])
    assert isinstance(tw, InternVIT300mVisionTower) and tw.hidden_size == 256 and tw.num_patches == 16
    feats = tw(T32(g["pixels"]).half().cuda()); sync()
    assert feats.dtype == torch.float16 and tuple(feats.shape) == g["feats_half"].shape
    assert rel(feats, T32(g["feats_half"])) < TOL_DEEP[dt]
    from omchat_amd.config import tiny
    e6 = Engine(tiny(), dtype=dt, max_seq=32, text=False)
    with pytest.raises(ValueError):
        InternVIT300mVisionTower("internvit-300m-448px", args, engine=e6)
    e6.close()


@pytest.mark.parametrize("dt", DTS)
def test_encode_images_and_fill_synthetic_300m(gpu_lib, dt):
    """projector behind the 300M tower (mm_hidden 256 -> text hidden) and the device-side synthetic fill of the new tensors"""
    cfg = tiny300m()
    a = Engine(cfg, dtype=dt, max_seq=64, max_tiles=2); a.fill_synthetic(0)
    b = Engine(cfg, dtype=dt, max_seq=64, max_tiles=2); b.load_state_dict(synth.state_dict(cfg, 0))
    px = T32(synth.pixels(2, 56, 3))
    fa, fb = a.encode_images(px), b.encode_images(px); sync()
    assert torch.equal(fa, fb)
    sd = {k: T32(v) for k, v in synth.state_dict(cfg, 0).items()}
    ref = oracle.encode_images(px, sd, cfg.vision)
    assert rel(fa, torch.stack(list(ref))) < TOL_DEEP[dt]
    a.close(); b.close()


@pytest.mark.parametrize("dt", ["bf16"])
def test_full_width_300m_layers(gpu_lib, dt):
    """two full-width InternViT-300M layers (1024 hidden, 16 x 64 heads, 4096 MLP) on one 448 x 448 tile (1025 tokens)"""
    cfg = omchat8b_21()
    cfg.vision["num_hidden_layers"] = 2
    e = Engine(cfg, dtype=dt, max_seq=32, max_tiles=1, text=False)
    sd = synth.state_dict(cfg, 0, synth.TOWER)
    sd.update(synth.state_dict(cfg, 0, "model.mm_projector."))
    e.load_state_dict(sd)
    px = T32(synth.pixels(1, 448, 1))
    got = e.vit_forward(px, select_layer=-1, select_feature="cls_patch"); sync()
    w = _sub({k: T32(v) for k, v in sd.items() if k.startswith(synth.TOWER)}, TOWER_PFX)
    ref = oracle.vision_tower_forward(px, w, cfg.vision, -1, "cls_patch")
    assert got.shape == (1, 1025, 1024) and rel(got, ref) < TOL_DEEP[dt], rel(got, ref)
    e.close()
