"""Weight-only fp8 (OCP e4m3) decode path (SURVEY.md 8 f-2 / BASELINE configs[4]).  The reference has no quantised path, so
parity is: (1) the quantiser against torch.float8_e4m3fn on the CPU, bit-exact; (2) the fp8 GEMV and the whole decode
step against the oracle run on the DE-QUANTISED weights, at the usual 16-bit tolerances."""
import ctypes as C
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
from gpu_util import DT, CODE, TOL, TOL_DEEP, dev, rnd, rel, sync, ptr, randn
from omchat_amd import synth, _lib
from omchat_amd.config import tiny, omchat13b
from omchat_amd.engine import Engine
import oracle

DTS = ["bf16", "f16"]
T32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()


def quant_ref(w):
    """per-row absmax / 448 scale, e4m3 round-to-nearest-even of w / scale (torch CPU)"""
    w = w.float()
    m = w.abs().amax(dim=1)
    s = torch.where(m > 0, m / 448.0, torch.ones_like(m))
    q = (w / s[:, None]).to(torch.float8_e4m3fn)
    return q, s


def dequant_ref(w):
    q, s = quant_ref(w)
    return q.float() * s[:, None]


def dev_quant(w_dev, dt):
    lib = _lib.lib()
    N, K = w_dev.shape
    w8 = torch.empty(N, K, dtype=torch.uint8, device="cuda")
    sc = torch.empty(N, dtype=torch.float32, device="cuda")
    _lib.check(lib.omchat_op_quant_fp8(CODE[dt], ptr(w_dev), N, K, ptr(w8), ptr(sc), None))
    sync()
    return w8, sc


@pytest.mark.parametrize("dt", DTS)
def test_quantiser_is_e4m3_rne_bit_exact(gpu_lib, dt):
    w = rnd(randn((96, 1024), 0, 0.05), dt)
    w[3] = 0                                             # zero row -> scale 1, bytes 0
    w[5, 7] = 3.0                                        # a dominant element -> exactly 448 -> 0x7e
    w[6] = rnd(torch.linspace(-1, 1, 1024), dt)          # dense coverage of the e4m3 grid incl. subnormals
    w8, sc = dev_quant(dev(w, dt), dt)
    q, s = quant_ref(w)
    assert torch.equal(sc.cpu(), s)
    assert torch.equal(w8.cpu(), q.view(torch.uint8))
    assert int(w8[5, 7]) == 0x7e and int(w8[3].max()) == 0


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("N,K,epi,ks", [(4608, 3584, "none", 1), (3584, 3584, "partial", 3), (3584, 18944, "partial", 8),
                                         (2048, 3584, "swiglu", 1), (3584, 2048, "resid", 1), (8192, 3584, "f32", 1)])
def test_gemv_fp8_vs_dequantised_reference(gpu_lib, dt, N, K, epi, ks):
    lib = _lib.lib()
    w = rnd(randn((N, K), 1, 0.02), dt)
    x = rnd(randn((K,), 2, 0.5), dt)
    w8, sc = dev_quant(dev(w, dt), dt)
    wd = dequant_ref(w).double()
    acc = wd @ x.double()
    xd = dev(x, dt)
    if epi == "partial":
        y = torch.empty(ks, N, dtype=torch.float32, device="cuda")
        _lib.check(lib.omchat_op_gemv_fp8(CODE[dt], ptr(xd), ptr(w8), ptr(sc), ptr(y), N, K, None, None, 5, 0, ks, None)); sync()
        assert rel(y.sum(0), acc) < 2e-5
    elif epi == "f32":
        bias = rnd(randn((N,), 3, 0.1), dt)
        y = torch.empty(N, dtype=torch.float32, device="cuda")
        bd = dev(bias, dt)
        _lib.check(lib.omchat_op_gemv_fp8(CODE[dt], ptr(xd), ptr(w8), ptr(sc), ptr(y), N, K, ptr(bd), None, 0, 1, 1, None)); sync()
        assert rel(y, acc + bias.double()) < 2e-5
    elif epi == "none":
        bias = rnd(randn((N,), 3, 0.1), dt)
        bd = dev(bias, dt)
        y = torch.empty(N, dtype=DT[dt], device="cuda")
        _lib.check(lib.omchat_op_gemv_fp8(CODE[dt], ptr(xd), ptr(w8), ptr(sc), ptr(y), N, K, ptr(bd), None, 0, 0, 1, None)); sync()
        assert rel(y, acc + bias.double()) < TOL[dt]
    elif epi == "resid":
        r = rnd(randn((N,), 4, 1.0), dt)
        rd = dev(r, dt)
        y = torch.empty(N, dtype=DT[dt], device="cuda")
        _lib.check(lib.omchat_op_gemv_fp8(CODE[dt], ptr(xd), ptr(w8), ptr(sc), ptr(y), N, K, None, ptr(rd), 3, 0, 1, None)); sync()
        assert rel(y, r.double() + rnd(acc.float(), dt).double()) < TOL[dt]
    else:   # swiglu: rows interleaved in 16-row blocks [gate 16 | up 16]
        y = torch.empty(N // 2, dtype=DT[dt], device="cuda")
        _lib.check(lib.omchat_op_gemv_fp8(CODE[dt], ptr(xd), ptr(w8), ptr(sc), ptr(y), N, K, None, None, 4, 0, 1, None)); sync()
        a = acc.view(N // 32, 2, 16)
        g, u = rnd(a[:, 0].reshape(-1).float(), dt), rnd(a[:, 1].reshape(-1).float(), dt)
        ref = rnd(torch.nn.functional.silu(g), dt) * u
        assert rel(y, ref) < TOL[dt]


def test_fp8_gemv_rejects_batches(gpu_lib):
    lib = _lib.lib()
    # the op-level entry is batch 1 by construction; the context-level switch needs loaded weights
    e = Engine(tiny(), dtype="bf16", max_seq=32, vision=False)
    with pytest.raises(ValueError):
        e.enable_fp8_decode()
    e.close()


def _dequant_decoder_weights(sd, dt):
    out = dict(sd)
    for k, v in sd.items():
        if (".self_attn." in k or ".mlp." in k or k == "lm_head.weight") and k.endswith("weight") and "layernorm" not in k:
            out[k] = dequant_ref(rnd(v, dt))
    return out


@pytest.mark.parametrize("dt", DTS)
def test_decode_with_fp8_weights_vs_oracle_on_dequantised_weights(gpu_lib, dt):
    cfg = tiny()
    e = Engine(cfg, dtype=dt, max_seq=64, max_batch=1, vision=False)
    sd = {k: T32(v) for k, v in synth.state_dict(cfg, 7).items() if not k.startswith(synth.TOWER) and "mm_projector" not in k}
    e.load_state_dict(sd)
    x = rnd(randn((1, 24, 256), 5, 0.5), dt)
    logits, _ = e.prefill(x); sync()
    cache = oracle.KVCache(cfg.text["num_hidden_layers"])
    h = oracle.qwen2_model(x, sd, cfg.text, cache)                         # prefill: 16-bit weights on both sides
    assert rel(logits[0], oracle.lm_head(h, sd)[0, -1]) < TOL_DEEP[dt]
    sdq = _dequant_decoder_weights(sd, dt)
    base_nxt, base_lg = None, None
    e.enable_fp8_decode(True)
    for tok in (3, 11, 200):
        nxt, lg = e.decode_step(torch.tensor([tok]), want_logits=True); sync()
        ref = oracle.decode_step(torch.tensor([[tok]]), sdq, cfg.text, cache)[0, 0]
        assert rel(lg[0], ref) < TOL_DEEP[dt], rel(lg[0], ref)
        assert int(nxt[0]) == int(torch.argmax(lg[0]))
    # and it really is a different (quantised) computation from the 16-bit one, by about the e4m3 step
    e.enable_fp8_decode(False)
    cache2 = oracle.KVCache(cfg.text["num_hidden_layers"])
    oracle.qwen2_model(x, sd, cfg.text, cache2)
    e.prefill(x)
    _, lg16 = e.decode_step(torch.tensor([3]), want_logits=True)
    e.enable_fp8_decode(True)
    e.prefill(x)
    _, lg8 = e.decode_step(torch.tensor([3]), want_logits=True); sync()
    d = rel(lg8[0], lg16[0])
    assert 1e-3 < d < 0.2, d
    e.close()


@pytest.mark.parametrize("dt", ["bf16"])
def test_full_width_layer_fp8_decode(gpu_lib, dt):
    """one Qwen2-7B-width layer: the production launch shapes (K slices, SwiGLU interleave, fused qkv) with fp8 weights"""
    cfg = omchat13b()
    cfg.text["num_hidden_layers"] = 1
    cfg.text["vocab_size"] = 2048
    e = Engine(cfg, dtype=dt, max_seq=512, max_batch=1, vision=False)
    sd = {k: T32(v) for k, v in synth.state_dict(cfg, 0).items() if not k.startswith(synth.TOWER) and "mm_projector" not in k}
    e.load_state_dict(sd)
    x = rnd(randn((1, 100, 3584), 1, 0.5), dt)
    e.prefill(x); sync()
    cache = oracle.KVCache(1)
    oracle.qwen2_model(x, sd, cfg.text, cache)
    sdq = _dequant_decoder_weights(sd, dt)
    e.enable_fp8_decode(True)
    for tok in (5, 9):
        nxt, lg = e.decode_step(torch.tensor([tok]), want_logits=True); sync()
        ref = oracle.decode_step(torch.tensor([[tok]]), sdq, cfg.text, cache)[0, 0]
        assert rel(lg[0], ref) < TOL_DEEP[dt], rel(lg[0], ref)
    e.close()
