"""GPU parity cases added in round 2 (VERDICT r01 'close the parity holes'):
  * the InternViT-6B tower wrapper goldens (select_layer -1 / -2, patch / cls_patch) and the full-width 25-head attention golden,
    consumed through the C ABI (they were CPU-only fixtures before)
  * left-padded batch PREFILL against logits captured from the reference (omchat_arch.py:176-184, :206-207), decode refused
  * a free-running (not teacher-forced) fp16 greedy sequence that must reproduce every reference id
  * id range checks of the splice, shape check of the loader, fp8 replica refreshed after a weight reload"""
import types
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
from conftest import golden
from gpu_util import DT, CODE, TOL, TOL_DEEP, dev, rnd, rel, sync, ptr
from omchat_amd import synth, _lib
from omchat_amd.config import tiny
from omchat_amd.engine import Engine
import oracle

T32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
DTS = ["bf16", "f16"]


@pytest.fixture(scope="module")
def tiny_engines(gpu_lib):
    out = {}
    for dt in DTS:
        e = Engine(tiny(), dtype=dt, max_seq=256, max_batch=2, max_tiles=3)
        e.load_state_dict(synth.state_dict(tiny(), 0))
        out[dt] = e
    yield out
    for e in out.values():
        e.close()


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("sel,feat", [(-1, "patch"), (-2, "patch"), (-1, "cls_patch")])
def test_tower6b_wrapper_vs_reference_fp16_run(tiny_engines, dt, sel, feat):
    """InternVITVisionTower(...)(images) against the reference wrapper's own fp16 run (internVIT_encoder.py:35-56)"""
    from omchat_amd.model.vision_tower import build_vision_tower
    g = golden(f"tower_wrapper_L{sel}_{feat}")
    args = types.SimpleNamespace(mm_vision_tower="internvit-6b-448px", mm_vision_select_layer=sel, mm_vision_select_feature=feat)
    tw = build_vision_tower(args, engine=tiny_engines[dt])
    assert tw.select_layer == sel and tw.select_feature == feat and tw.is_loaded
    feats = tw(T32(g["pixels"]).half().cuda()); sync()
    assert feats.dtype == torch.float16 and tuple(feats.shape) == g["feats_half"].shape      # .to(images.dtype), :54
    assert rel(feats, T32(g["feats_half"])) < TOL_DEEP[dt], rel(feats, T32(g["feats_half"]))


@pytest.mark.parametrize("dt", DTS)
def test_vit_attention_full_width_vs_reference_golden(gpu_lib, dt):
    """InternAttention at the production width (3200 channels, 25 heads x 128, joint q/k RMSNorm over all heads) on 33 tokens:
    qkv GEMM -> vit_qknorm -> flash MHA -> proj GEMM through the op-level C ABI, against the module's output captured from the
    reference (modeling_intern_vit.py:138-155)"""
    g = golden("vit_attn_full")
    P = "g.attnfull."
    C_, H = 3200, 25
    w = {k: rnd(synth.uniform(P + k, s, 0, 0.05 if "norm" in k else 0.02, 1.0 if "norm" in k else 0.0), dt)
         for k, s in {"qkv.weight": (9600, 3200), "q_norm.weight": (3200,), "k_norm.weight": (3200,), "proj.weight": (3200, 3200),
                      "proj.bias": (3200,)}.items()}
    x = rnd(T32(g["x"]), dt)[0]                                        # [33, 3200]
    N = x.shape[0]
    code = CODE[dt]
    dx, dwq, dwp, dbp = dev(x, dt), dev(w["qkv.weight"], dt), dev(w["proj.weight"], dt), dev(w["proj.bias"], dt)
    dqn, dkn = dev(w["q_norm.weight"], dt), dev(w["k_norm.weight"], dt)
    qkv = torch.empty(N, 3 * C_, dtype=DT[dt], device="cuda")
    _lib.check(gpu_lib.omchat_op_gemm(code, ptr(dx), C_, ptr(dwq), C_, ptr(qkv), 3 * C_, N, 3 * C_, C_, None, None, None, 0, _lib.EPI_NONE, 0, None))
    _lib.check(gpu_lib.omchat_op_vit_qknorm(code, ptr(qkv), 3 * C_, ptr(dqn), ptr(dkn), N, C_, C_, 1e-6, 128 ** -0.5, None))
    ao = torch.empty(1, N, H, 128, dtype=DT[dt], device="cuda")
    _lib.check(gpu_lib.omchat_mha_fwd(ptr(qkv), 1, N, H, 1.0, 0, ptr(ao), code, None))       # q already carries head_dim^-0.5 (N5)
    out = torch.empty(N, C_, dtype=DT[dt], device="cuda")
    _lib.check(gpu_lib.omchat_op_gemm(code, ptr(ao), C_, ptr(dwp), C_, ptr(out), C_, N, C_, C_, ptr(dbp), None, None, 0, _lib.EPI_NONE, 0, None))
    sync()
    ref = T32(g["y"])[0]
    assert rel(out, ref) < 2 * TOL[dt], rel(out, ref)


@pytest.mark.parametrize("dt", DTS)
def test_left_padded_batch_prefill_vs_reference_logits(gpu_lib, dt):
    """tokenizer_padding_side='left': splice (bit-exact golden elsewhere) + prefill with RoPE on arange(S) and the padded keys masked;
    logits of position S - 1 of both rows against the reference's forward on the same batch; the right-padded run of the same
    batch against its own golden; decode after the left-padded prefill is refused"""
    g = golden("leftpad_prefill")
    cfg = tiny()
    e = Engine(cfg, dtype=dt, max_seq=128, max_batch=2, max_tiles=3)
    e.load_state_dict(synth.state_dict(cfg, int(g["seed"])))
    ids, mask = torch.from_numpy(g["ids"]).long(), torch.from_numpy(g["mask"]).long()
    feats = rnd(T32(g["feats"]), dt)
    lens_want = [int(x) for x in g["lengths"]]
    for side, key in (("left", "logits_left_last"), ("right", "logits_right_last")):
        embeds, lengths, valid = e.splice(ids, mask, feats.cuda(), padding_side=side)
        assert lengths == lens_want and embeds.shape[1] == int(g["S"])
        logits, _ = e.prefill(embeds, lengths, padding_side=side); sync()
        ref = T32(g[key])
        for i in range(2):
            assert rel(logits[i], ref[i]) < TOL_DEEP[dt], (side, i, rel(logits[i], ref[i]))
        if side == "left":
            with pytest.raises(ValueError, match="left-padded"):
                e.decode_step(torch.tensor([1, 2]))
        else:
            e.decode_step(torch.tensor([1, 2]))              # right-padded batches decode as before
    # the reference-shaped forward() picks the side from the attention mask
    from omchat_amd.model.omchat_qwen2 import OmChatQwen2ForCausalLM
    m = OmChatQwen2ForCausalLM(cfg.clone(), e)
    m.config.mm["tokenizer_padding_side"] = "left"
    m.encode_images = lambda images: feats.to(DT[dt]).cuda()
    out = m(input_ids=ids, attention_mask=mask, images=torch.zeros(3, 3, 56, 56))
    sync()
    assert rel(out.logits[:, 0], T32(g["logits_left_last"])) < TOL_DEEP[dt]
    e.close()


def test_free_running_fp16_sequence_equals_every_reference_id(gpu_lib):
    """no teacher forcing: the fp16 HIP path generates 16 tokens on its own and every id must equal the reference's fp16 run
    (margins >= 0.013, an order of magnitude above the fp16 noise of this model)"""
    g = golden("e2e_free_f16")
    cfg = tiny()
    e = Engine(cfg, dtype="f16", max_seq=128, max_batch=1, max_tiles=2)
    e.load_state_dict(synth.state_dict(cfg, int(g["seed"])))
    px = T32(synth.pixels(int(g["n_tiles"]), 56, int(g["pixel_seed"])))
    embeds, lengths, _ = e.splice(torch.from_numpy(g["ids"]).long(), None, e.encode_images(px))
    logits, _ = e.prefill(embeds, lengths)
    tok = e.argmax(logits)
    got = [int(tok[0])]
    ref = [int(t) for t in g["tokens"]]
    for _ in range(len(ref) - 1):
        tok, _ = e.decode_step(tok)
        got.append(int(tok[0]))
    assert float(g["margins"].min()) > 0.01 and len(set(ref)) >= 3
    assert got == ref, (got, ref)
    e.close()


def test_splice_and_loader_reject_bad_inputs(tiny_engines):
    e = tiny_engines["bf16"]
    feats = torch.zeros(1, 16, 256, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(IndexError):
        e.splice(torch.tensor([[3, -200, 320]]), None, feats)          # id == vocab
    with pytest.raises(IndexError):
        e.splice(torch.tensor([[3, -200, -100]]), None, feats)         # IGNORE_INDEX-like stray negative id
    e.splice(torch.tensor([[3, -200, -100]]), torch.tensor([[1, 1, 0]]), feats)      # masked out: dropped before the lookup
    w = torch.zeros(512, 256)                                           # fc1 is [512, 256]; a transposed [256, 512] must not load
    with pytest.raises(ValueError, match="shape mismatch"):
        e.load_tensor(synth.TOWER + "encoder.layers.0.mlp.fc1.weight", w.t().contiguous())


def test_fp8_replica_follows_a_weight_reload(gpu_lib):
    """after omchat_load_tensor the e4m3 replica is stale; enabling fp8 decode again must re-quantise it"""
    cfg = tiny()
    sd = synth.state_dict(cfg, 5)
    e = Engine(cfg, dtype="bf16", max_seq=64, max_batch=1, max_tiles=1, vision=False)
    e.load_state_dict({k: v for k, v in sd.items() if not k.startswith(synth.TOWER) and "mm_projector" not in k})
    x = torch.randn(1, 9, 256, generator=torch.Generator().manual_seed(0)) * 0.5

    def fp8_step():
        e.enable_fp8_decode(True)
        e.prefill(x)
        _, lg = e.decode_step(torch.tensor([7]), want_logits=True); sync()
        e.enable_fp8_decode(False)
        return lg.clone()
    a = fp8_step()
    name = "model.layers.1.mlp.down_proj.weight"
    e.load_tensor(name, torch.from_numpy(sd[name]) * 3.0)
    b = fp8_step()
    assert not torch.equal(a, b)                                       # a stale replica would reproduce `a` bit for bit
    e.prefill(x)
    _, ref = e.decode_step(torch.tensor([7]), want_logits=True); sync()
    assert rel(b, ref) < 0.12                                          # e4m3 weights vs the 16-bit weights of the SAME (new) values
    e.close()


# ---------------------------------------------------------------------------------------------------------------------
# batched decode GEMV with packed operands (gemv.hip: gemv_pk_kernel)
# ---------------------------------------------------------------------------------------------------------------------
def _unpack_x(packed, b, K, NB):
    """inverse of common.h packed_x_index on the host"""
    p = packed.cpu().view(K // 64, 2, NB, 4, 16, 8)           # [chunk][half][nb][g][row%16][j]
    x = p.permute(2, 4, 0, 1, 3, 5).reshape(NB * 16, K)       # row = nb*16 + r ; k = chunk*64 + half*32 + g*8 + j
    return x[:b]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("wp", [0, 1])
@pytest.mark.parametrize("b,N,K,ks", [(2, 320, 512, 1), (16, 4608, 3584, 1), (17, 320, 576, 2), (32, 3584, 18944, 8), (32, 37888, 3584, 1),
                                      (24, 160, 64, 1), (5, 3584, 3584, 3), (32, 32768 + 64, 256, 1),
                                      # x-stationary persistent forms (packed W, K = 3584 in one piece / 18944 in 7 x 40 + 16 chunks): ragged batches
                                      (17, 32768, 3584, 1), (3, 16384, 3584, 1), (20, 3584, 18944, 8)])
def test_gemv_packed_operands(gpu_lib, dt, wp, b, N, K, ks):
    from test_gpu_ops import _gemm_ref
    from gpu_util import randn
    X = rnd(randn((b, K), 1), dt); W = rnd(randn((N, K), 2, 0.03), dt); bias = rnd(randn((N,), 3, 0.1), dt)
    dX, dW, db = dev(X, dt), dev(W, dt), dev(bias, dt)
    y = X @ W.t()
    code = CODE[dt]
    NB = 2 if b > 16 else 1
    out = torch.full((b, N), float("nan"), dtype=DT[dt], device="cuda")
    _lib.check(gpu_lib.omchat_op_gemv_packed(code, ptr(dX), K, ptr(dW), K, ptr(out), N, b, N, K, ptr(db), _lib.EPI_NONE, 0, 1, wp, 0, None))
    sync(); assert rel(out, rnd(y + bias, dt)) < TOL[dt]
    outf = torch.full((b, N), float("nan"), dtype=torch.float32, device="cuda")
    _lib.check(gpu_lib.omchat_op_gemv_packed(code, ptr(dX), K, ptr(dW), K, ptr(outf), N, b, N, K, None, _lib.EPI_NONE, 1, 1, wp, 0, None))
    sync(); assert rel(outf, y) < 1e-4
    part = torch.full((ks, b, N), float("nan"), dtype=torch.float32, device="cuda")
    _lib.check(gpu_lib.omchat_op_gemv_packed(code, ptr(dX), K, ptr(dW), K, ptr(part), N, b, N, K, None, 5, 0, ks, wp, 0, None))
    sync(); assert torch.isfinite(part).all() and rel(part.sum(0), y) < 1e-4
    if N % 32 == 0 and (N // 2) % 64 == 0:
        ref = _gemm_ref(X, W, None, None, None, _lib.EPI_SWIGLU, dt)
        o3 = torch.full((b, N // 2), float("nan"), dtype=DT[dt], device="cuda")
        _lib.check(gpu_lib.omchat_op_gemv_packed(code, ptr(dX), K, ptr(dW), K, ptr(o3), N // 2, b, N, K, None, _lib.EPI_SWIGLU, 0, 1, wp, 0, None))
        sync(); assert rel(o3, ref) < TOL[dt]
        o4 = torch.zeros(NB * 16 * (N // 2), dtype=DT[dt], device="cuda")
        _lib.check(gpu_lib.omchat_op_gemv_packed(code, ptr(dX), K, ptr(dW), K, ptr(o4), 0, b, N, K, None, _lib.EPI_SWIGLU, 0, 1, wp, 1, None))
        sync(); assert torch.equal(_unpack_x(o4, b, N // 2, NB), o3.cpu())           # same values, packed for the next GEMV


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("replica", [1, 0])
def test_batched_decode_packed_path_vs_oracle(gpu_lib, dt, replica):
    """batched decode steps (b = 3 and b = 20) run on the packed path (packed activations; packed weight replica or the row-major
    weights): logits against the ORACLE (fp32 restatement on the same 16-bit-rounded inputs, two decode steps per checked row) and
    against single-sequence steps of the same model (b = 1: whole-row kernels on row-major activations)"""
    import oracle
    cfg = tiny(q_heads=4, kv_heads=2)
    sd = synth.state_dict(cfg, 3)
    sdt = {k: rnd(torch.from_numpy(v), dt) for k, v in sd.items() if not k.startswith(synth.TOWER) and "mm_projector" not in k}
    gpu_lib.omchat_op_set_tuning(6, replica)
    try:
        for b in (3, 20):
            e = Engine(cfg, dtype=dt, max_seq=64, max_batch=b, max_tiles=1, vision=False)
            e.load_state_dict({k: v for k, v in sd.items() if not k.startswith(synth.TOWER) and "mm_projector" not in k})
            S = 9
            x = rnd(torch.randn(b, S, 256, generator=torch.Generator().manual_seed(b)) * 0.5, dt)
            lens = [S - (i % 4) for i in range(b)]
            e.prefill(x, lens)
            toks = torch.arange(b) % 300 + 5
            nxt, lg = e.decode_step(toks, want_logits=True)
            nxt2, lg2 = e.decode_step(nxt, want_logits=True); sync()
            # against single-sequence decode on the same engine (b = 1: whole-row kernels, row-major activations)
            e1 = Engine(cfg, dtype=dt, max_seq=64, max_batch=1, max_tiles=1, vision=False)
            e1.load_state_dict({k: v for k, v in sd.items() if not k.startswith(synth.TOWER) and "mm_projector" not in k})
            for i in (0, 1, b - 1):
                e1.prefill(x[i:i + 1, :lens[i]].contiguous(), [lens[i]])
                _, l1 = e1.decode_step(toks[i:i + 1], want_logits=True)
                _, l2 = e1.decode_step(nxt[i:i + 1], want_logits=True); sync()
                assert rel(lg[i], l1[0]) < TOL_DEEP[dt] and rel(lg2[i], l2[0]) < TOL_DEEP[dt], (b, i, rel(lg[i], l1[0]), rel(lg2[i], l2[0]))
                # the oracle leg: prefill row i, then the same two tokens
                cache = oracle.KVCache(cfg.text["num_hidden_layers"])
                oracle.qwen2_model(x[i:i + 1, :lens[i]], sdt, cfg.text, cache)
                o1 = oracle.decode_step(toks[i:i + 1][None].long(), sdt, cfg.text, cache)[0, 0]
                o2 = oracle.decode_step(nxt[i:i + 1].cpu()[None].long(), sdt, cfg.text, cache)[0, 0]
                assert rel(lg[i], o1) < TOL_DEEP[dt] and rel(lg2[i], o2) < TOL_DEEP[dt], (b, i, rel(lg[i], o1), rel(lg2[i], o2))
            e.close(); e1.close()
    finally:
        gpu_lib.omchat_op_set_tuning(6, 1)


@pytest.mark.parametrize("dt", DTS)
def test_decode_attention_tiles_per_wave_in_the_model(gpu_lib, dt):
    """the multi-tile decode attention kernel (several 64-key tiles per wave, running max / sum, fused RoPE + KV append in the tile that
    owns the new position) inside the decode step: contexts that cross tile and split boundaries, ragged lengths, 4 steps; logits against
    the one-tile kernel (tuning key 10)"""
    cfg = tiny(q_heads=4, kv_heads=2)
    sd = {k: v for k, v in synth.state_dict(cfg, 5).items() if not k.startswith(synth.TOWER) and "mm_projector" not in k}
    b, S = 4, 200
    x = torch.randn(b, S, 256, generator=torch.Generator().manual_seed(11)) * 0.5
    lens = [200, 127, 129, 64]              # new positions 200.. (4th tile), 127 (last row of a tile), 129, 64 (first row of a tile)
    runs = {}
    try:
        for tpw in (1, 2, 4):
            gpu_lib.omchat_op_set_tuning(10, tpw)
            e = Engine(cfg, dtype=dt, max_seq=256, max_batch=b, max_tiles=1, vision=False)
            e.load_state_dict(sd)
            e.prefill(x, lens)
            tok = torch.arange(b) % 300 + 7
            out = []
            for _ in range(4):
                tok, lg = e.decode_step(tok, want_logits=True)
                out.append((tok.cpu().clone(), lg.float().cpu().clone()))
            sync()
            runs[tpw] = out
            e.close()
    finally:
        gpu_lib.omchat_op_set_tuning(10, 0)
    for tpw in (2, 4):
        for step in range(4):
            assert torch.isfinite(runs[tpw][step][1]).all()
            r = rel(runs[tpw][step][1], runs[1][step][1])
            assert r < (2e-2 if dt == "bf16" else 4e-3), (tpw, step, r)      # another split of the same sum: 16-bit rounding of P and of the partial merge
