"""GPU, round 6.

  * ragged M in the MFMA GEMMs (tuning key 43): 64-row blocks / waves that lie wholly beyond M issue no operand reads and no MFMAs.  The valid
    rows are computed by exactly the same instructions, so not a bit may differ from the full issue -- every epilogue, every tile kernel the ViT
    shapes use (M = 3075 = 12 x 256 + 3 is the shape this is for: modeling_intern_vit.py:124,136,184-185 at 3 tiles).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
from gpu_util import DT, CODE, TOL, dev, rnd, rel, ptr, sync, randn
from omchat_amd import _lib

DTS = ["bf16", "f16"]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("tile", [2, 10, 11, 9, 1, 12])
@pytest.mark.parametrize("M", [259, 200, 321, 3075, 64, 513])
def test_gemm_dead_row_blocks_skipped_same_bits(gpu_lib, dt, tile, M):
    N, K = 1024, 192
    A = rnd(randn((M, K), 1), dt); W = rnd(randn((N, K), 2, 0.05), dt)
    bias = rnd(randn((N,), 3, 0.1), dt); ls = rnd(randn((N,), 4, 0.1) + 0.1, dt); resid = rnd(randn((M, N), 5), dt)
    dA, dW, db, dl, dr = dev(A, dt), dev(W, dt), dev(bias, dt), dev(ls, dt), dev(resid, dt)
    y = A @ W.t() + bias
    for epi in (_lib.EPI_NONE, _lib.EPI_GELU, _lib.EPI_LS_RESID):
        outs = {}
        try:
            for key in (1, 0):
                gpu_lib.omchat_op_set_tuning(43, key)
                # one guard row behind the matrix: nothing may be written beyond row M - 1
                out = torch.full((M + 1, N), 77.0, dtype=DT[dt], device="cuda")
                _lib.check(gpu_lib.omchat_op_gemm(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(out), N, M, N, K, ptr(db), ptr(dl), ptr(dr), N, epi, tile, None))
                sync()
                outs[key] = out.clone()
        finally:
            gpu_lib.omchat_op_set_tuning(43, 1)
        assert torch.equal(outs[1], outs[0]), (epi, tile, M)
        assert bool((outs[1][M] == 77.0).all())
        if epi == _lib.EPI_NONE:
            assert rel(outs[1][:M], rnd(y, dt)) < TOL[dt]


# ---------------------------------------------------------------------------------------------------------------------
# the ViT layer with its norms folded into the GEMMs (model.hip vit_layer_fused; modeling_intern_vit.py:39-44,143-148,210-222)
# ---------------------------------------------------------------------------------------------------------------------
import ctypes as C
import numpy as np


def _f32(t):
    return t.float().contiguous().cuda()


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("tile,M,N,K", [(2, 515, 3200, 128), (10, 515, 3200, 128), (9, 300, 1000, 64), (2, 1025, 9600, 192), (0, 3075, 3200, 256), (1, 130, 520, 64), (12, 3075, 3200, 128), (13, 2050, 12800, 64)])
def test_gemm_statistics_epilogue_and_row_scale(gpu_lib, dt, tile, M, N, K):
    """EPI_LS_RESID_STATS / EPI_NONE_STATS: the stored values are those of the plain epilogue bit for bit, and a row's slots sum to the sum of
    squares of the STORED 16-bit values; row_scale multiplies the fp32 accumulators (y = rstd * (x W^T): InternRMSNorm :39-44 in front of a linear map)"""
    A = rnd(randn((M, K), 1), dt); W = rnd(randn((N, K), 2, 0.05), dt)
    bias = rnd(randn((N,), 3, 0.1), dt); ls = rnd(randn((N,), 4, 0.1) + 0.1, dt); resid = rnd(randn((M, N), 5), dt)
    rs = (torch.rand(M, generator=torch.Generator().manual_seed(6)) + 0.5)
    dA, dW, db, dl, dr, drs = dev(A, dt), dev(W, dt), dev(bias, dt), dev(ls, dt), dev(resid, dt), _f32(rs)
    cap, ld = (N + 47) // 48 + 2, M + 5                # slot-major statistics: [slots][ld >= M]
    for epi, base in ((_lib.EPI_LS_RESID_STATS, _lib.EPI_LS_RESID), (_lib.EPI_NONE_STATS, _lib.EPI_NONE)):
        for scale in (None, drs):
            plain = torch.full((M, N), 77.0, dtype=DT[dt], device="cuda")
            _lib.check(gpu_lib.omchat_op_gemm_fused(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(plain), N, M, N, K, ptr(db), ptr(dl), ptr(dr), N, base, tile,
                                                    ptr(scale) if scale is not None else None, None, 0, 0, 0, 0.0, None, 0, None, None))
            out = torch.full((M, N), 77.0, dtype=DT[dt], device="cuda")
            stats = torch.full((cap, ld), -1.0, dtype=torch.float32, device="cuda")
            slot = C.c_int(0)
            _lib.check(gpu_lib.omchat_op_gemm_fused(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(out), N, M, N, K, ptr(db), ptr(dl), ptr(dr), N, epi, tile,
                                                    ptr(scale) if scale is not None else None, None, 0, 0, 0, 0.0, ptr(stats), ld, C.byref(slot), None))
            sync()
            ns = slot.value                                   # slots written: one per wave tile of the kernel(s) that ran
            assert 1 <= ns <= cap
            assert torch.equal(out, plain), (epi, tile)
            got = stats[:ns, :M].double().sum(0).cpu()
            want = out.double().pow(2).sum(1).cpu()
            assert bool((stats[:ns, :M] >= 0).all()) and bool((stats[ns:] == -1.0).all()) and bool((stats[:, M:] == -1.0).all())       # every slot written once, nothing beyond
            assert float(((got - want).abs() / want).max()) < 1e-5, (epi, tile)
            # a slot is one wave tile's run of columns: for the single-launch tile kernels the widths are 64, 112 or 48
            if tile not in (12, 13):
                sq = out.double().pow(2)
                ok = False
                for w in (64, 112, 48):
                    if (N + w - 1) // w != ns:
                        continue
                    pad = torch.zeros(M, ns * w, dtype=torch.float64, device="cuda"); pad[:, :N] = sq
                    want_s = pad.reshape(M, ns, w).sum(-1).t()
                    ok = ok or float(((stats[:ns, :M].double() - want_s).abs() / (want_s + 1e-30)).max()) < 1e-5
                assert ok, (epi, tile, ns)
            if base == _lib.EPI_NONE:
                acc = A @ W.t()
                ref = rnd((acc * rs[:, None] if scale is not None else acc) + bias, dt)
                assert rel(out, ref) < TOL[dt]
    # the row scale finished INSIDE the launch from statistics slots: slots whose sum gives rs^-2 * dim - eps, split unevenly over 29 slots
    dim, eps, ns, rld = 3200, 1e-6, 29, M + 3
    tot = (rs.double().pow(-2) - eps) * dim
    wts = torch.rand(M, ns, generator=torch.Generator().manual_seed(7)).double() + 0.1
    slots = torch.full((ns + 2, rld), 1e30)          # slot-major; the slots behind the last one must never be read
    slots[:ns, :M] = (wts / wts.sum(1, keepdim=True) * tot[:, None]).float().t()
    dsl = _f32(slots)
    for epi in (_lib.EPI_NONE, _lib.EPI_GELU):
        a_ = torch.full((M, N), 77.0, dtype=DT[dt], device="cuda"); b_ = torch.full((M, N), 77.0, dtype=DT[dt], device="cuda")
        _lib.check(gpu_lib.omchat_op_gemm_fused(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(a_), N, M, N, K, ptr(db), None, None, 0, epi, tile,
                                                ptr(drs), None, 0, 0, 0, 0.0, None, 0, None, None))
        _lib.check(gpu_lib.omchat_op_gemm_fused(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(b_), N, M, N, K, ptr(db), None, None, 0, epi, tile,
                                                None, ptr(dsl), rld, ns, dim, eps, None, 0, None, None))
        sync()
        # the same factor up to fp32 rounding of the slot sums: equal to one 16-bit ulp on isolated entries
        assert rel(b_, a_) < 2e-3 and float((a_ != b_).float().mean()) < 0.02, (epi, tile)


@pytest.mark.parametrize("dt", DTS)
def test_stats_finish_row_sumsq_and_fold_cols(gpu_lib, dt):
    M, H = 301, 3200
    x = rnd(randn((M, H), 1) * 3.0, dt)
    dx = dev(x, dt)
    stats = torch.zeros(3, M + 7, dtype=torch.float32, device="cuda")
    _lib.check(gpu_lib.omchat_op_row_sumsq(CODE[dt], ptr(dx), H, M, H, ptr(stats), None))
    sync()
    want = x.double().pow(2).sum(1)
    assert float(((stats[0, :M].double().cpu() - want).abs() / want).max()) < 1e-5 and bool((stats[1:] == 0).all()) and bool((stats[0, M:] == 0).all())
    slots = torch.rand(60, M + 9, generator=torch.Generator().manual_seed(2)) * 40          # slot-major
    rstd = torch.empty(M, dtype=torch.float32, device="cuda")
    dslots = _f32(slots)
    _lib.check(gpu_lib.omchat_op_stats_finish(ptr(dslots), M + 9, 3, 50, 1, M, H, 1e-6, ptr(rstd), None))
    sums = torch.empty(M, 2, dtype=torch.float32, device="cuda")
    _lib.check(gpu_lib.omchat_op_stats_finish(ptr(dslots), M + 9, 2, 29, 2, M, 0, 0.0, ptr(sums), None))      # two groups of 29 slots, raw sums
    sync()
    ref = torch.rsqrt(slots[3:53, :M].double().sum(0) / H + 1e-6)
    assert float(((rstd.double().cpu() - ref).abs() / ref).max()) < 1e-5
    for g in range(2):
        want_g = slots[2 + 29 * g:2 + 29 * (g + 1), :M].double().sum(0)
        assert float(((sums[:, g].double().cpu() - want_g).abs() / want_g).max()) < 1e-5
    W = rnd(randn((96, H), 3, 0.05), dt); n = rnd(randn((H,), 4, 0.05) + 1.0, dt)
    out = torch.empty(96, H, dtype=DT[dt], device="cuda")
    dW, dn = dev(W, dt), dev(n, dt)
    _lib.check(gpu_lib.omchat_op_fold_cols(CODE[dt], ptr(dW), ptr(dn), ptr(out), 96, H, None))
    sync()
    assert torch.equal(out.float().cpu(), rnd(W * n[None, :], dt))


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B,S,H", [(2, 1025, 25), (3, 257, 4), (1, 65, 25)])
def test_qk_norm_from_gemm_statistics_vs_the_separate_norm_launch(gpu_lib, dt, B, S, H):
    """the joint-head q / k RMSNorm (modeling_intern_vit.py:143-148) split in two: K normalised in place straight from the statistics slots
    (omchat_op_vit_knorm_slots, which leaves the q sums), Q normalised + scaled where attn2_kernel loads it (omchat_op_mha_qnorm) -- against the round-5 pair
    omchat_op_vit_qknorm + omchat_mha_fwd, which keeps the same rounding points (statistics summed in another fp32 order: the normed
    values may differ in the last 16-bit ulp of a few entries), and against the fp32 restatement of _naive_attn"""
    Cq = H * 128
    M = B * S
    qkv = rnd(randn((M, 3 * Cq), 1) * 1.5, dt)
    wq = rnd(randn((Cq,), 2, 0.05) + 1.0, dt); wk = rnd(randn((Cq,), 3, 0.05) + 1.0, dt)
    qscale = 128 ** -0.5
    # slots as the qkv GEMM's epilogue leaves them: sum of squares per 64-column block
    slots = qkv.double().pow(2).reshape(M, 3 * Cq // 64, 64).sum(-1).float().t().contiguous()       # slot-major [slots][M]
    ld = M + 3
    st = torch.zeros(slots.shape[0], ld); st[:, :M] = slots
    dst = _f32(st)
    # round-5 pair
    a = dev(qkv, dt).clone()
    dwq, dwk, draw = dev(wq, dt), dev(wk, dt), dev(qkv, dt)
    _lib.check(gpu_lib.omchat_op_vit_qknorm(CODE[dt], ptr(a), 3 * Cq, ptr(dwq), ptr(dwk), M, Cq, Cq, 1e-6, qscale, None))
    o_old = torch.empty(B, S, H, 128, dtype=DT[dt], device="cuda")
    _lib.check(gpu_lib.omchat_mha_fwd(ptr(a), B, S, H, 1.0, 0, ptr(o_old), CODE[dt], None))
    # round-6 pair: K half of the norm in place straight from the slots (which also leaves the q sums), Q on load
    b = dev(qkv, dt).clone()
    sums = torch.empty(M, dtype=torch.float32, device="cuda")
    kview = b.view(-1)[Cq:]
    _lib.check(gpu_lib.omchat_op_vit_knorm_slots(CODE[dt], ptr(kview), 3 * Cq, ptr(dwk), M, Cq, Cq, 1e-6, ptr(dst), ld, Cq // 64, ptr(sums), None))
    o_new = torch.empty(B, S, H, 128, dtype=DT[dt], device="cuda")
    _lib.check(gpu_lib.omchat_op_mha_qnorm(CODE[dt], ptr(b), B, S, H, ptr(sums), 1, Cq, ptr(dwq), 1e-6, qscale, ptr(o_new), None))
    sync()
    # K rows: the same values up to the last ulp of isolated entries; V and Q untouched by the K pass
    ka, kb = a[:, Cq:2 * Cq].float(), b[:, Cq:2 * Cq].float()
    assert rel(kb, ka) < 2e-3 and float((ka != kb).float().mean()) < 0.02
    assert torch.equal(b[:, :Cq], draw[:, :Cq]) and torch.equal(b[:, 2 * Cq:], draw[:, 2 * Cq:])
    assert rel(o_new, o_old) < TOL[dt]
    # fp32 restatement (oracle/vit.py vit_attention, the q / k / softmax / v part) on the 16-bit inputs
    q, k, v = qkv[:, :Cq], qkv[:, Cq:2 * Cq], qkv[:, 2 * Cq:]
    nrm = lambda t, w: w * (t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + 1e-6))
    qn = nrm(q, wq).reshape(B, S, H, 128).transpose(1, 2); kn = nrm(k, wk).reshape(B, S, H, 128).transpose(1, 2)
    vv = v.reshape(B, S, H, 128).transpose(1, 2)
    ref = (torch.softmax((qn * qscale) @ kn.transpose(-2, -1), dim=-1) @ vv).transpose(1, 2)
    assert rel(o_new, ref) < 2 * TOL[dt]
    assert rel(o_old, ref) < 2 * TOL[dt]


@pytest.mark.parametrize("dt", DTS)
def test_fused_vit_layer_equals_the_round5_launches_within_one_op_tolerance(gpu_lib, dt):
    """tuning key 44: the whole tower with the fused layer against the eight-launch layer on the same weights and tiles (tiny config with the
    production head dim, 4 layers), and both against the oracle (modeling_intern_vit.py:210-222 restated in oracle/vit.py)"""
    import oracle
    from omchat_amd import synth
    from omchat_amd.config import tiny
    from omchat_amd.engine import Engine
    from gpu_util import TOL_DEEP
    cfg = tiny(layers_v=4, heads_v=3, mlp_v=1024)
    sd = synth.state_dict(cfg, 3)
    px = torch.from_numpy(synth.pixels(3, 56, 1))
    outs = {}
    try:
        for key in (1, 0):
            gpu_lib.omchat_op_set_tuning(44, key)
            e = Engine(cfg, dtype=dt, max_seq=64, max_batch=1, max_tiles=3)
            e.load_state_dict(sd)
            outs[key] = (e.vit_forward(px).float().cpu(), e.encode_images(px).float().cpu())
            sync()
            e.close()
    finally:
        gpu_lib.omchat_op_set_tuning(44, 1)
    sdt = {k: rnd(torch.from_numpy(v), dt) for k, v in sd.items()}
    tower = {k[len(synth.TOWER):]: v for k, v in sdt.items() if k.startswith(synth.TOWER)}
    ref = oracle.vision_tower_forward(px, tower, cfg.vision)
    for key in (1, 0):
        assert rel(outs[key][0], ref) < TOL_DEEP[dt], (key, rel(outs[key][0], ref))
    assert rel(outs[1][0], outs[0][0]) < TOL_DEEP[dt]
    assert rel(outs[1][1], outs[0][1]) < TOL_DEEP[dt]


# ---------------------------------------------------------------------------------------------------------------------
# sequence-parallel norms under tensor parallelism (model.hip gemm_sp, tuning key 45)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("min_rows", [0, 8])
def test_tp2_sequence_parallel_norms_vs_all_reduce_form_and_oracle(gpu_lib, dt, min_rows):
    """two rank contexts on one GPU (the hook harness of test_gpu_tp_single.py): the tower and the prefill with the row-parallel projections ending
    in reduce-scatter -> residual + norm on the owned rows -> all-gather (counters say the form ran; min_rows = 8: several row chunks on the
    communication stream, the consumer GEMMs behind the per-chunk gather events) against the all-reduce form (key 45 = 0) and the oracle; ragged
    row blocks: 2 tiles x 17 tokens = 34 tower rows, 38 prefill rows over 2 ranks"""
    import ctypes as C
    import oracle
    from omchat_amd import synth
    from omchat_amd.config import tiny
    from omchat_amd.engine import Engine
    from gpu_util import TOL_DEEP
    from test_gpu_tp_single import Group, _run_ranks, T32
    cfg = tiny(q_heads=4, kv_heads=2, heads_v=3)
    sd = synth.state_dict(cfg, 13)
    px = T32(synth.pixels(2, 56, 1))
    ids = torch.tensor([[3, -200, 17, -200, 19, 20, 21]])
    res = {}
    try:
        if min_rows:
            gpu_lib.omchat_op_set_tuning(4, min_rows)
        for key in (1, 0):
            gpu_lib.omchat_op_set_tuning(45, key)
            grp = Group(2)
            engines, hooks = [], []
            for r in range(2):
                e = Engine(cfg, dtype=dt, max_seq=128, max_batch=1, max_tiles=2, tp_rank=r, tp_size=2, comm=C.c_void_p(1))
                h = grp.hook_for(r)
                _lib.check(gpu_lib.omchat_set_allreduce_hook(e.h, C.cast(h, C.c_void_p), None))
                e.load_state_dict(sd)
                engines.append(e); hooks.append(h)

            def run(r):
                e = engines[r]
                feats = e.encode_images(px)
                embeds, lengths, _ = e.splice(ids, None, feats)
                logits, hid = e.prefill(embeds, lengths, want_hidden=True)
                torch.cuda.synchronize()
                return feats.float().cpu(), logits.float().cpu(), hid.float().cpu()
            out = _run_ranks(run, 2)
            st = engines[0].comm_stats()
            assert (st["sp_reduce_scatters"] > 0) == (key == 1) and (st["sp_all_gathers"] > 0) == (key == 1), st
            assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][2], out[1][2])      # both ranks hold the same gathered stream
            res[key] = out[0]
            for e in engines:
                e.close()
    finally:
        gpu_lib.omchat_op_set_tuning(45, 1)
        gpu_lib.omchat_op_set_tuning(4, 1024)
    sdt = {k: T32(v) for k, v in sd.items()}
    ref_feats = oracle.encode_images(px, sdt, cfg.vision)
    for key in (1, 0):
        assert rel(res[key][0], ref_feats) < TOL_DEEP[dt], (key, rel(res[key][0], ref_feats))
    # the two forms differ by where the residual joins the sum (one rounding point): equal within one multi-layer tolerance
    assert rel(res[1][0], res[0][0]) < TOL_DEEP[dt] and rel(res[1][2], res[0][2]) < TOL_DEEP[dt]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B,S,H", [(2, 1025, 25), (1, 65, 3), (3, 129, 2), (1, 1026, 2), (2, 64, 2)])
def test_mha_odd_key_folded_into_the_initial_softmax_state(gpu_lib, dt, B, S, H):
    """tuning key 46: a key count of whole 64-key tiles + 1 (the ViT's 1025) initialises the online softmax with the odd key (m = q . k, l = 1, O = v)
    instead of running a tile step for it -- against the 17-step form (equal to 16-bit rounding: the odd key's probability is exact either way) and
    against the fp32 softmax(q k^T) v (modeling_intern_vit.py:148-151); key counts that are not tiles + 1 take the old path bit for bit"""
    qkv = rnd(randn((B, S, 3, H, 128), 1) * 0.7, dt)
    d = dev(qkv, dt)
    outs = {}
    try:
        for key in (1, 0):
            gpu_lib.omchat_op_set_tuning(46, key)
            o = torch.full((B, S, H, 128), float("nan"), dtype=DT[dt], device="cuda")
            _lib.check(gpu_lib.omchat_mha_fwd(ptr(d), B, S, H, 128 ** -0.5, 0, ptr(o), CODE[dt], None))
            sync()
            outs[key] = o
    finally:
        gpu_lib.omchat_op_set_tuning(46, 1)
    q, k, v = [qkv[:, :, i].transpose(1, 2) for i in range(3)]
    ref = (torch.softmax((q @ k.transpose(-2, -1)) * 128 ** -0.5, dim=-1) @ v).transpose(1, 2)
    assert torch.isfinite(outs[1].float()).all()
    assert rel(outs[1], ref) < TOL[dt] and rel(outs[0], ref) < TOL[dt]
    if S % 64 == 1 and S > 64:
        assert rel(outs[1], outs[0]) < TOL[dt] / 2
    else:
        assert torch.equal(outs[1], outs[0])
