"""GPU (one device): the C++ tensor-parallel dataflow end to end.  Two rank contexts (tp_size = 2) live on the same GPU,
run in two threads, and their all-reduces are served by the test through omchat_set_allreduce_hook (sum of both
ranks' buffers).  Outputs must equal the tp_size = 1 engine / the oracle.  RCCL itself is exercised by bench.py --gpus N."""
import ctypes as C
import threading
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
from gpu_util import rel, sync, TOL_DEEP
from omchat_amd import synth, _lib
from omchat_amd.config import tiny
from omchat_amd.engine import Engine
import oracle

T32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
TDT = {_lib.F16: torch.float16, _lib.BF16: torch.bfloat16, _lib.F32: torch.float32}


class Group:
    def __init__(self, n):
        self.n = n
        self.barrier = threading.Barrier(n)
        self.slots = [None] * n
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.calls = 0

    def hook_for(self, rank):
        def hook(user, buf, count, dtype, stream):
            torch.cuda.synchronize()
            self.slots[rank] = buf
            self.barrier.wait()
            if rank == 0:
                tdt = TDT[dtype]
                nbytes = count * torch.empty(0, dtype=tdt).element_size()
                parts = []
                for b in self.slots:
                    t = torch.empty(count, dtype=tdt, device="cuda")
                    assert self.hip.hipMemcpy(t.data_ptr(), b, nbytes, 3) == 0
                    parts.append(t.float())
                s = sum(parts).to(tdt)
                for b in self.slots:
                    assert self.hip.hipMemcpy(b, s.data_ptr(), nbytes, 3) == 0
                torch.cuda.synchronize()
                self.calls += 1
            self.barrier.wait()
            return 0
        return _lib.ALLREDUCE_FN(hook)


def _run_ranks(fn, n):
    out, err = [None] * n, [None] * n
    def work(r):
        try:
            out[r] = fn(r)
        except BaseException as e:          # noqa
            err[r] = e
    th = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in th: t.start()
    for t in th: t.join(timeout=300)
    for e in err:
        if e is not None:
            raise e
    return out


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("q,kv,hv", [(4, 2, 3), (7, 1, 2)])
def test_tp2_on_one_gpu_equals_tp1(gpu_lib, dt, q, kv, hv):
    cfg = tiny(q_heads=q, kv_heads=kv, heads_v=hv)
    sd = synth.state_dict(cfg, 13)
    grp = Group(2)
    engines, hooks = [], []
    for r in range(2):
        e = Engine(cfg, dtype=dt, max_seq=128, max_batch=1, max_tiles=2, tp_rank=r, tp_size=2, comm=C.c_void_p(1))
        h = grp.hook_for(r)
        _lib.check(gpu_lib.omchat_set_allreduce_hook(e.h, C.cast(h, C.c_void_p), None))
        e.load_state_dict(sd)
        engines.append(e); hooks.append(h)
    px = T32(synth.pixels(2, 56, 1))
    ids = torch.tensor([[3, -200, 17, -200, 19, 20]])

    def run(r):
        e = engines[r]
        feats = e.encode_images(px)
        embeds, lengths, _ = e.splice(ids, None, feats)
        logits, _ = e.prefill(embeds, lengths)
        nxt, lg = e.decode_step(torch.tensor([5]), want_logits=True)
        torch.cuda.synchronize()
        return feats.float().cpu(), logits.float().cpu(), lg.float().cpu(), int(nxt[0])

    res = _run_ranks(run, 2)
    assert grp.calls > 0
    assert torch.equal(res[0][0], res[1][0])                      # both ranks hold identical activations after the all-reduce
    sdt = {k: T32(v) for k, v in sd.items()}
    ref_feats = oracle.encode_images(px, sdt, cfg.vision)
    assert rel(res[0][0], ref_feats) < TOL_DEEP[dt]
    ref_logits, cache, _ = oracle.prefill(ids, px, sdt, cfg.vision, cfg.text)
    V = cfg.text["vocab_size"]
    full = torch.cat([res[0][1], res[1][1]], dim=-1)              # vocab-parallel lm_head: rank r holds rows [r*V/2, (r+1)*V/2)
    assert full.shape[-1] == V
    assert rel(full[0], ref_logits[0, -1]) < TOL_DEEP[dt]
    ref_step = oracle.decode_step(torch.tensor([[5]]), sdt, cfg.text, cache)
    assert rel(torch.cat([res[0][2], res[1][2]], dim=-1)[0], ref_step[0, 0]) < TOL_DEEP[dt]
    assert res[0][3] == res[1][3]                                 # vocab-parallel greedy: (max, argmax) pairs exchanged
    full_step = torch.cat([res[0][2], res[1][2]], dim=-1)[0]
    assert res[0][3] == int(torch.argmax(full_step))
    for e in engines:
        e.close()


@pytest.mark.parametrize("min_rows", [8, 16])
def test_tp2_with_pipelined_allreduce_chunks(gpu_lib, min_rows):
    """the row-parallel projections cut into 2 / 4 row chunks with the all-reduce on the communication stream (gemm_allreduce):
    same results as the unchunked TP run, and more all-reduce calls"""
    cfg = tiny(q_heads=4, kv_heads=2, heads_v=2)
    sd = synth.state_dict(cfg, 13)
    px = T32(synth.pixels(2, 56, 1))
    ids = torch.tensor([[3, -200, 17, -200, 19, 20]])

    def run_all(knob):
        gpu_lib.omchat_op_set_tuning(4, knob)
        grp = Group(2)
        engines, hooks = [], []
        for r in range(2):
            e = Engine(cfg, dtype="bf16", max_seq=128, max_batch=1, max_tiles=2, tp_rank=r, tp_size=2, comm=C.c_void_p(1))
            h = grp.hook_for(r)
            _lib.check(gpu_lib.omchat_set_allreduce_hook(e.h, C.cast(h, C.c_void_p), None))
            e.load_state_dict(sd)
            engines.append(e); hooks.append(h)

        def run(r):
            e = engines[r]
            feats = e.encode_images(px)
            embeds, lengths, _ = e.splice(ids, None, feats)
            logits, _ = e.prefill(embeds, lengths)
            torch.cuda.synchronize()
            return feats.float().cpu(), logits.float().cpu()
        res = _run_ranks(run, 2)
        for e in engines:
            e.close()
        return res, grp.calls
    try:
        base, calls0 = run_all(1 << 20)
        chunked, calls1 = run_all(min_rows)
    finally:
        gpu_lib.omchat_op_set_tuning(4, 1024)
    assert calls1 > calls0
    for r in range(2):
        assert torch.equal(base[r][0], chunked[r][0]) and torch.equal(base[r][1], chunked[r][1])


def test_rccl_bootstrap_single_rank(gpu_lib):
    """omchat_comm_unique_id / omchat_comm_init / omchat_comm_destroy through the ctypes binding (world of 1: the multi-GPU
    RCCL data path itself is exercised by `bench.py --gpus N` under torchrun)"""
    buf = C.create_string_buffer(128)
    _lib.check(gpu_lib.omchat_comm_unique_id(buf))
    assert any(b != 0 for b in buf.raw)
    comm = C.c_void_p()
    _lib.check(gpu_lib.omchat_comm_init(bytes(buf.raw), 0, 1, C.byref(comm)))
    assert comm.value
    # the data-path collective on that communicator (sum over 1 rank = identity) for the three dtypes the path reduces
    for dt, code in ((torch.bfloat16, _lib.BF16), (torch.float16, _lib.F16), (torch.float32, _lib.F32)):
        x = torch.randn(3584 * 3, device="cuda").to(dt)
        y = x.clone()
        _lib.check(gpu_lib.omchat_comm_allreduce(comm, C.c_void_p(y.data_ptr()), y.numel(), code, None))
        torch.cuda.synchronize()
        assert torch.equal(x, y)
    gpu_lib.omchat_comm_destroy(comm)


def test_tp2_fp8_decode_on_one_gpu(gpu_lib):
    """weight-only e4m3 decode under tensor parallelism (round 3: every rank quantises and streams its own shard; per-row scales of the
    row-parallel o_proj / down_proj cover the LOCAL K slice): two rank contexts on one GPU, logits against the oracle on the 16-bit
    weights with the fp8 tolerance of tests/test_gpu_fp8.py, both ranks agreeing bit for bit on the activations they share"""
    dt = "bf16"
    cfg = tiny(q_heads=4, kv_heads=2, heads_v=2)
    sd = synth.state_dict(cfg, 13)
    grp = Group(2)
    engines, hooks = [], []
    for r in range(2):
        e = Engine(cfg, dtype=dt, max_seq=128, max_batch=1, max_tiles=2, tp_rank=r, tp_size=2, comm=C.c_void_p(1), vision=False)
        h = grp.hook_for(r)
        _lib.check(gpu_lib.omchat_set_allreduce_hook(e.h, C.cast(h, C.c_void_p), None))
        e.load_state_dict({k: v for k, v in sd.items() if not k.startswith(synth.TOWER) and "mm_projector" not in k})
        e.enable_fp8_decode(True)
        engines.append(e); hooks.append(h)
    x = torch.randn(1, 11, 256, generator=torch.Generator().manual_seed(3)) * 0.5

    def run(r):
        e = engines[r]
        e.prefill(x)
        outs = []
        tok = torch.tensor([5])
        for _ in range(3):
            tok, lg = e.decode_step(tok, want_logits=True)
            outs.append(lg.float().cpu())
        torch.cuda.synchronize()
        return outs, int(tok[0])

    res = _run_ranks(run, 2)
    assert res[0][1] == res[1][1]
    sdt = {k: T32(v) for k, v in sd.items()}
    cache = oracle.KVCache(cfg.text["num_hidden_layers"])
    oracle.qwen2_model(x, sdt, cfg.text, cache)
    tok = 5
    for s in range(3):
        ref = oracle.decode_step(torch.tensor([[tok]]), sdt, cfg.text, cache)[0, 0]
        full = torch.cat([res[0][0][s], res[1][0][s]], dim=-1)[0]
        assert rel(full, ref) < 0.12, (s, rel(full, ref))           # e4m3 weights vs 16-bit weights (same bound as the TP = 1 fp8 tests)
        tok = int(torch.argmax(full))
    for e in engines:
        e.close()


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("min_rows", [1 << 20, 8])
def test_tp2_fp32_partial_sums_are_closer_to_tp1(gpu_lib, dt, min_rows):
    """tuning key 29: the row-parallel projections of the ViT and the prefill hand their raw fp32 accumulators to the all-reduce and the epilogue
    (bias, layer scale, residual) runs once on the sum (launch_tp_finish) -- unchunked and with the all-reduce pipelined in row chunks.  Both
    ranks hold the same bits; against the TP = 1 ENGINE (same kernels, one rank) the error is within the multi-layer tolerance and not above
    the default path's (every rank rounding its own partial)"""
    cfg = tiny(q_heads=4, kv_heads=2, heads_v=2)
    sd = synth.state_dict(cfg, 13)
    px = T32(synth.pixels(2, 56, 1))
    ids = torch.tensor([[3, -200, 17, -200, 19, 20]])
    e1 = Engine(cfg, dtype=dt, max_seq=128, max_batch=1, max_tiles=2)
    e1.load_state_dict(sd)
    f1 = e1.encode_images(px)
    emb1, len1, _ = e1.splice(ids, None, f1)
    l1, _ = e1.prefill(emb1, len1)
    torch.cuda.synchronize()
    f1, l1 = f1.float().cpu(), l1.float().cpu()
    e1.close()

    def run_all(f32):
        gpu_lib.omchat_op_set_tuning(29, f32); gpu_lib.omchat_op_set_tuning(4, min_rows)
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
            logits, _ = e.prefill(embeds, lengths)
            torch.cuda.synchronize()
            return feats.float().cpu(), logits.float().cpu()
        res = _run_ranks(run, 2)
        for e in engines:
            e.close()
        return res
    try:
        base = run_all(0)
        f32 = run_all(1)
    finally:
        gpu_lib.omchat_op_set_tuning(29, 0); gpu_lib.omchat_op_set_tuning(4, 1024)
    assert torch.equal(f32[0][0], f32[1][0])
    err = {}
    for name, res in (("default", base), ("fp32", f32)):
        full = torch.cat([res[0][1], res[1][1]], dim=-1)
        err[name] = (rel(res[0][0], f1), rel(full, l1))
        assert err[name][0] < TOL_DEEP[dt] and err[name][1] < TOL_DEEP[dt], (name, err[name])
    assert err["fp32"][0] <= err["default"][0] * 1.25 + 1e-6 and err["fp32"][1] <= err["default"][1] * 1.25 + 1e-6, err


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_gemm_fp32_output_epilogue(gpu_lib, dt):
    """EPI_F32OUT: C (float) = A W^T, the raw fp32 accumulators of the 256^2 and the 128^2 kernels, ragged M / N edges"""
    from gpu_util import DT, CODE, dev, ptr, randn, rnd
    for M, N, K in ((300, 264, 128), (1100, 3200, 256)):
        A = rnd(randn((M, K), 1), dt); W = rnd(randn((N, K), 2), dt)
        dA, dW = dev(A, dt), dev(W, dt)
        out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
        _lib.check(gpu_lib.omchat_op_gemm(CODE[dt], ptr(dA), K, ptr(dW), K, ptr(out), N, M, N, K, None, None, None, 0, 6, 0, None))
        sync()
        ref = A.double() @ W.double().t()
        assert torch.isfinite(out).all()
        assert float((out.double().cpu() - ref).norm() / ref.norm()) < 1e-5


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("b,dims", [(5, dict(q_heads=4, kv_heads=2)), (20, dict(q_heads=4, kv_heads=2)),
                                    (20, dict(q_heads=8, kv_heads=2, hidden_t=1024, mlp_t=1024))])
def test_tp2_batched_decode_equals_the_oracle(gpu_lib, dt, b, dims):
    """round 5: BASELINE configs[2] decodes a BATCH under tensor parallelism (packed activations, split-K slices summed over the ranks inside the
    residual + RMSNorm step, the shard-width launch shapes of tuning key 34) -- two rank contexts on one GPU, ragged right-padded batch of b
    sequences, two decode steps: every checked row against the oracle run of that sequence alone (transformers modeling_qwen2.py:269-298), both
    ranks' greedy picks equal"""
    cfg = tiny(layers_t=2, **dims)
    keep = lambda k: not k.startswith(synth.TOWER) and "mm_projector" not in k
    sd = {k: v for k, v in synth.state_dict(cfg, 17).items() if keep(k)}
    H = cfg.text["hidden_size"]
    grp = Group(2)
    engines, hooks = [], []
    for r in range(2):
        e = Engine(cfg, dtype=dt, max_seq=96, max_batch=b, max_tiles=1, tp_rank=r, tp_size=2, comm=C.c_void_p(1), vision=False)
        h = grp.hook_for(r)
        _lib.check(gpu_lib.omchat_set_allreduce_hook(e.h, C.cast(h, C.c_void_p), None))
        e.load_state_dict(sd)
        engines.append(e); hooks.append(h)
    S = 24
    from gpu_util import rnd
    x = rnd(torch.randn(b, S, H, generator=torch.Generator().manual_seed(b)) * 0.5, dt)
    lens = [S - (i % 5) for i in range(b)]
    toks = torch.arange(b) % 300 + 5

    def run(r):
        e = engines[r]
        e.prefill(x, lens)
        n1, l1 = e.decode_step(toks, want_logits=True)
        n2, l2 = e.decode_step(n1, want_logits=True)
        torch.cuda.synchronize()
        return l1.float().cpu(), l2.float().cpu(), n1.cpu(), n2.cpu()

    res = _run_ranks(run, 2)
    assert torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][3], res[1][3])      # vocab-parallel greedy agrees on both ranks
    full1 = torch.cat([res[0][0], res[1][0]], dim=-1); full2 = torch.cat([res[0][1], res[1][1]], dim=-1)
    sdt = {k: rnd(T32(v), dt) for k, v in sd.items()}
    for i in sorted({0, 1, b // 2, b - 1}):
        cache = oracle.KVCache(cfg.text["num_hidden_layers"])
        oracle.qwen2_model(x[i:i + 1, :lens[i]], sdt, cfg.text, cache)
        o1 = oracle.decode_step(toks[i:i + 1][None].long(), sdt, cfg.text, cache)[0, 0]
        o2 = oracle.decode_step(res[0][2][i:i + 1][None].long(), sdt, cfg.text, cache)[0, 0]
        assert rel(full1[i], o1) < TOL_DEEP[dt] and rel(full2[i], o2) < TOL_DEEP[dt], (i, rel(full1[i], o1), rel(full2[i], o2))
        assert int(res[0][2][i]) == int(torch.argmax(full1[i]))
    for e in engines:
        e.close()
