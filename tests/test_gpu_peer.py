"""GPU (one device): the peer all-reduce of csrc/comm.hip (IPC-mapped buffers + flag barrier; the xGMI collective for
decode-sized tensor-parallel sums).

Exercised three ways on ONE GPU: (1) rank members in threads of one process (pointers exchanged directly), (2) rank members in
separate PROCESSES that map each other's buffers with hipIpcGetMemHandle / hipIpcOpenMemHandle exactly as one-process-per-GPU
tensor parallelism does, (3) two tensor-parallel engine processes whose every all-reduce goes through it, against the TP = 1
engine.  Sums are checked bit for bit (fixed rank order, fp32, one rounding)."""
import ctypes as C
import os
import socket
import threading
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
from omchat_amd import _lib, synth, tp
from omchat_amd.config import tiny


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _local_group(lib, n, cap, max_blocks=16, fast=False, oneshot_max=0):
    peers = []
    for r in range(n):
        p = C.c_void_p(); h = C.create_string_buffer(64)
        _lib.check(lib.omchat_peer_create(r, n, cap, C.byref(p), h))
        peers.append(p)
    bases = (C.c_void_p * n)(*[lib.omchat_peer_base(p) for p in peers])
    for p in peers:
        _lib.check(lib.omchat_peer_connect_local(p, bases))
        _lib.check(lib.omchat_peer_set_mode(p, int(fast), oneshot_max, max_blocks))
    return peers


def _threads(fn, n):
    out, err = [None] * n, [None] * n
    def work(r):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                out[r] = fn(r)
                torch.cuda.current_stream().synchronize()
        except BaseException as e:      # noqa
            err[r] = e
    th = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in th: t.start()
    for t in th: t.join(timeout=300)
    for e in err:
        if e is not None:
            raise e
    return out


# Rank members as THREADS of one process are not a usable test vehicle: HIP multiplexes the streams of a process onto a few hardware
# queues, and two spin-waiting kernels that land in the same queue wait for each other until the timeout (measured: which pairs
# collide changes from run to run).  One process per rank -- the deployment shape -- has a queue set per rank: every multi-rank
# case below runs in separate processes over the IPC path.
def test_peer_timeout_is_reported_not_hung(gpu_lib):
    """a rank that never arrives: the waiting kernel gives up after its wall-clock bound and raises the sticky error word"""
    if os.environ.get("OMCHAT_SKIP_SLOW"):
        pytest.skip("slow")
    peers = _local_group(gpu_lib, 2, 1 << 20, max_blocks=2)
    x = torch.ones(4096, device="cuda")
    _lib.check(gpu_lib.omchat_peer_allreduce(peers[0], _lib.ptr(x), 4096, _lib.F32, _lib.cur_stream()))      # rank 1 never calls
    err = C.c_int(0)
    _lib.check(gpu_lib.omchat_peer_error(peers[0], C.byref(err)))
    assert err.value == 1
    _lib.check(gpu_lib.omchat_peer_error(peers[0], C.byref(err)))
    assert err.value == 0          # sticky until read, then cleared
    for p in peers:
        gpu_lib.omchat_peer_destroy(p)


# ---------------------------------------------------------------------------------------------------------------------
# separate processes on one GPU: the IPC path (hipIpcGetMemHandle / hipIpcOpenMemHandle), bootstrap over gloo
# ---------------------------------------------------------------------------------------------------------------------
def _proc_selftest(rank, size, port, q, fast=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        peer = tp.init_peer(rank, size, cap_bytes=4 << 20, max_blocks=8, fast=fast)
        ok, detail = tp.peer_selftest(peer, rank, size, iters=8, sizes=(16, 3584 * 4, 40000, 300 * 1024, 3 << 20, 9 << 20))
        if ok:                 # random payloads: the result is the fp32 sum in rank order rounded once (what gather + sum gives)
            for i, (dt, code, cnt) in enumerate([(torch.bfloat16, _lib.BF16, 3584 * 5), (torch.float16, _lib.F16, 3200 * 7),
                                                  (torch.float32, _lib.F32, 3 * 3584), (torch.bfloat16, _lib.BF16, 1 << 20)]):
                data = [torch.randn(cnt, generator=torch.Generator().manual_seed(100 * r + i)).to(dt) for r in range(size)]
                x = data[rank].cuda()
                _lib.check(_lib.lib().omchat_peer_allreduce(peer, _lib.ptr(x), cnt, code, _lib.cur_stream()))
                torch.cuda.synchronize()
                acc = torch.zeros(cnt, dtype=torch.float32)
                for r in range(size):
                    acc = acc + data[r].float()
                if not torch.equal(x.cpu(), acc.to(dt)):
                    ok, detail = False, f"random payload case {i} differs from the ordered fp32 sum"
        if ok:                 # round 6: the two halves of the all-reduce as collectives of their own (sequence-parallel norms), bit for bit
            lib = _lib.lib()
            for i, (dt, code, blk) in enumerate([(torch.bfloat16, _lib.BF16, 3200 * 3), (torch.float16, _lib.F16, 3584 * 5), (torch.float32, _lib.F32, 4 * 1000),
                                                  (torch.bfloat16, _lib.BF16, 3200 * 241)]):       # the last one: 1.5 MB blocks, several pieces per call
                data = [torch.randn(size, blk, generator=torch.Generator().manual_seed(1000 * r + i)).to(dt) for r in range(size)]
                x = data[rank].cuda()
                _lib.check(lib.omchat_peer_reduce_scatter(peer, _lib.ptr(x), blk, code, _lib.cur_stream()))
                torch.cuda.synchronize()
                acc = torch.zeros(blk, dtype=torch.float32)
                for r in range(size):
                    acc = acc + data[r][rank].float()
                want = data[rank].clone(); want[rank] = acc.to(dt)                # my block summed in rank order, the others untouched
                if not torch.equal(x.cpu(), want):
                    ok, detail = False, f"reduce-scatter case {i}: block {rank} is not the ordered fp32 sum / another block changed"
                # all-gather: every rank contributes its (summed) block; afterwards all ranks hold all summed blocks = the all-reduce
                _lib.check(lib.omchat_peer_all_gather(peer, _lib.ptr(x), blk, code, _lib.cur_stream()))
                torch.cuda.synchronize()
                full = torch.zeros(size, blk, dtype=torch.float32)
                for r in range(size):
                    full = full + data[r].float()
                if not torch.equal(x.cpu(), full.to(dt)):
                    ok, detail = False, f"all-gather after reduce-scatter, case {i}: differs from the ordered all-reduce"
        dist.barrier()
        _lib.lib().omchat_peer_destroy(peer)
        q.put((rank, ok, detail))
    except BaseException as e:      # noqa
        q.put((rank, False, repr(e)))
    finally:
        dist.destroy_process_group()


def _proc_fused_norm(rank, size, port, q):
    """omchat_peer_resid_rmsnorm == omchat_peer_allreduce(slices) + the local residual + RMSNorm kernel, bit for bit, on every rank"""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        lib = _lib.lib()
        peer = tp.init_peer(rank, size, cap_bytes=8 << 20, max_blocks=8)
        ok, detail = True, ""
        cases = [("bf16", 1, 3584, 3, 0), ("f16", 1, 3584, 8, 0), ("bf16", 5, 448, 1, 0), ("bf16", 32, 3584, 8, 2), ("f16", 12, 1024, 2, 1),
                 ("bf16", 1, 3584, 3, 0)]          # the first case again: slot parity / epoch bookkeeping after mixed calls
        for i, (dt, rows, H, ks, nb) in enumerate(cases):
            tdt = torch.bfloat16 if dt == "bf16" else torch.float16
            code = _lib.BF16 if dt == "bf16" else _lib.F16
            g = lambda seed: torch.Generator().manual_seed(seed)
            x0 = torch.randn(rows, H, generator=g(7 + i)).to(tdt)                       # replicated residual stream
            w = (torch.randn(H, generator=g(9 + i)) * 0.1 + 1).to(tdt)
            part = torch.randn(ks, rows, H, generator=g(1000 * rank + i)) * 0.3         # this rank's slices
            n_out = rows * H if nb == 0 else 16 * nb * H
            outs = []
            for fused in (True, False, True):
                x = x0.cuda().clone(); pt = part.cuda().clone(); xn = torch.zeros(n_out, dtype=tdt, device="cuda")
                if fused:
                    _lib.check(lib.omchat_peer_resid_rmsnorm(peer, code, _lib.ptr(x), H, _lib.ptr(pt), ks, _lib.ptr(w.cuda()), _lib.ptr(xn), H, rows, H,
                                                             1e-6, nb, _lib.cur_stream()))
                else:
                    _lib.check(lib.omchat_peer_allreduce(peer, _lib.ptr(pt), ks * rows * H, _lib.F32, _lib.cur_stream()))
                    _lib.check(lib.omchat_op_resid_rmsnorm(code, _lib.ptr(x), H, _lib.ptr(pt), ks, _lib.ptr(w.cuda()), _lib.ptr(xn), H, rows, H, 1e-6, nb,
                                                           _lib.cur_stream()))
                torch.cuda.synchronize()
                outs.append((x.cpu(), xn.cpu()))
            for j in (1, 2):
                if not (torch.equal(outs[0][0], outs[j][0]) and torch.equal(outs[0][1], outs[j][1])):
                    ok, detail = False, f"case {i} {dt} rows={rows} H={H} ks={ks} nb={nb}: fused != unfused (run {j})"
            if not torch.isfinite(outs[0][1].float()).all():
                ok, detail = False, f"case {i}: non-finite"
            gathered = [None] * size
            dist.all_gather_object(gathered, outs[0][0].float().numpy().tobytes())      # every rank must hold the same bits
            if any(gb != gathered[0] for gb in gathered):
                ok, detail = False, f"case {i}: ranks differ"
        err = C.c_int(0)
        _lib.check(lib.omchat_peer_error(peer, C.byref(err)))
        if err.value:
            ok, detail = False, "barrier timeout"
        dist.barrier()
        lib.omchat_peer_destroy(peer)
        q.put((rank, ok, detail))
    except BaseException as e:      # noqa
        import traceback
        q.put((rank, False, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _proc_tp_engine(rank, size, port, q, fuse=1):
    import torch.distributed as dist
    from omchat_amd.engine import Engine
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        cfg = tiny(q_heads=7, kv_heads=1, heads_v=3)
        peer = tp.init_peer(rank, size, cap_bytes=4 << 20, max_blocks=8)
        _lib.check(_lib.lib().omchat_op_set_tuning(9, fuse))      # 1: decode sums + residual + RMSNorm in one peer launch; 0: two launches
        e = Engine(cfg, dtype="bf16", max_seq=128, max_batch=2, max_tiles=2, tp_rank=rank, tp_size=size, comm=None)
        e.set_peer(peer, 0, all_sizes=True)
        e.fill_synthetic(13)                   # device-side sharding of the TP = 1 synthetic values
        px = torch.from_numpy(synth.pixels(2, 56, 1))
        ids = torch.tensor([[3, -200, 17, -200, 19, 20]])
        feats = e.encode_images(px)
        embeds, lengths, _ = e.splice(ids, None, feats)
        logits, _ = e.prefill(embeds, lengths)
        first = e.argmax(logits)
        toks = [int(first[0])]
        tok = first
        for _ in range(6):
            tok, _ = e.decode_step(tok)
            toks.append(int(tok[0]))
        # data-parallel tower: replicated vision-only context, tiles dealt to the ranks, one all-reduce gathers the features
        tower = Engine(cfg, dtype="bf16", max_seq=32, max_batch=1, max_tiles=2, text=False)
        tower.fill_synthetic(13)
        px3 = torch.from_numpy(synth.pixels(3, 56, 2))
        feats_dp = e.encode_images_dp(tower, px3)
        torch.cuda.synchronize()
        err = C.c_int(0)
        _lib.check(_lib.lib().omchat_peer_error(peer, C.byref(err)))
        st = e.comm_stats()
        dist.barrier()
        q.put((rank, feats.float().cpu().numpy(), logits.float().cpu().numpy(), toks, err.value, st, feats_dp.float().cpu().numpy()))
        tower.close()
        e.close()
        _lib.lib().omchat_peer_destroy(peer)
    except BaseException as e:      # noqa
        import traceback
        q.put((rank, None, traceback.format_exc(), None, 1, None, None))
    finally:
        dist.destroy_process_group()


def _spawn(target, size, timeout=240, extra=()):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, size, port, q) + tuple(extra)) for r in range(size)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in procs:
            res.append(q.get(timeout=timeout))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    return sorted(res, key=lambda x: x[0])


@pytest.mark.parametrize("size", [2, 4])
def test_peer_fused_reduce_resid_rmsnorm_equals_two_launches(gpu_lib, size):
    res = _spawn(_proc_fused_norm, size)
    assert len(res) == size
    for rank, ok, detail in res:
        assert ok, (rank, detail)


@pytest.mark.parametrize("size,fast", [(2, False), (4, False), (8, False), (4, True)])
def test_peer_allreduce_across_processes_ipc(gpu_lib, size, fast):
    res = _spawn(_proc_selftest, size, extra=(fast,))
    assert len(res) == size
    for rank, ok, detail in res:
        assert ok, (rank, detail)


def test_tp2_engine_processes_over_peer_allreduce_equal_tp1(gpu_lib):
    """two rank PROCESSES, every tensor-parallel sum on the peer kernels, weights = device-side shards of the synthetic TP = 1
    values: features / logits agree with the TP = 1 engine within the multi-layer tolerance, both ranks bit-identical"""
    from gpu_util import rel, TOL_DEEP
    from omchat_amd.engine import Engine
    res = _spawn(_proc_tp_engine, 2)
    for r in res:
        assert r[1] is not None, r[2]
        assert r[4] == 0
        assert r[5]["peer_allreduces"] > 0 and r[5]["rccl_allreduces"] == 0
    assert np.array_equal(res[0][1], res[1][1])
    assert res[0][3] == res[1][3]
    # the fused decode launch (all-reduce + residual + RMSNorm) changes nothing: same token ids with it switched off
    res2 = _spawn(_proc_tp_engine, 2, extra=(0,))
    assert res2[0][1] is not None, res2[0][2]
    assert res2[0][3] == res[0][3] and res2[1][3] == res[0][3]
    cfg = tiny(q_heads=7, kv_heads=1, heads_v=3)
    e = Engine(cfg, dtype="bf16", max_seq=128, max_batch=2, max_tiles=2)
    e.fill_synthetic(13)
    px = torch.from_numpy(synth.pixels(2, 56, 1))
    ids = torch.tensor([[3, -200, 17, -200, 19, 20]])
    feats = e.encode_images(px)
    embeds, lengths, _ = e.splice(ids, None, feats)
    logits, _ = e.prefill(embeds, lengths)
    torch.cuda.synchronize()
    assert rel(torch.from_numpy(res[0][1]), feats) < TOL_DEEP["bf16"]
    full = np.concatenate([res[0][2], res[1][2]], axis=-1)
    assert rel(torch.from_numpy(full), logits) < TOL_DEEP["bf16"]
    # the data-parallel tower gathers exactly the bits one tower produces (x + 0 is exact), on both ranks
    f3 = e.encode_images(torch.from_numpy(synth.pixels(3, 56, 2))); torch.cuda.synchronize()
    assert np.array_equal(res[0][6], f3.float().cpu().numpy()) and np.array_equal(res[1][6], res[0][6])
    e.close()
