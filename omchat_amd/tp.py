"""Tensor-parallel sharding plan (Megatron-style, one all-reduce per sub-block) -- new functionality required by the
north star; the reference has no tensor parallelism.

ViT (25 heads): heads are zero-padded to a multiple of the TP degree (26/28/32); padded q/k/v rows and proj columns
are zero, and the joint q/k RMSNorm keeps its divisor at the full 3200 channels with the sum of squares all-reduced
across ranks (modeling_intern_vit.py:143-146).  Decoder (28 q / 4 kv heads): kv heads are split while they divide
the degree; beyond that each kv head is replicated on `tp/kv` ranks which split its 7 query heads 4+3 (+1 zero head).
Column-parallel: qkv / fc1 / gate / up / lm_head rows.  Row-parallel: proj / fc2 / o_proj / down_proj columns."""
import numpy as np


def _ceil(a, b):
    return (a + b - 1) // b


def local_dims(cfg, rank, size):
    v, t = cfg.vision, cfg.text
    nq, nkv = t["num_attention_heads"], t["num_key_value_heads"]
    if size == 1:
        return dict(v_heads=v["num_attention_heads"], v_mlp=v["intermediate_size"], t_heads=nq, t_kv_heads=nkv,
                    t_mlp=t["intermediate_size"], t_vocab=t["vocab_size"])
    if v["intermediate_size"] % (size * 64) or t["intermediate_size"] % (size * 64) or t["vocab_size"] % size:
        raise ValueError("intermediate sizes must be multiples of 64*tp and the vocabulary a multiple of tp")
    if nkv % size == 0:
        if nq % size:
            raise ValueError("query heads must divide by tp when kv heads do")
        tq, tkv = nq // size, nkv // size
    elif size % nkv == 0:
        tq, tkv = _ceil(nq // nkv, size // nkv), 1
    else:
        raise ValueError(f"tp={size} incompatible with {nkv} kv heads")
    return dict(v_heads=_ceil(v["num_attention_heads"], size), v_mlp=v["intermediate_size"] // size, t_heads=tq, t_kv_heads=tkv,
                t_mlp=t["intermediate_size"] // size, t_vocab=t["vocab_size"] // size)


def decoder_head_map(cfg, rank, size):
    """(list of global q head ids or -1 for a zero head, list of global kv head ids) owned by `rank`."""
    t = cfg.text
    nq, nkv = t["num_attention_heads"], t["num_key_value_heads"]
    d = local_dims(cfg, rank, size)
    if size == 1 or nkv % size == 0:
        return list(range(rank * d["t_heads"], (rank + 1) * d["t_heads"])), list(range(rank * d["t_kv_heads"], (rank + 1) * d["t_kv_heads"]))
    rep = size // nkv
    grp = nq // nkv
    j, sub = rank // rep, rank % rep
    qs = [j * grp + sub * d["t_heads"] + i for i in range(d["t_heads"])]
    qs = [q if q < (j + 1) * grp else -1 for q in qs]
    return qs, [j]


def vit_head_map(cfg, rank, size):
    """Balanced contiguous split of the real heads, each rank zero-padded (-1) to ceil(heads / size)."""
    h = cfg.vision["num_attention_heads"]
    hl = _ceil(h, size)
    base, extra = divmod(h, size)
    start = rank * base + min(rank, extra)
    n = base + (1 if rank < extra else 0)
    return list(range(start, start + n)) + [-1] * (hl - n)


def _is_torch(x):
    try:
        import torch
        return torch.is_tensor(x)
    except ImportError:
        return False


def _cat(parts):
    if _is_torch(parts[0]):
        import torch
        return torch.cat(parts, dim=0)
    return np.concatenate(parts, axis=0)


def _t(x):
    return x.t() if _is_torch(x) else x.T


def _take_heads(x, heads, axis_len_per_head=128):
    """x [..heads*hd.., cols] along axis 0 -> rows of the selected heads (zeros for -1).  numpy or torch (any device / dtype)."""
    if not _is_torch(x):
        x = np.asarray(x)
    blocks = x.reshape(-1, axis_len_per_head, *x.shape[1:])
    z = blocks[0] * 0
    return _cat([blocks[h] if h >= 0 else z for h in heads])


def shard_tensor(name, tensor, cfg, rank, size):
    """Full tensor (numpy or torch, host or device) -> the rank-local tensor the C ABI expects under the same name.
    torch tensors keep their dtype and device (the 13B synthetic weights are sharded on the GPU), numpy goes through fp32."""
    if size == 1:
        return tensor
    if _is_torch(tensor):
        return _shard(name, tensor, cfg, rank, size).contiguous()
    return np.ascontiguousarray(_shard(name, np.asarray(tensor, dtype=np.float32), cfg, rank, size))


def _shard(name, x, cfg, rank, size):
    v, t = cfg.vision, cfg.text
    sl = lambda n: slice(rank * (n // size), (rank + 1) * (n // size))
    if ".encoder.layers." in name:
        vh = vit_head_map(cfg, rank, size)
        C = v["hidden_size"]
        hd = v.get("head_dim", 128)
        if name.endswith("attn.qkv.weight"):
            q, k, vv = x[:C], x[C:2 * C], x[2 * C:]
            return _cat([_take_heads(q, vh, hd), _take_heads(k, vh, hd), _take_heads(vv, vh, hd)])
        if name.endswith("attn.q_norm.weight") or name.endswith("attn.k_norm.weight"):
            return _take_heads(x, vh, hd)
        if name.endswith("attn.proj.weight"):
            return _t(_take_heads(_t(x), vh, hd))
        if name.endswith("mlp.fc1.weight") or name.endswith("mlp.fc1.bias"):
            return x[sl(v["intermediate_size"])]
        if name.endswith("mlp.fc2.weight"):
            return x[:, sl(v["intermediate_size"])]
        return x
    if name.startswith("model.layers."):
        qs, kvs = decoder_head_map(cfg, rank, size)
        if "q_proj" in name:
            return _take_heads(x, qs)
        if "k_proj" in name or "v_proj" in name:
            return _take_heads(x, kvs)
        if name.endswith("o_proj.weight"):
            return _t(_take_heads(_t(x), qs))
        if "gate_proj" in name or "up_proj" in name:
            return x[sl(t["intermediate_size"])]
        if name.endswith("down_proj.weight"):
            return x[:, sl(t["intermediate_size"])]
        return x
    if name == "lm_head.weight":
        return x[sl(t["vocab_size"])]
    return x


def rccl_libraries_mapped():
    """paths of every librccl mapped into this process (two different RCCL builds in one process are a bug: VERDICT r01)"""
    out = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "librccl" in line:
                    out.add(line.split()[-1])
    except OSError:
        pass
    return sorted(out)


def init_comm(rank, size):
    """Create the RCCL communicator of the tensor-parallel group inside the library (one process per GPU).
    The 128-byte unique id is created on rank 0 and shipped over the already-initialised torch.distributed group
    (any backend; it is bootstrap plumbing only -- the data path all-reduces run on RCCL over xGMI in C++)."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from . import _lib
    lib = _lib.lib()
    buf = C.create_string_buffer(128)
    if rank == 0:
        _lib.check(lib.omchat_comm_unique_id(buf))
    t = torch.tensor(list(buf.raw), dtype=torch.uint8)
    dist.broadcast(t, src=0)
    raw = bytes(t.tolist())
    comm = C.c_void_p()
    _lib.check(lib.omchat_comm_init(raw, rank, size, C.byref(comm)))
    libs = rccl_libraries_mapped()
    if len(libs) > 1:
        raise _lib.OmchatError(f"more than one RCCL build is mapped into this process: {libs}")
    n = C.c_int(0)
    _lib.check(lib.omchat_comm_count(comm, C.byref(n)))
    if n.value != size:
        raise _lib.OmchatError(f"RCCL communicator has {n.value} ranks, expected {size}")
    return comm


def init_peer(rank, size, cap_bytes=64 << 20, fast=False, oneshot_max=0, max_blocks=0):
    """Create this rank's peer all-reduce group member (csrc/comm.hip): allocate the shared buffer, exchange the 64-byte IPC
    handles over the initialised torch.distributed group (bootstrap plumbing), map every peer's buffer."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from . import _lib
    lib = _lib.lib()
    peer = C.c_void_p()
    hbuf = C.create_string_buffer(64)
    _lib.check(lib.omchat_peer_create(rank, size, cap_bytes, C.byref(peer), hbuf))
    mine = torch.tensor(list(hbuf.raw), dtype=torch.uint8)
    table = [torch.empty(64, dtype=torch.uint8) for _ in range(size)]
    dist.all_gather(table, mine)
    raw = b"".join(bytes(t.tolist()) for t in table)
    _lib.check(lib.omchat_peer_connect(peer, raw))
    _lib.check(lib.omchat_peer_set_mode(peer, int(fast), oneshot_max, max_blocks))
    dist.barrier()          # nobody launches a kernel that touches a peer buffer before every mapping exists
    return peer


def peer_selftest(peer, rank, size, iters=48, sizes=(16, 3584 * 4, 3 * 3584 * 4, 300 * 1024 + 16, 1 << 20, 1024 * 3200 * 2, 3584 * 3584 * 2 + 32)):
    """Run the peer all-reduce on data whose sum every rank can compute by itself (small integers: exact in fp32, bf16 and f16)
    and compare on the device.  Returns (ok, detail).  Catches stale reads / lost flags on the hardware it runs on."""
    import ctypes as C
    import torch
    from . import _lib
    lib = _lib.lib()
    bad = 0
    err = C.c_int(0)

    def one(it, nbytes, dt, code):
        n = nbytes // (4 if dt == torch.float32 else 2)
        i = torch.arange(n, device="cuda", dtype=torch.int64)
        val = lambda r: (((i * 7 + it * 13 + r * 5) % 31) - 15).to(torch.float32)      # |sum over 8 ranks| <= 120: exact in bf16
        x = val(rank).to(dt).contiguous()
        _lib.check(lib.omchat_peer_allreduce(peer, _lib.ptr(x), n, code, _lib.cur_stream()))
        ref = sum(val(r) for r in range(size))
        return int((x.to(torch.float32) != ref).sum())

    # Every verdict is taken COLLECTIVELY (MIN of the ok flag over the bootstrap group): a rank that left alone would leave the others
    # issuing all-reduces against a peer that no longer takes part -- one barrier timeout per call -- and would leave the per-block
    # epochs of the ranks out of step.
    def agree(ok):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or size == 1:
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t[0]))

    def healthy():
        _lib.check(lib.omchat_peer_error(peer, C.byref(err)))
        return agree(not err.value and not bad)

    # one small call first, checked at once: a group that cannot talk (a peer that never arrives, a mapping that does not work across
    # these devices) shows as a barrier timeout here, and the test must cost ONE timeout, not one per call
    bad += one(0, 3584 * 4, torch.float32, _lib.F32)
    if not healthy():
        return False, dict(mismatched_elements=bad, timeout=bool(err.value), iters=0)
    for it in range(iters):
        for nbytes in sizes:
            for dt, code in ((torch.float32, _lib.F32), (torch.bfloat16, _lib.BF16)):
                bad += one(it, nbytes, dt, code)
        if it in (0, 1, 7):                                  # early exits: never sit through hundreds of timeouts
            if not healthy():
                return False, dict(mismatched_elements=bad, timeout=bool(err.value), iters=it + 1)
    ok = healthy()
    return ok, dict(mismatched_elements=bad, timeout=bool(err.value), iters=iters)
