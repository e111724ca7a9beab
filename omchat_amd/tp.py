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


def _take_heads(x, heads, axis_len_per_head=128):
    """x [..heads*128.., cols] along axis 0 -> rows of the selected heads (zeros for -1)."""
    x = np.asarray(x)
    blocks = x.reshape(-1, axis_len_per_head, *x.shape[1:])
    z = np.zeros_like(blocks[0])
    return np.concatenate([blocks[h] if h >= 0 else z for h in heads], axis=0)


def shard_tensor(name, tensor, cfg, rank, size):
    """Full tensor (numpy or torch) -> the rank-local tensor the C ABI expects under the same name."""
    if size == 1:
        return tensor
    import torch
    is_torch = torch.is_tensor(tensor)
    x = tensor.detach().cpu().float().numpy() if is_torch else np.asarray(tensor, dtype=np.float32)
    out = _shard_np(name, x, cfg, rank, size)
    return torch.from_numpy(np.ascontiguousarray(out)) if is_torch else np.ascontiguousarray(out)


def _shard_np(name, x, cfg, rank, size):
    v, t = cfg.vision, cfg.text
    d = local_dims(cfg, rank, size)
    sl = lambda n: slice(rank * (n // size), (rank + 1) * (n // size))
    if ".encoder.layers." in name:
        vh = vit_head_map(cfg, rank, size)
        C = v["hidden_size"]
        hd = v.get("head_dim", 128)
        if name.endswith("attn.qkv.weight"):
            q, k, vv = x[:C], x[C:2 * C], x[2 * C:]
            return np.concatenate([_take_heads(q, vh, hd), _take_heads(k, vh, hd), _take_heads(vv, vh, hd)], axis=0)
        if name.endswith("attn.q_norm.weight") or name.endswith("attn.k_norm.weight"):
            return _take_heads(x, vh, hd)
        if name.endswith("attn.proj.weight"):
            return _take_heads(x.T, vh, hd).T
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
            return _take_heads(x.T, qs).T
        if "gate_proj" in name or "up_proj" in name:
            return x[sl(t["intermediate_size"])]
        if name.endswith("down_proj.weight"):
            return x[:, sl(t["intermediate_size"])]
        return x
    if name == "lm_head.weight":
        return x[sl(t["vocab_size"])]
    return x


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
    return comm
