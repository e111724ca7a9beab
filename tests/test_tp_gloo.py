"""CPU, world_size 2 / 4 / 8 over gloo: the tensor-parallel sharding plan (omchat_amd/tp.py) executed with the same dataflow as
the C++ loops (model.hip: rank 0 carries bias + residual, one all-reduce per sub-block, joint q/k-norm sum of squares
all-reduced with the divisor kept at the full channel count) must reproduce the unsharded oracle.

Round 6: the SEQUENCE-PARALLEL form of the same layers (model.hip gemm_sp, tuning key 45): a row-parallel projection ends in a
reduce-scatter over row blocks, the owner adds the residual and normalises ITS rows, an all-gather hands every rank the normalised
activation; the residual stream stays row-sharded between sub-blocks and is gathered once at the end."""
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from omchat_amd import synth, tp
from omchat_amd.config import tiny
import oracle
from oracle.pipeline import _sub, TOWER_PFX

T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _vit_layer_tp(x, sd_full, cfg, rank, size):
    P = synth.TOWER + "encoder.layers.0."
    w = {k[len(P):]: T(tp.shard_tensor(k, v, cfg, rank, size)) for k, v in sd_full.items() if k.startswith(P)}
    C = cfg.vision["hidden_size"]
    hl = tp.local_dims(cfg, rank, size)["v_heads"]
    B, N, _ = x.shape
    xn = oracle.rms_norm(x, w["norm1.weight"])
    qkv = F.linear(xn, w["attn.qkv.weight"]).reshape(B, N, 3, hl * 128)
    q, k, v = qkv.unbind(2)
    ss = torch.stack([q.pow(2).sum(-1), k.pow(2).sum(-1)], -1)
    dist.all_reduce(ss)                                              # launch_vit_qk_sumsq + allreduce_f32
    q = w["attn.q_norm.weight"] * (q * torch.rsqrt(ss[..., 0:1] / C + 1e-6))
    k = w["attn.k_norm.weight"] * (k * torch.rsqrt(ss[..., 1:2] / C + 1e-6))
    q = q.view(B, N, hl, 128).transpose(1, 2); k = k.view(B, N, hl, 128).transpose(1, 2); v = v.reshape(B, N, hl, 128).transpose(1, 2)
    a = ((q * 128 ** -0.5) @ k.transpose(-2, -1)).softmax(-1) @ v
    a = a.transpose(1, 2).reshape(B, N, hl * 128)
    y = F.linear(a, w["attn.proj.weight"], w["attn.proj.bias"] if rank == 0 else None) * w["ls1"]
    if rank == 0:
        y = y + x
    dist.all_reduce(y)
    x = y
    xn = oracle.rms_norm(x, w["norm2.weight"])
    h = F.gelu(F.linear(xn, w["mlp.fc1.weight"], w["mlp.fc1.bias"]))
    y = F.linear(h, w["mlp.fc2.weight"], w["mlp.fc2.bias"] if rank == 0 else None) * w["ls2"]
    if rank == 0:
        y = y + x
    dist.all_reduce(y)
    return y


def _dec_layer_tp(x, sd_full, cfg, rank, size):
    P = "model.layers.0."
    w = {k: T(tp.shard_tensor(k, v, cfg, rank, size)) for k, v in sd_full.items() if k.startswith(P)}
    d = tp.local_dims(cfg, rank, size)
    local = dict(cfg.text); local["num_attention_heads"] = d["t_heads"]; local["num_key_value_heads"] = d["t_kv_heads"]
    b, S, _ = x.shape
    cos, sin = oracle.rope_cos_sin(torch.arange(S)[None], 128, cfg.text["rope_theta"], torch.float32)
    xn = oracle.rms_norm(x, w[P + "input_layernorm.weight"])
    y = oracle.qwen2_attention(xn, w, P, local, cos, sin, None, 0)   # o_proj has no bias: partial sums
    if rank == 0:
        y = y + x
    dist.all_reduce(y)
    x = y
    xn = oracle.rms_norm(x, w[P + "post_attention_layernorm.weight"])
    y = oracle.qwen2_mlp(xn, w, P)
    if rank == 0:
        y = y + x
    dist.all_reduce(y)
    # vocab-parallel lm_head: all-gather of the local logit slices
    lm = T(tp.shard_tensor("lm_head.weight", sd_full["lm_head.weight"], cfg, rank, size))
    part = F.linear(y, lm)
    parts = [torch.empty_like(part) for _ in range(size)]
    dist.all_gather(parts, part)
    return y, torch.cat(parts, -1)


# ---- sequence-parallel dataflow (model.hip gemm_sp): row blocks of cdiv(M, size) rows, rank r owns block r ------------------------------
def _blocks(M, size):
    blk = -(-M // size)
    return blk, [(min(r * blk, M), min((r + 1) * blk, M)) for r in range(size)]


def _reduce_scatter_rows(p, rank, size):
    """p [M, N] partial of this rank -> the summed rows this rank owns (gloo has no reduce_scatter: all-reduce, keep the own block -- the
    values ncclReduceScatter delivers)"""
    M = p.shape[0]
    _, blocks = _blocks(M, size)
    tot = p.clone()
    dist.all_reduce(tot)
    lo, hi = blocks[rank]
    return tot[lo:hi], (lo, hi)


def _all_gather_rows(mine, M, rank, size):
    """every rank contributes its row block -> the whole [M, N] on every rank (padded to equal blocks, as ncclAllGather needs)"""
    blk, blocks = _blocks(M, size)
    pad = torch.zeros(blk, mine.shape[1])
    pad[:mine.shape[0]] = mine
    parts = [torch.empty_like(pad) for _ in range(size)]
    dist.all_gather(parts, pad)
    return torch.cat([parts[r][:blocks[r][1] - blocks[r][0]] for r in range(size)], dim=0)


def _sp_sub_block(partial, x_own, own, norm_w, rank, size, M):
    """gemm_sp after the GEMM: reduce-scatter the partials, residual + norm on the owned rows, all-gather the normed rows"""
    tot, (lo, hi) = _reduce_scatter_rows(partial, rank, size)
    assert (lo, hi) == own
    x_own = x_own + tot
    xn_own = oracle.rms_norm(x_own, norm_w) if norm_w is not None else None
    xn = _all_gather_rows(xn_own, M, rank, size) if norm_w is not None else None
    return x_own, xn


def _vit_layer_tp_sp(x, sd_full, cfg, rank, size):
    P = synth.TOWER + "encoder.layers.0."
    w = {k[len(P):]: T(tp.shard_tensor(k, v, cfg, rank, size)) for k, v in sd_full.items() if k.startswith(P)}
    C = cfg.vision["hidden_size"]
    hl = tp.local_dims(cfg, rank, size)["v_heads"]
    B, N, _ = x.shape
    M = B * N
    _, blocks = _blocks(M, size)
    own = blocks[rank]
    xf = x.reshape(M, C)
    x_own = xf[own[0]:own[1]].clone()                               # the residual stream: this rank keeps ITS rows current
    xn = oracle.rms_norm(xf, w["norm1.weight"])                     # layer 0: the embeddings are replicated
    qkv = F.linear(xn, w["attn.qkv.weight"]).reshape(B, N, 3, hl * 128)
    q, k, v = qkv.unbind(2)
    ss = torch.stack([q.pow(2).sum(-1), k.pow(2).sum(-1)], -1)
    dist.all_reduce(ss)
    q = w["attn.q_norm.weight"] * (q * torch.rsqrt(ss[..., 0:1] / C + 1e-6))
    k = w["attn.k_norm.weight"] * (k * torch.rsqrt(ss[..., 1:2] / C + 1e-6))
    q = q.view(B, N, hl, 128).transpose(1, 2); k = k.view(B, N, hl, 128).transpose(1, 2); v = v.reshape(B, N, hl, 128).transpose(1, 2)
    a = ((q * 128 ** -0.5) @ k.transpose(-2, -1)).softmax(-1) @ v
    a = a.transpose(1, 2).reshape(M, hl * 128)
    part = F.linear(a, w["attn.proj.weight"], w["attn.proj.bias"] if rank == 0 else None) * w["ls1"]      # bias on rank 0, layer scale everywhere, NO residual
    x_own, xn = _sp_sub_block(part, x_own, own, w["norm2.weight"], rank, size, M)
    h = F.gelu(F.linear(xn, w["mlp.fc1.weight"], w["mlp.fc1.bias"]))
    part = F.linear(h, w["mlp.fc2.weight"], w["mlp.fc2.bias"] if rank == 0 else None) * w["ls2"]
    x_own, _ = _sp_sub_block(part, x_own, own, None, rank, size, M)                                        # last layer: no norm, gather the stream
    return _all_gather_rows(x_own, M, rank, size).reshape(B, N, C)


def _dec_layer_tp_sp(x, sd_full, cfg, rank, size):
    P = "model.layers.0."
    w = {k: T(tp.shard_tensor(k, v, cfg, rank, size)) for k, v in sd_full.items() if k.startswith(P)}
    d = tp.local_dims(cfg, rank, size)
    local = dict(cfg.text); local["num_attention_heads"] = d["t_heads"]; local["num_key_value_heads"] = d["t_kv_heads"]
    b, S, H = x.shape
    M = b * S
    _, blocks = _blocks(M, size)
    own = blocks[rank]
    x_own = x.reshape(M, H)[own[0]:own[1]].clone()
    cos, sin = oracle.rope_cos_sin(torch.arange(S)[None], 128, cfg.text["rope_theta"], torch.float32)
    xn = oracle.rms_norm(x, w[P + "input_layernorm.weight"])
    part = oracle.qwen2_attention(xn, w, P, local, cos, sin, None, 0).reshape(M, H)
    x_own, xn = _sp_sub_block(part, x_own, own, w[P + "post_attention_layernorm.weight"], rank, size, M)
    part = oracle.qwen2_mlp(xn.reshape(b, S, H), w, P).reshape(M, H)
    x_own, _ = _sp_sub_block(part, x_own, own, None, rank, size, M)
    y = _all_gather_rows(x_own, M, rank, size).reshape(b, S, H)
    lm = T(tp.shard_tensor("lm_head.weight", sd_full["lm_head.weight"], cfg, rank, size))
    part = F.linear(y, lm)
    parts = [torch.empty_like(part) for _ in range(size)]
    dist.all_gather(parts, part)
    return y, torch.cat(parts, -1)



def _worker(rank, size, port, case, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    torch.manual_seed(0)
    sp = case.endswith("+sp")
    case = case[:-3] if sp else case
    try:
        if case.startswith("vit"):
            hv = 3 if case == "vit" else 5                          # 3 heads over 2 ranks: [0,1] and [2,pad]; 5 heads over 4 / 8 ranks:
            cfg = tiny(heads_v=hv, layers_v=1)                      # the 25-head pattern (one rank with an extra real head, the rest padded)
            sd = synth.state_dict(cfg, 3, synth.TOWER)
            x = torch.randn(2, 17, 128 * hv)
            out = (_vit_layer_tp_sp if sp else _vit_layer_tp)(x, sd, cfg, rank, size)
            ref = oracle.vit_layer(x, _sub({k: T(v) for k, v in sd.items()}, TOWER_PFX), 0, hv)
            err = float((out - ref).norm() / ref.norm())
        else:
            # kv heads split / kv head replicated with 4+3(+pad) q heads / the Qwen2-7B head pattern (28 q, 4 kv): TP = 4 splits the kv
            # heads (7 q each), TP = 8 replicates every kv head on two ranks that take 4 and 3 (+1 zero) of its query heads
            qh, kvh = {"dec_split": (4, 2), "dec_replicated_kv": (7, 1), "dec28_4": (28, 4)}[case]
            cfg = tiny(q_heads=qh, kv_heads=kvh, layers_t=1)
            sd = {k: v for k, v in synth.state_dict(cfg, 4).items() if k.startswith(("model.layers.0.", "lm_head"))}
            x = torch.randn(1, 9, 256)
            out, logits = (_dec_layer_tp_sp if sp else _dec_layer_tp)(x, sd, cfg, rank, size)
            sdt = {k: T(v) for k, v in sd.items()}
            cos, sin = oracle.rope_cos_sin(torch.arange(9)[None], 128, 1e6, torch.float32)
            ref = oracle.qwen2_layer(x, sdt, 0, cfg.text, cos, sin, None)
            err = max(float((out - ref).norm() / ref.norm()), float((logits - F.linear(ref, sdt["lm_head.weight"])).norm() / logits.norm()))
        q.put((rank, err))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,size", [("vit", 2), ("dec_split", 2), ("dec_replicated_kv", 2),
                                       ("vit5", 4), ("vit5", 8), ("dec28_4", 4), ("dec28_4", 8),
                                       # sequence-parallel norms (round 6): 34 ViT rows / 9 decoder rows over 2 / 4 / 8 ranks -- ragged last
                                       # blocks, and ranks that own NO row (9 rows over 8 ranks: blocks of 2, ranks 5..7 empty)
                                       ("vit+sp", 2), ("dec_split+sp", 2), ("dec_replicated_kv+sp", 2),
                                       ("vit5+sp", 4), ("vit5+sp", 8), ("dec28_4+sp", 4), ("dec28_4+sp", 8)])
def test_tp_equals_unsharded(case, size):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, size, port, case, q)) for r in range(size)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err in res:
        assert err < 1e-5, (case, rank, err)
