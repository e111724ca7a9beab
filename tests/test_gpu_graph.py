"""Decode step replayed as a hipGraph (omchat_enable_decode_graph): same kernels, so tokens and logits must be bit-identical
to the eager step -- across re-captures (sequence outgrows the captured split-KV grid), batch sizes, the fp8 replica, and with
profiling brackets interleaved."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
from gpu_util import rnd, sync, randn
from omchat_amd import synth
from omchat_amd.config import tiny
from omchat_amd.engine import Engine

T32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()


def _run(e, x, lens, first, steps, want_logits):
    e.prefill(x, lengths=lens)
    tok, toks, lgs = first.clone(), [], []
    for _ in range(steps):
        tok, lg = e.decode_step(tok, want_logits=want_logits)
        toks.append(tok.clone())
        if want_logits:
            lgs.append(lg.clone())
    sync()
    return torch.stack(toks).cpu(), (torch.stack(lgs).cpu() if want_logits else None)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("b", [1, 3, 20])
def test_graph_replay_equals_eager(gpu_lib, dt, b):
    cfg = tiny()
    e = Engine(cfg, dtype=dt, max_seq=2048, max_batch=b, vision=False)
    e.load_state_dict({k: v for k, v in synth.state_dict(cfg, 3).items() if not k.startswith(synth.TOWER) and "mm_projector" not in k})
    x = rnd(randn((b, 12, 256), 1, 0.5), dt)
    lens = [12 - (i % 4) for i in range(b)]
    first = torch.arange(5, 5 + b, dtype=torch.int32)
    t0, l0 = _run(e, x, lens, first, 6, True)
    e.enable_decode_graph(True)
    t1, l1 = _run(e, x, lens, first, 6, True)
    st = e.decode_graph_stats()
    assert st["replays"] == 6 and st["captures"] == 1
    assert torch.equal(t0, t1) and torch.equal(l0, l1)
    # tokens only (the bench path), and a second run re-uses the captured graph
    t2, _ = _run(e, x, lens, first, 6, False)
    assert torch.equal(t0, t2) and e.decode_graph_stats()["captures"] == 1
    assert e.kv_lengths(b) == [n + 6 for n in lens]
    e.enable_decode_graph(False)
    t3, _ = _run(e, x, lens, first, 6, False)
    assert torch.equal(t0, t3) and e.decode_graph_stats()["replays"] == 12
    e.close()


def test_graph_recaptures_when_the_sequence_outgrows_the_grid(gpu_lib):
    cfg = tiny()
    e = Engine(cfg, dtype="bf16", max_seq=1400, max_batch=1, vision=False)
    e.load_state_dict({k: v for k, v in synth.state_dict(cfg, 3).items() if not k.startswith(synth.TOWER) and "mm_projector" not in k})
    x = rnd(randn((1, 8, 256), 1, 0.5), "bf16")
    first = torch.tensor([7], dtype=torch.int32)
    n = 1100                                                # capture covers 8 + 1 + 1024 keys -> one re-capture on the way
    t0, _ = _run(e, x, [8], first, n, False)
    e.enable_decode_graph(True)
    t1, _ = _run(e, x, [8], first, n, False)
    st = e.decode_graph_stats()
    assert st["captures"] == 2 and st["replays"] == n
    assert torch.equal(t0, t1)
    with pytest.raises(ValueError):                         # KV cache full is still reported
        for _ in range(400):
            e.decode_step(first)
    e.close()


def test_graph_with_fp8_replica_and_profiling_brackets(gpu_lib):
    cfg = tiny()
    e = Engine(cfg, dtype="bf16", max_seq=256, max_batch=1, vision=False)
    e.load_state_dict({k: v for k, v in synth.state_dict(cfg, 3).items() if not k.startswith(synth.TOWER) and "mm_projector" not in k})
    x = rnd(randn((1, 10, 256), 1, 0.5), "bf16")
    first = torch.tensor([9], dtype=torch.int32)
    e.enable_fp8_decode(True)
    t0, l0 = _run(e, x, [10], first, 20, True)
    e.enable_decode_graph(True)
    e.prof_enable(True)                                     # every 8th step runs eagerly with event brackets
    t1, l1 = _run(e, x, [10], first, 20, True)
    e.prof_enable(False)
    assert torch.equal(t0, t1) and torch.equal(l0, l1)
    st = e.decode_graph_stats()
    assert 0 < st["replays"] < 20
    ms, cnt = e.prof_read(0)
    assert cnt >= 2 and ms > 0
    e.enable_fp8_decode(False)                              # a different graph (16-bit weights) is captured for the same batch
    t2, _ = _run(e, x, [10], first, 4, False)
    assert e.decode_graph_stats()["captures"] == 2
    e.close()
