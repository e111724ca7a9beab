"""Experiment (round 5): is a TP rank's decode step latency-bound enough that TWO independent half-batch chains on two streams finish
sooner than one full-batch chain?  Rank 0 of a TP = N group (shard shapes, exchanges removed) on one GPU.

  A: one context, b = B            (today's path)
  B: one context, b = B / 2 alone  (what one chain costs)
  C: two contexts of b = B / 2, each driven by its own host thread on its own stream, concurrently

usage: python3 tools/exp_two_chains.py [N=8] [B=32] [S=3584] [steps=64]"""
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
from omchat_amd.config import omchat13b
from omchat_amd.engine import Engine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = int(sys.argv[3]) if len(sys.argv) > 3 else 3584
STEPS = int(sys.argv[4]) if len(sys.argv) > 4 else 64
cfg = omchat13b()
GRAPH = len(sys.argv) > 5 and sys.argv[5] == 'graph'


def mk(b):
    e = Engine(cfg, dtype="bf16", max_seq=S + 4 * STEPS + 64, max_batch=b, max_tiles=1, max_prefill_rows=S * b, tp_rank=0, tp_size=N, vision=False)
    e.fill_synthetic(0, local=True)
    e.set_noop_allreduce()
    if GRAPH:
        e.enable_decode_graph(True)
    g = torch.Generator(device="cuda").manual_seed(1)
    emb = (torch.randn(b, S, 3584, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    e.prefill(emb, [S] * b, want_logits=False)
    torch.cuda.synchronize()
    return e


def run(e, b, steps, stream):
    tok = torch.zeros(b, dtype=torch.int32, device="cuda")
    with torch.cuda.stream(stream):
        for _ in range(steps):
            tok, _ = e.decode_step(tok)


def timed(engs, bs, steps):
    streams = [torch.cuda.Stream() for _ in engs]
    for e, b, s in zip(engs, bs, streams):       # warm-up
        run(e, b, 4, s)
    torch.cuda.synchronize()
    ths = [threading.Thread(target=run, args=(e, b, steps, s)) for e, b, s in zip(engs, bs, streams)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps


a = mk(B)
tA = timed([a], [B], STEPS)
a.close(); del a; torch.cuda.empty_cache()
h1, h2 = mk(B // 2), mk(B // 2)
tB = timed([h1], [B // 2], STEPS)
tC = timed([h1, h2], [B // 2, B // 2], STEPS)
print(f"graph={GRAPH} TP rank 0 of {N}, S = {S}: one chain b = {B}: {tA:.3f} ms/step; one chain b = {B // 2}: {tB:.3f}; two concurrent chains of b = {B // 2}: {tC:.3f} ms per step pair")
if B >= 4:
    q = [h1, h2, mk(B // 2), mk(B // 2)]
    # four chains of b / 4 would need b / 4 contexts; here: four b / 2 chains to see how far concurrency scales (2 x the work of C)
    tD = timed(q, [B // 2] * 4, STEPS)
    print(f"four concurrent chains of b = {B // 2}: {tD:.3f} ms per step quadruple (= {tD / 2:.3f} per {B} sequences)")
